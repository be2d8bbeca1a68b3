"""A/B in one process, interleaved: the segment kernel on the caller's stream ("segment_overlap" = 0) or on a second side stream (= 1).
    python scripts/debug/segment_stream_ab.py"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hpc_amd import CSR, SpMMOpt, synth
dev = torch.device("cuda:0")
def timed(f, warm=2, reps=8):
    for _ in range(warm): f()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(reps): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
for name in ("citation", "wikikg2", "products", "ppa", "yelp", "protein", "reddit.dgl", "collab", "arxiv", "youtube", "am", "c2"):
    ptr, idx = synth.csr_powerlaw(1 << 20, 32.0, 4096) if name == "c2" else synth.csr_dataset_shaped(name)
    M = ptr.size - 1
    vals = synth.make_values(idx.size)
    d = [torch.from_numpy(a).to(dev) for a in (ptr, idx, vals)]
    for N in (32, 128, 256):
        B = torch.randn(M, N, device=dev) * 0.1; C = torch.empty(M, N, device=dev)
        ops = {}
        for so in (0, 1):
            op = SpMMOpt(CSR(M, idx.size, *d), N); op.set_option("segment_overlap", so); op.preprocess(B, C); ops[so] = op
        t = {0: [], 1: []}
        for rnd in range(3):
            for so in (0, 1):
                t[so].append(timed(lambda: ops[so].run(B, C)))
        o = ops[0]
        segmax = min(o.get_option("max_row_nnz"), o.get_option("long_row_threshold"))
        print(f"{name:11s} N={N:<4d} chunks {o.get_option('n_chunks'):7d} hubs {o.get_option('n_hub_rows'):5d} longest segment {segmax:6d} | same stream {np.median(t[0]):8.4f} | side stream {np.median(t[1]):8.4f} | side/same {np.median(t[1]) / np.median(t[0]):.3f}", flush=True)
        del ops, B, C
    del d
