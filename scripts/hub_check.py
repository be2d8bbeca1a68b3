import sys, os
import numpy as np, torch
sys.path.insert(0, os.getcwd())
from hpc_amd import CSR, SpMMOpt, synth
dev = torch.device("cuda:0")
def timed(f, warm=3, reps=20):
    for _ in range(warm): f()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(reps): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
for name, M, nnz_t, mx in (("am", 881_680, 5_668_682, 154_828), ("arxiv", 169_343, 1_166_243, 13_155), ("youtube", 1_138_499, 5_980_886, 28_754)):
    ptr, idx = synth.csr_powerlaw(M, nnz_t / M, mx, seed=sum(map(ord, name)) % 1000 + 1, force_max=True)
    vals = synth.make_values(idx.size)
    d = [torch.from_numpy(a).to(dev) for a in (ptr, idx, vals)]
    for N in (32, 256):
        B = torch.randn(M, N, device=dev) * 0.1; C = torch.empty(M, N, device=dev)
        op = SpMMOpt(CSR(M, idx.size, *d), N); op.preprocess(B, C)
        print(name, N, round(timed(lambda: op.run(B, C)), 4), "ms  long", op.get_option("n_long_rows"), "chunks", op.get_option("n_chunks"), "thr", op.get_option("long_row_threshold"), flush=True)
