#!/usr/bin/env python3
"""Hub thresholds above the auto rule's largest candidate (8192): would rows of 8 - 32 K nonzeros be better off as segments (in column strips where those apply)?
    python scripts/debug/hub_threshold_high_sweep.py reddit.dgl amazon_cogdl products --N 32 128"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from hpc_amd import CSR, SpMMOpt, synth

ap = argparse.ArgumentParser()
ap.add_argument("names", nargs="+")
ap.add_argument("--N", type=int, nargs="+", default=[32, 128])
args = ap.parse_args()
dev = torch.device("cuda", 0)
for name in args.names:
    ptr, idx = synth.csr_dataset_shaped(name)
    M, nnz = ptr.size - 1, idx.size
    d_ptr, d_idx = torch.from_numpy(ptr).to(dev), torch.from_numpy(idx).to(dev)
    d_val = torch.randn(nnz, device=dev) * 0.1
    print(name, M, nnz, flush=True)
    for N in args.N:
        d_B = (torch.randn(M, N, device=dev) * 0.1).contiguous()
        ops = {}
        for thr in (0, 8192, 16384, 32768, 65536):
            op = SpMMOpt(CSR(M, nnz, d_ptr, d_idx, d_val), N)
            op.set_option("long_row_threshold", thr)
            C = torch.empty(M, N, device=dev)
            op.preprocess(d_B, C)
            for _ in range(2): op.run(d_B, C)
            ops[thr] = (op, C)
        best = {t: 1e9 for t in ops}
        for rnd in range(3):
            for t, (op, C) in ops.items():
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize(); a.record()
                for _ in range(5): op.run(d_B, C)
                b.record(); torch.cuda.synchronize()
                best[t] = min(best[t], a.elapsed_time(b) / 5)
        for t, (op, C) in ops.items():
            print(f"  N {N:4d} hub threshold {t:6d} -> {op.get_option('long_row_threshold'):6d}: hubs {op.get_option('n_hub_rows'):5d}  strips {op.get_option('n_col_strips'):2d}  {best[t]:8.3f} ms", flush=True)
        del ops, d_B
        torch.cuda.empty_cache()
    del d_ptr, d_idx, d_val
    torch.cuda.empty_cache()
