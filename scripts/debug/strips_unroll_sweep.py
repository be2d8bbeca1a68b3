#!/usr/bin/env python3
"""segment_unroll (B-row gathers in flight per lane group) x column strips: with L2-resident strips the gathers return sooner -- does a
shallower pipeline at higher occupancy pay?    python scripts/debug/strips_unroll_sweep.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from hpc_amd import CSR, SpMMOpt, synth

dev = torch.device("cuda", 0)
if len(sys.argv) > 1:                                    # a dataset-shaped graph (random ascending columns) instead of the stratified one
    ptr, idx = synth.csr_dataset_shaped(sys.argv[1])
    M = ptr.size - 1
    d_ptr, d_idx = torch.from_numpy(ptr).to(dev), torch.from_numpy(idx).to(dev)
else:
    M = 1 << 17
    d_ptr, d_idx = synth.csr_long_rows_device(M, dev)
nnz = int(d_idx.numel())
d_val = torch.randn(nnz, device=dev) * 0.1
print(sys.argv[1:] or "long rows (device generator)", M, nnz, flush=True)
for N in (32, 128, 256):
    d_B = (torch.randn(M, N, device=dev) * 0.1).contiguous()
    ops = {}
    for S in (0,):
        for un in (8, 16, 32):
            op = SpMMOpt(CSR(M, nnz, d_ptr, d_idx, d_val), N)
            op.set_option("col_strips", S)
            op.set_option("segment_unroll", un)
            C = torch.empty(M, N, device=dev)
            op.preprocess(d_B, C)
            for _ in range(2): op.run(d_B, C)
            ops[(S, un)] = (op, C)
    best = {k: 1e9 for k in ops}
    for rnd in range(3):
        for k, (op, C) in ops.items():
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(); a.record()
            for _ in range(10): op.run(d_B, C)
            b.record(); torch.cuda.synchronize()
            best[k] = min(best[k], a.elapsed_time(b) / 10)
    for k, (op, C) in ops.items():
        print(f"N {N:4d} col_strips {k[0]} -> {op.get_option('n_col_strips'):2d} strips, segment_unroll {k[1]:2d}: {best[k]:.3f} ms", flush=True)
    del ops, d_B
    torch.cuda.empty_cache()
