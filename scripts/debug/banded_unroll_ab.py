#!/usr/bin/env python3
"""A/B on BANDED graphs only: rows kernel 8 vs 16 gathers in flight ("rows_unroll"), interleaved, bits compared -- is "banded -> 16" a rule or one graph?"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from hpc_amd import CSR, SpMMOpt, synth
from hpc_amd.spmm import count_bitdiff

dev = torch.device("cuda:0")
cases = [(1 << 17, 2048, 300, 700), (1 << 17, 512, 300, 700), (1 << 17, 2048, 100, 200), (1 << 18, 2048, 50, 100), (1 << 20, 2048, 16, 48), (1 << 20, 256, 4, 12),
         (1 << 19, 4096, 64, 128), (1 << 16, 1024, 600, 1000)]
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
def batch(f, reps):
    torch.cuda.synchronize(); ev[0].record()
    for _ in range(reps): f()
    ev[1].record(); torch.cuda.synchronize()
    return ev[0].elapsed_time(ev[1]) / reps
for M, width, lo, hi in cases:
    p, i = synth.csr_banded_long_rows_device(M, dev, width=width, lo=lo, hi=hi, seed=11)
    nnz = int(i.numel())
    v = torch.randn(nnz, device=dev) * 0.1
    for N in (64, 128, 256, 512):
        B = torch.randn(M, N, device=dev) * 0.1
        ops, Cs = [], []
        for u in (8, 16):
            op = SpMMOpt(CSR(M, nnz, p, i, v), N); op.set_option("rows_unroll", u); C = torch.full((M, N), float("nan"), device=dev); op.preprocess(B, C)
            for _ in range(2): op.run(B, C)
            ops.append(op); Cs.append(C)
        t = [[], []]
        for _ in range(4):
            for k in (0, 1): t[k].append(batch(lambda: ops[k].run(B, Cs[k]), 5))
        print(json.dumps({"M": M, "width": width, "deg": [lo, hi], "N": N, "ms8": round(min(t[0]), 4), "ms16": round(min(t[1]), 4), "ratio": round(min(t[1]) / min(t[0]), 3),
                          "bitdiff": count_bitdiff(Cs[0], Cs[1])[0], "local": ops[0].get_option("column_locality_pct"), "mthr": ops[0].get_option("medium_row_threshold"),
                          "tile": ops[0].get_option("lanes_per_row") * 4, "segments": ops[0].get_option("n_chunks"), "strips": ops[0].get_option("n_col_strips"), "launches": ops[0].get_option("n_launches")}), flush=True)
        del ops, Cs, B
