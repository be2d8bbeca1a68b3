#!/usr/bin/env python3
"""A/B: rows kernel with 8 (default) vs 16 B-row gathers in flight per lane group ("rows_unroll"), interleaved, bits compared."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from hpc_amd import CSR, SpMMOpt, synth
from hpc_amd.spmm import count_bitdiff

dev = torch.device("cuda:0")
import numpy as np
def host(t): return tuple(torch.from_numpy(a).to(dev) for a in t)
graphs = {
    "C1-uniform": lambda: host(synth.csr_uniform(1 << 20, 16, 48)),
    "C2-powerlaw": lambda: host(synth.csr_powerlaw(1 << 20, 32.0, 4096)),
    "uniform-deg8": lambda: host(synth.csr_uniform(1 << 21, 4, 12)),
    "uniform-deg128": lambda: host(synth.csr_uniform(1 << 18, 96, 160)),
    "denseish": lambda: host(synth.csr_uniform(1 << 18, 300, 700)),
    "rmat20-permuted": lambda: host(synth.csr_rmat(20, 32)),
    "ppa-shuffled": lambda: synth.csr_dataset_structured_device("ppa", dev, order="shuffled"),
    "products-shuffled": lambda: synth.csr_dataset_structured_device("products", dev, order="shuffled"),
    "citation-shuffled": lambda: synth.csr_dataset_structured_device("citation", dev, order="shuffled"),
}
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
def batch(f, reps):
    torch.cuda.synchronize(); ev[0].record()
    for _ in range(reps): f()
    ev[1].record(); torch.cuda.synchronize()
    return ev[0].elapsed_time(ev[1]) / reps
for name, build in graphs.items():
    p, i = build()
    M, nnz = p.numel() - 1, int(i.numel())
    v = torch.randn(nnz, device=dev) * 0.1
    for N in (32, 64, 128, 256, 512):
        B = torch.randn(M, N, device=dev) * 0.1
        ops, Cs = [], []
        for u in (8, 16):
            op = SpMMOpt(CSR(M, nnz, p, i, v), N); op.set_option("rows_unroll", u); C = torch.full((M, N), float("nan"), device=dev); op.preprocess(B, C)
            for _ in range(2): op.run(B, C)
            ops.append(op); Cs.append(C)
        t = [[], []]
        for _ in range(4):
            for k in (0, 1): t[k].append(batch(lambda: ops[k].run(B, Cs[k]), 5))
        nd = count_bitdiff(Cs[0], Cs[1])[0]
        print(json.dumps({"graph": name, "N": N, "ms_unroll8": round(min(t[0]), 4), "ms_unroll16": round(min(t[1]), 4), "ratio": round(min(t[1]) / min(t[0]), 3), "bitdiff": nd,
                          "local": ops[0].get_option("column_locality_pct"), "mthr": ops[0].get_option("medium_row_threshold"), "launches": ops[0].get_option("n_launches")}), flush=True)
        del ops, Cs, B
