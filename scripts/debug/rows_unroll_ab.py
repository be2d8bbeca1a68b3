#!/usr/bin/env python3
"""A/B: rows kernel with 8 (default) vs 16 B-row gathers in flight per lane group ("rows_unroll"), interleaved, bits compared."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from hpc_amd import CSR, SpMMOpt, synth
from hpc_amd.spmm import count_bitdiff

dev = torch.device("cuda:0")
graphs = {
    "banded-long-rows": lambda: synth.csr_banded_long_rows_device(1 << 17, dev),
    "protein-unsorted": lambda: synth.csr_dataset_structured_device("protein", dev, sort_cols=False),
    "ppa-community": lambda: synth.csr_dataset_structured_device("ppa", dev),
    "products-community": lambda: synth.csr_dataset_structured_device("products", dev),
    "yelp-community": lambda: synth.csr_dataset_structured_device("yelp", dev),
    "citation-community": lambda: synth.csr_dataset_structured_device("citation", dev),
    "C1-uniform": lambda: tuple(torch.from_numpy(a).to(dev) for a in synth.csr_uniform(1 << 20, 16, 48)),
    "sbm": lambda: synth.csr_dcsbm_device(1 << 20, 32 << 20, 64, dev, alpha=0, mean_comm=2048, p_in=0.9),
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
    for N in (32, 128, 256):
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
                          "mthr": ops[0].get_option("medium_row_threshold"), "launches": ops[0].get_option("n_launches")}), flush=True)
        del ops, Cs, B
