#!/usr/bin/env python3
"""Debug: SpMMRef-style handle preprocessed first, then a default handle on the same CSR (native harness order)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hpc_amd import CSR, SpMMOpt, synth
from hpc_amd.spmm import MiSpmmError

dev = torch.device("cuda:0")
ptr, idx = synth.csr_powerlaw(30000, 20.0, 1500, seed=12)
vals = synth.normal_f32(idx.size, 3)
M, N = 30000, 32
d_ptr, d_idx, d_val = (torch.from_numpy(a).to(dev) for a in (ptr, idx, vals))
d_B = torch.randn(M, N, device=dev)
d_C = torch.empty(M, N, device=dev)
for first in ({"long_row_threshold": 1 << 30, "medium_row_threshold": 1 << 30, "block_path": 0}, {"long_row_threshold": 1 << 30},
              {"medium_row_threshold": 1 << 30}, {"block_path": 0}, {}):
    a = SpMMOpt(CSR(M, idx.size, d_ptr, d_idx, d_val), N)
    for k, v in first.items():
        a.set_option(k, v)
    a.preprocess(d_B, d_C)
    b = SpMMOpt(CSR(M, idx.size, d_ptr, d_idx, d_val), N)
    try:
        b.preprocess(d_B, d_C)
        print("first", first, "-> second ok")
    except MiSpmmError as e:
        print("first", first, "-> second FAILED", e)
    del a, b
