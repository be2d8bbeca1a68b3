#!/usr/bin/env python3
"""One-off scale check near the int32 limits: M = K = 2^24 + 4096 rows (K > 2^24 forces the 64-bit-address kernels),
degree ~ U{16..48} (~540 M nonzeros, 25 % of the int32 range), N = 32.  Row sample vs the oracle, bitwise."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hpc_amd import CSR, SpMMOpt, synth
from oracle import oracle

dev = torch.device("cuda:0")
M = (1 << 24) + 4096
N = 32
t = time.time()
g = np.random.Generator(np.random.Philox(key=[321, 0]))
deg = g.integers(16, 49, size=M, dtype=np.int64)
ptr = np.zeros(M + 1, dtype=np.int64); np.cumsum(deg, out=ptr[1:])
nnz = int(ptr[-1]); assert nnz < 2**31
idx = g.integers(0, M, size=nnz, dtype=np.int32)          # unsorted columns, duplicates possible: legal CSR for the operator
ptr = ptr.astype(np.int32)
vals = synth.normal_f32(nnz, 5)
print(f"M={M} nnz={nnz} gen {time.time()-t:.1f}s", flush=True)
d_ptr, d_idx, d_val = (torch.from_numpy(a).to(dev) for a in (ptr, idx, vals))
d_B = torch.empty(M, N, device=dev); 
from hpc_amd.spmm import fill_normal
fill_normal(d_B.view(-1), seed=9)
d_C = torch.full((M, N), float("nan"), device=dev)
op = SpMMOpt(CSR(M, nnz, d_ptr, d_idx, d_val), N)
t = time.time(); op.preprocess(d_B, d_C); torch.cuda.synchronize(); print(f"preprocess {1e3*(time.time()-t):.1f} ms", flush=True)
op.run(d_B, d_C); torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(5): op.run(d_B, d_C)
b.record(); torch.cuda.synchronize()
ms = a.elapsed_time(b) / 5
model = synth.bytes_model(M, M, N, nnz)
print(f"run {ms:.3f} ms  {model['flops']/ms/1e6:.0f} GFLOP/s  {model['bytes_alg']/ms/1e6:.0f} GB/s gather model  wide={op.get_option('wide_addressing')}", flush=True)
assert not torch.isnan(d_C).any()
rows = np.unique(np.concatenate([[0, 1, M - 2, M - 1], np.random.Generator(np.random.Philox(key=[1, 1])).integers(0, M, 4096)]))
hB = d_B.cpu().numpy()
dd = np.diff(ptr)[rows]
sp = np.concatenate([[0], np.cumsum(dd)]).astype(np.int32)
take = np.concatenate([np.arange(ptr[r], ptr[r + 1]) for r in rows])
exp = oracle.spmm_omp(sp, idx[take], vals[take], hB)
got = d_C[rows.tolist()].cpu().numpy()
same = np.array_equal(got.view(np.uint32), exp.view(np.uint32))
print(f"{rows.size} sampled rows bitwise equal to the oracle: {same}")
assert same
