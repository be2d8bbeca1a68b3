"""Odd dense widths (N not a multiple of 4 -> dword path) and odd pitches on the C1 structure."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.getcwd())
from hpc_amd import CSR, SpMMOpt, synth
dev = torch.device("cuda:0")
def timed(f, warm=3, reps=10):
    for _ in range(warm): f()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(reps): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
M = 1 << 20
ptr, idx = synth.csr_uniform(M, 16, 48)
vals = synth.make_values(idx.size)
d = [torch.from_numpy(a).to(dev) for a in (ptr, idx, vals)]
for N in (30, 32, 33, 63, 64, 100, 127, 128, 129, 130, 132, 200, 250, 256, 260):
    B = torch.randn(M, N, device=dev) * 0.1; C = torch.empty(M, N, device=dev)
    op = SpMMOpt(CSR(M, idx.size, *d), N); op.preprocess(B, C)
    ms = timed(lambda: op.run(B, C))
    alg = 8 * idx.size + 4 * (M + 1) + 4 * N * idx.size + 4 * M * N
    print(f"N={N:4d} V={op.get_option('vector_width')} lpr={op.get_option('lanes_per_row'):2d} {ms:7.3f} ms  {alg/ms/1e9:6.2f} TB/s", flush=True)
