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
for N in (256, 257, 260, 300, 384, 500, 512, 602, 1000, 1024, 1433):
    B = torch.randn(M, N, device=dev) * 0.1; C = torch.empty(M, N, device=dev)
    alg = 8 * idx.size + 4 * (M + 1) + 4 * N * idx.size + 4 * M * N
    out = []
    for sc in (0, 1):
        op = SpMMOpt(CSR(M, idx.size, *d), N); op.set_option("split_cols", sc); op.preprocess(B, C)
        ms = timed(lambda: op.run(B, C))
        out.append(f"split_cols={sc}: {ms:7.3f} ms {alg/ms/1e9:5.2f} TB/s launches={op.get_option('n_launches')}")
    print(f"N={N:4d}  " + "   ".join(out), flush=True)
    del B, C
