"""medium_row_threshold sweep (rows above it leave the rows kernel for the length-sorted segment kernel)."""
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
cases = {"c1": lambda: synth.csr_uniform(1 << 20, 16, 48), "c2": lambda: synth.csr_powerlaw(1 << 20, 32.0, 4096),
         "rmat": lambda: synth.csr_rmat(20, 32), "u64": lambda: synth.csr_uniform(1 << 19, 32, 96)}
for name in sys.argv[1:] or list(cases):
    ptr, idx = cases[name]()
    M = ptr.size - 1
    vals = synth.make_values(idx.size)
    d = [torch.from_numpy(a).to(dev) for a in (ptr, idx, vals)]
    for N in (32, 128, 256):
        B = torch.randn(M, N, device=dev) * 0.1; C = torch.empty(M, N, device=dev)
        row = []
        for _ in range(2):
            for mt in (16, 32, 48, 64, 96, 128):
                op = SpMMOpt(CSR(M, idx.size, *d), N)
                op.set_option("medium_row_threshold", mt)
                op.preprocess(B, C)
                row.append(f"{mt}: {timed(lambda: op.run(B, C)):.3f}")
            row.append("|")
        print(name, "N", N, " ".join(row), flush=True)
