"""Hub rows kept as ONE exact segment: cost of the serial chain, 8 vs 32 gathers in flight."""
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
cases = {"c2": lambda: synth.csr_powerlaw(1 << 20, 32.0, 4096), "denseish": lambda: synth.csr_uniform(1 << 18, 300, 700), "rmat": lambda: synth.csr_rmat(20, 32),
         "am": lambda: synth.csr_powerlaw(881_680, 5_668_682 / 881_680, 154_828, seed=7, force_max=True)}
for name in sys.argv[1:] or list(cases):
    ptr, idx = cases[name]()
    M = ptr.size - 1
    vals = synth.make_values(idx.size)
    d = [torch.from_numpy(a).to(dev) for a in (ptr, idx, vals)]
    for N in (32, 64, 128, 256, 512):
        B = torch.randn(M, N, device=dev) * 0.1; C = torch.empty(M, N, device=dev)
        row = []
        for thr, deep in ((0, 0), (0, 16), (0, 32), (0, 0), (0, 16), (0, 32)):
            op = SpMMOpt(CSR(M, idx.size, *d), N)
            op.set_option("long_row_threshold", thr); op.set_option("segment_unroll", deep or 8)
            op.preprocess(B, C)
            row.append(f"thr={'auto' if thr == 0 else thr} deep={deep}: {timed(lambda: op.run(B, C)):.3f}")
        print(name, "N", N, "max", int(np.diff(ptr).max()), " | ".join(row), flush=True)
