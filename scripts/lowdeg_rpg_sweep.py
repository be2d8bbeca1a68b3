import sys, os, json
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
for name, M, nnz_t, mx in (("citation", 2_927_963, 30_387_995, 1738), ("wikikg2", 2_500_604, 16_109_182, 911), ("youtube", 1_138_499, 5_980_886, 28_754), ("collab", 235_868, 2_358_104, 671), ("yelp", 716_847, 13_954_819, 4886)):
    ptr, idx = synth.csr_powerlaw(M, nnz_t / M, mx, seed=7, force_max=True)
    vals = synth.make_values(idx.size)
    d = [torch.from_numpy(a).to(dev) for a in (ptr, idx, vals)]
    for N in (32, 128, 256):
        B = torch.randn(M, N, device=dev) * 0.1; C = torch.empty(M, N, device=dev)
        op = SpMMOpt(CSR(M, idx.size, *d), N); op.preprocess(B, C)
        gpb = 256 // max(8, min(64, N // 4))
        res = {}
        for rnd in range(2):
            for rpg in (1, 2, 3, 4, 7):
                op.set_option("rows_per_block", rpg * gpb)
                res.setdefault(rpg, []).append(timed(lambda: op.run(B, C)))
        print(name, "N", N, "mean deg %.1f" % (idx.size / M), {k: round(min(v), 4) for k, v in res.items()}, flush=True)
