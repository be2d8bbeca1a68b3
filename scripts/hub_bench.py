"""Hub rows: stored order through the hub kernel (default) against the opt-in split mode and against no hub path at all.
    python scripts/hub_bench.py [case ...]        (GPU box)
"""
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
def shaped(name, M, nnz_t, mx):
    return lambda: synth.csr_powerlaw(M, nnz_t / M, min(mx, M), seed=sum(map(ord, name)) % 1000 + 1, force_max=True)
cases = {"c2": lambda: synth.csr_powerlaw(1 << 20, 32.0, 4096), "rmat": lambda: synth.csr_rmat(20, 32),
         "am": shaped("am", 881_680, 5_668_682, 154_828), "arxiv": shaped("arxiv", 169_343, 1_166_243, 13_155),
         "youtube": shaped("youtube", 1_138_499, 5_980_886, 28_754), "ddi": shaped("ddi", 4_267, 2_135_822, 2_234),
         "reddit": shaped("reddit.dgl", 232_965, 114_615_892, 21_657), "protein": shaped("protein", 132_534, 79_122_504, 7_750)}
variants = [("hub", {}), ("hubser", {"hub_overlap": 0}), ("hub16", {"hub_slice": 16}), ("hub32", {"hub_slice": 32}), ("hub64", {"hub_slice": 64}), ("split", {"split_long_rows": 1}), ("nohub", {"long_row_threshold": 1 << 30})]
for thr in (256, 512, 1024, 2048, 4096, 8192):
    variants.append((f"hubt{thr}", {"long_row_threshold": thr}))
if os.environ.get("HUB_VARIANTS"):
    variants = [v for v in variants if v[0] in os.environ["HUB_VARIANTS"].split(",")]
for name in sys.argv[1:] or list(cases):
    ptr, idx = cases[name]()
    M = ptr.size - 1
    vals = synth.make_values(idx.size)
    d = [torch.from_numpy(a).to(dev) for a in (ptr, idx, vals)]
    deg = np.diff(ptr)
    for N in (32, 128, 256):
        B = torch.randn(M, N, device=dev) * 0.1; C = torch.empty(M, N, device=dev)
        row = []
        ref = None
        for vn, opts in variants:
            if vn == "nohub" and deg.max() > 70000 and N > 32: continue
            op = SpMMOpt(CSR(M, idx.size, *d), N)
            for k, v in opts.items(): op.set_option(k, v)
            op.preprocess(B, C)
            t = timed(lambda: op.run(B, C))
            extra = ""
            if vn == "hub": ref = C.clone(); extra = f" (hubs {op.get_option('n_hub_rows')}, thr {op.get_option('long_row_threshold')})"
            elif vn != "split" and ref is not None: extra = " ==" if torch.equal(ref.view(torch.int32), C.view(torch.int32)) else " !!DIFF"
            row.append(f"{vn} {t:.3f}{extra}")
            del op
        print(name, "N", N, "M", M, "nnz", idx.size, "max", int(deg.max()), "|", " | ".join(row), flush=True)
        del B, C
