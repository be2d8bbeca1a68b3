"""Side-stream priorities (hub / segment kernels) x process context (fresh, or after a graph handle has lived): ms per step.
    python scripts/debug/priority_probe.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hpc_amd import CSR, SpMMOpt, synth
dev = torch.device("cuda:0")
def timed(f, warm=3, reps=10):
    for _ in range(warm): f()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(reps): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
cases = [("arxiv", 256), ("youtube", 32), ("am", 256), ("am", 32), ("yelp", 32), ("products", 32), ("rmat", 128), ("c2", 128)]
data = {}
for name, N in cases:
    if name not in data:
        ptr, idx = synth.csr_rmat(20, 32) if name == "rmat" else synth.csr_powerlaw(1 << 20, 32.0, 4096) if name == "c2" else synth.csr_dataset_shaped(name)
        vals = synth.make_values(idx.size)
        data[name] = (ptr.size - 1, idx.size, [torch.from_numpy(a).to(dev) for a in (ptr, idx, vals)])
def run_all(tag):
    for name, N in cases:
        M, nnz, d = data[name]
        B = torch.randn(M, N, device=dev) * 0.1; C = torch.empty(M, N, device=dev)
        row = []
        for pr in (3, 3, 3):
            op = SpMMOpt(CSR(M, nnz, *d), N); op.set_option("side_priority", pr); op.preprocess(B, C)
            row.append(f"prio {pr}: {timed(lambda: op.run(B, C)):.4f}")
            del op
        op = SpMMOpt(CSR(M, nnz, *d), N); op.set_option("hub_overlap", 0); op.preprocess(B, C)
        row.append(f"no side streams: {timed(lambda: op.run(B, C)):.4f}")
        del op
        for rep in range(3):
            op = SpMMOpt(CSR(M, nnz, *d), N); op.set_option("segment_overlap", 0); op.preprocess(B, C)
            row.append(f"hub only on side: {timed(lambda: op.run(B, C)):.4f}")
            del op
        print(f"{tag:28s} {name:9s} N={N:<4d} | " + " | ".join(row), flush=True)
        del B, C
run_all("fresh")
M, nnz, d = data["arxiv"]
B = torch.randn(M, 32, device=dev) * 0.1; C = torch.empty(M, 32, device=dev)
g = SpMMOpt(CSR(M, nnz, *d), 32); g.set_option("use_graph", 1); g.preprocess(B, C); timed(lambda: g.run(B, C)); del g, B, C
run_all("after a graph handle lived")
