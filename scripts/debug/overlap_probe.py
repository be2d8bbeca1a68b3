"""Does the plain path's side-stream overlap (hub kernel beside the rows kernel) survive other handles / graph handles having
existed in the process?   python scripts/debug/overlap_probe.py [dataset] [N]      (GPU box)"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hpc_amd import CSR, SpMMOpt, synth
dev = torch.device("cuda:0")
name = sys.argv[1] if len(sys.argv) > 1 else "youtube"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 32
ptr, idx = synth.csr_dataset_shaped(name)
M = ptr.size - 1
vals = synth.make_values(idx.size)
d = [torch.from_numpy(a).to(dev) for a in (ptr, idx, vals)]
B = torch.randn(M, N, device=dev) * 0.1
C = torch.empty(M, N, device=dev)
def timed(f, warm=3, reps=10):
    for _ in range(warm): f()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(reps): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
def make(**opts):
    op = SpMMOpt(CSR(M, idx.size, *d), N)
    for k, v in opts.items(): op.set_option(k, v)
    op.preprocess(B, C)
    return op
def report(tag, op):
    print(f"{tag:60s} {timed(lambda: op.run(B, C)):.4f} ms   launches {op.get_option('n_launches')} hubs {op.get_option('n_hub_rows')} chunks {op.get_option('n_chunks')}", flush=True)
op = make(); report("fresh process, plain", op)
op0 = make(hub_overlap=0); report("plain, hub_overlap=0 (everything on the caller's stream)", op0); del op0
report("plain again (same handle as the first)", op)
del op
op = make(); report("new plain handle", op); del op
g = make(use_graph=1); report("graph handle", g)
op = make(); report("plain handle while a graph handle is alive", op)
del g
report("same plain handle after the graph handle was destroyed", op); del op
op = make(); report("new plain handle after the graph handle was destroyed", op); del op
for i in range(3):
    g = make(use_graph=1); g.run(B, C); torch.cuda.synchronize(); del g
op = make(); report("new plain handle after three more graph handles came and went", op)
g = make(use_graph=1); report("graph handle again", g)
# ---- what scripts/report_table.py has in the process besides our handles: the vendor comparator and the reference's kernels
del op, g
from hpc_amd.comparator import SpMMRocSparse
V = torch.empty(M, N, device=dev)
gcsr = CSR(M, idx.size, *d)
vend = SpMMRocSparse(gcsr, N); vend.preprocess(B, V)
print(f"{'rocSPARSE comparator':60s} {timed(lambda: vend.run(B, V)):.4f} ms", flush=True)
op = make(); report("plain handle while a rocSPARSE handle is alive", op)
g = make(use_graph=1); report("graph handle while a rocSPARSE handle is alive", g)
del vend
report("plain handle after the rocSPARSE handle was destroyed", op)
report("graph handle after the rocSPARSE handle was destroyed", g)
from oracle import oracle
if oracle.ref_available():
    ro = oracle.RefOpt(d[0], d[1], d[2], M, N)
    print(f"{'reference SpmmOptKernel':60s} {timed(lambda: ro.run(B, V), 1, 3):.4f} ms", flush=True)
    del ro
    report("plain handle after the reference kernel ran", op)
    report("graph handle after the reference kernel ran", g)
del op, g
op = make(); report("new plain handle at the end", op)
g = make(use_graph=1); report("new graph handle at the end", g)
s2 = torch.cuda.Stream(device=dev)
with torch.cuda.stream(s2):
    report("plain handle on a torch side stream (not the null stream)", op)
    report("graph handle on a torch side stream (not the null stream)", g)
