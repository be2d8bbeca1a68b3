"""Kernel trace of the plain path's side-stream overlap after a graph handle has lived in the process (the context in which
scripts/report_table.py saw the plain path lose its overlap).  Run under rocprofv3 --kernel-trace; read Start/End/Queue_Id.
    python scripts/debug/overlap_trace.py [dataset] [N] [graph_first: 0/1]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hpc_amd import CSR, SpMMOpt, synth
dev = torch.device("cuda:0")
name = sys.argv[1] if len(sys.argv) > 1 else "arxiv"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 256
graph_first = int(sys.argv[3]) if len(sys.argv) > 3 else 1
ptr, idx = synth.csr_dataset_shaped(name)
M = ptr.size - 1
vals = synth.make_values(idx.size)
d = [torch.from_numpy(a).to(dev) for a in (ptr, idx, vals)]
def timed(f, warm=3, reps=10):
    for _ in range(warm): f()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(reps): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
if graph_first:
    B0 = torch.randn(M, 32, device=dev) * 0.1; C0 = torch.empty(M, 32, device=dev)
    p0 = SpMMOpt(CSR(M, idx.size, *d), 32); p0.preprocess(B0, C0); print("plain N=32", timed(lambda: p0.run(B0, C0)))
    g = SpMMOpt(CSR(M, idx.size, *d), 32); g.set_option("use_graph", 1); g.preprocess(B0, C0)
    print("graph N=32", timed(lambda: g.run(B0, C0)))
    del g, p0, B0, C0
B = torch.randn(M, N, device=dev) * 0.1; C = torch.empty(M, N, device=dev)
op = SpMMOpt(CSR(M, idx.size, *d), N); op.preprocess(B, C)
print(f"plain N={N}", timed(lambda: op.run(B, C)), "launches", op.get_option("n_launches"))
