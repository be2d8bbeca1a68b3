"""Does a handle lose its hub-stream overlap because OTHER handles (plain ones, graph ones, the vendor's) exist or existed in the process?
    python scripts/debug/coexist_probe.py"""
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
ptr, idx = synth.csr_dataset_shaped("youtube")
M = ptr.size - 1
vals = synth.make_values(idx.size)
d = [torch.from_numpy(a).to(dev) for a in (ptr, idx, vals)]
B = torch.randn(M, 32, device=dev) * 0.1; C = torch.empty(M, 32, device=dev)
def make(**o):
    op = SpMMOpt(CSR(M, idx.size, *d), 32)
    for k, v in o.items(): op.set_option(k, v)
    op.preprocess(B, C); return op
def t(op): return (round(timed(lambda: op.run(B, C)), 4), op.get_option("side_stream_overlaps"))
a = make(); print("plain A alone", t(a), flush=True)
b = make(); print("plain B while A lives", t(b), "| A again", t(a), flush=True)
c = make(); print("plain C while A, B live", t(c), flush=True)
del a, b
e = make(); print("plain E after A, B were destroyed (C lives)", t(e), "| C", t(c), flush=True)
del c, e
keep = [make() for _ in range(6)]
print("six plain handles alive:", [t(h) for h in keep], flush=True)
del keep
f = make(); print("plain F after all of them were destroyed", t(f), flush=True)
g = make(use_graph=1); print("graph G while F lives", t(g), "| F", t(f), flush=True)
h = make(); print("plain H while F, G live", t(h), flush=True)
del g
i = make(); print("plain I after G was destroyed (F, H live)", t(i), "| F", t(f), "| H", t(h), flush=True)
del f, h, i
j = make(); print("plain J after everything was destroyed", t(j), flush=True)
