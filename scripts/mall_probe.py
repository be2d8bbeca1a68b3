"""Does an Infinity-Cache-resident B gather faster than an HBM-resident one?
Same rows / nnz / N as C1, columns confined to K = 2^k (B = K * N * 4 bytes)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.getcwd())
from hpc_amd import CSR, SpMMOpt, synth
dev = torch.device("cuda:0")
M, N = 1 << 20, int(sys.argv[1]) if len(sys.argv) > 1 else 128
def timed(f, warm=5, reps=20):
    for _ in range(warm): f()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(reps): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
for k in (20, 19, 18, 17, 16, 14):
    K = 1 << k
    ptr, idx = synth.csr_uniform(M, 16, 48, K=K)
    vals = synth.make_values(idx.size)
    d = [torch.from_numpy(a).to(dev) for a in (ptr, idx, vals)]
    B = torch.randn(K, N, device=dev) * 0.1; C = torch.empty(M, N, device=dev)
    op = SpMMOpt(CSR(M, idx.size, *d), N, num_cols=K); op.preprocess(B, C)
    ms = timed(lambda: op.run(B, C))
    alg = 8 * idx.size + 4 * (M + 1) + 4 * N * idx.size + 4 * M * N
    print(f"K=2^{k} B={K*N*4/2**20:.0f} MiB nnz={idx.size} {ms:.4f} ms  gather-model {alg/ms/1e6:.0f} GB/s", flush=True)
