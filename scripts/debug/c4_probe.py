"""C4 probe: the block-dense structure with the columns confined to K' rows of B (same rows, same nonzeros per row, same
pieces per group): is the item kernel bound by the B traffic?  K' = 2^20 is C4 itself; 2^16 makes B 64 MiB (Infinity Cache),
2^12 makes it 4 MiB (one XCD's L2)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.getcwd())
from hpc_amd import CSR, SpMMOpt, synth
dev = torch.device("cuda:0")
def timed(f, warm=3, reps=20):
    for _ in range(warm): f()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(reps): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
M, N = 1 << 20, 256
for K in (1 << 20, 1 << 18, 1 << 16, 1 << 14, 1 << 12):
    d_ptr, d_idx = synth.csr_block_dense_fast_device(M, dev, K=K)
    nnz = int(d_idx.numel())
    d_val = torch.randn(nnz, device=dev) * 0.1
    B = torch.randn(K, N, device=dev) * 0.1; C = torch.empty(M, N, device=dev)
    op = SpMMOpt(CSR(M, nnz, d_ptr, d_idx, d_val), N, num_cols=K)
    op.preprocess(B, C)
    t = timed(lambda: op.run(B, C))
    print(f"K' = {K:8d}: B = {K * N * 4 / 2**20:7.1f} MiB, nnz {nnz}, {t:.3f} ms, {2.0 * nnz * N / t / 1e9:.1f} TFLOP/s = {2.0 * nnz * N / t / 1e9 / 157.3:.3f} of the MFMA peak; items {op.get_option('n_block_items')} shared {op.get_option('n_block_shared_items')} passes {op.get_option('n_block_passes')}", flush=True)
    del op, B, C, d_ptr, d_idx, d_val
