"""C4 probe 2: what do the second runs (carried tiles, second pass) cost?  The C4 generator with every group's second run removed
(one launch, no carried tile) and with every group given a second run (twice the carried tiles), against C4 itself."""
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
orig = synth._block_dense_fast_params
for name in ("C4", "no second runs", "every group two runs"):
    def params(M_, rpb, K, seed, name=name):
        l1, l2, s1, s2 = orig(M_, rpb, K, seed)
        if name == "no second runs": l2 = np.zeros_like(l2)
        if name == "every group two runs": l2 = np.where(l2 == 0, 64, l2)
        return l1, l2, s1, s2
    synth._block_dense_fast_params = params
    d_ptr, d_idx = synth.csr_block_dense_fast_device(M, dev)
    nnz = int(d_idx.numel())
    d_val = torch.randn(nnz, device=dev) * 0.1
    B = torch.randn(M, N, device=dev) * 0.1; C = torch.empty(M, N, device=dev)
    op = SpMMOpt(CSR(M, nnz, d_ptr, d_idx, d_val), N)
    op.preprocess(B, C)
    t = timed(lambda: op.run(B, C))
    print(f"{name:22s}: nnz {nnz}, {t:.3f} ms, {2.0 * nnz * N / t / 1e9:.1f} TFLOP/s = {2.0 * nnz * N / t / 1e9 / 157.3:.3f} of the MFMA peak; "
          f"pieces {op.get_option('n_block_pieces')} items {op.get_option('n_block_items')} passes {op.get_option('n_block_passes')}; ns per MFMA-row-unit {t * 1e6 / (nnz / 16):.4f}", flush=True)
    del op, B, C, d_ptr, d_idx, d_val
