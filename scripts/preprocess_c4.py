import sys, time, numpy as np, torch
sys.path.insert(0, '.')
from hpc_amd import CSR, SpMMOpt, synth
dev = torch.device('cuda:0')
d_ptr, d_idx = synth.csr_block_dense_fast_device(1 << 20, dev)
nnz = int(d_idx.numel())
d_val = torch.zeros(nnz, device=dev)
for N in (128, 256):
    B = torch.zeros(1 << 20, N, device=dev); C = torch.empty_like(B)
    op = SpMMOpt(CSR(1 << 20, nnz, d_ptr, d_idx, d_val), N)
    for rep in range(4):
        torch.cuda.synchronize(); t = time.perf_counter(); op.preprocess(B, C); torch.cuda.synchronize(); dt = (time.perf_counter() - t) * 1e3
        print("C4 N", N, "total ms %.2f" % dt, {k: op.get_option(k) for k in ("pre_d2h_us", "pre_colcheck_us", "pre_detect_us", "pre_table_us", "pre_upload_us", "n_block_items")}, flush=True)
