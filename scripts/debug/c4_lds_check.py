"""C4 at full size: the item kernels (default) against "block_lds" = 1 -- whole-C bit comparison, the guard word, ms per step.
    python scripts/debug/c4_lds_check.py [M]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hpc_amd import CSR, SpMMOpt, synth
from hpc_amd.spmm import count_bitdiff, fill_normal
dev = torch.device("cuda:0")
M = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
N = 256
d_ptr, d_idx = synth.csr_block_dense_fast_device(M, dev)
nnz = int(d_idx.numel())
d_val = torch.empty(nnz, dtype=torch.float32, device=dev); fill_normal(d_val, synth.SEED_VALS)
B = torch.empty(M * N, dtype=torch.float32, device=dev); fill_normal(B, synth.SEED_B); B = B.view(M, N)
def timed(f, warm=3, reps=20):
    for _ in range(warm): f()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(reps): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
C0 = torch.full((M, N), float("nan"), device=dev)
op0 = SpMMOpt(CSR(M, nnz, d_ptr, d_idx, d_val), N); op0.preprocess(B, C0); op0.run(B, C0); torch.cuda.synchronize()
print("items:", round(timed(lambda: op0.run(B, C0)), 4), "ms", {k: op0.get_option(k) for k in ("n_block_items", "n_block_pieces", "n_block_passes", "n_launches")}, flush=True)
C1 = torch.full((M, N), float("nan"), device=dev)
op1 = SpMMOpt(CSR(M, nnz, d_ptr, d_idx, d_val), N); op1.set_option("block_lds", 1); op1.preprocess(B, C1)
print("lds workgroups:", op1.get_option("n_block_lds_workgroups"), flush=True)
op1.run(B, C1); torch.cuda.synchronize()
print("guard word after one step:", op1.get_option("block_lds_error"), "| NaN left:", int(torch.isnan(C1).sum()), "| bitdiff vs items:", count_bitdiff(C0, C1), flush=True)
if op1.get_option("block_lds_error") == 0:
    for _ in range(3):
        print("lds:  ", round(timed(lambda: op1.run(B, C1)), 4), "ms   items:", round(timed(lambda: op0.run(B, C0)), 4), "ms", flush=True)
    print("guard word:", op1.get_option("block_lds_error"), "| bitdiff:", count_bitdiff(C0, C1))
if os.environ.get("LDS_STAMPS") == "1":
    import numpy as np
    nwg = op1.get_option("n_block_lds_workgroups")
    dbg = torch.zeros((nwg + 8) * 64, dtype=torch.int64, device=dev)
    op1.set_option("block_lds_dbg_ptr", dbg.data_ptr())     # (does not touch the plan)
    op1.run(B, C1); torch.cuda.synchronize()
    d = dbg.cpu().numpy().reshape(-1, 8, 8)[:nwg + 2]
    ld = d[:, 4, :]                      # loader wave: start, after barrier, first publish, end, wait for slot, wait for vmcnt, stages, pieces
    ok = ld[:, 6] > 0
    ld = ld[ok]; dd = d[ok]
    for np_ in (1, 2, 3, 4):
        for nst in (4, 8):
            sel = (ld[:, 7] == np_) & (ld[:, 6] == nst)
            if sel.sum() == 0: continue
            L = ld[sel]; D_ = dd[sel]
            c0 = D_[:, 0, :]             # consumer 0: start, after barrier, loop entry, first stage there, loop end, poll wait, stores done, stages
            print(f"pieces {np_} stages {nst}: {int(sel.sum()):6d} wgs | loader: barrier {np.mean(L[:,1]-L[:,0]):7.0f} first publish {np.mean(L[:,2]-L[:,0]):7.0f} end {np.mean(L[:,3]-L[:,0]):7.0f} "
                  f"(slot wait {np.mean(L[:,4]):7.0f}, vmcnt wait {np.mean(L[:,5]):7.0f}) | consumer 0 ({np.mean(c0[:,7]):.1f} stages): stage 0 there {np.mean(c0[:,3]-c0[:,0]):7.0f} loop end {np.mean(c0[:,4]-c0[:,0]):7.0f} "
                  f"(poll wait {np.mean(c0[:,5]):7.0f}) stores done {np.mean(c0[:,6]-c0[:,0]):7.0f}")
