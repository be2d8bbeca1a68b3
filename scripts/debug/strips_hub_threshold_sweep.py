import sys, os
sys.path.insert(0, "/root/repo")
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np, torch
from hpc_amd import CSR, SpMMOpt, synth
dev = torch.device("cuda", 0)
for name in ("protein", "reddit.dgl"):
    ptr, idx = synth.csr_dataset_shaped(name)
    M, nnz = ptr.size - 1, idx.size
    d_ptr, d_idx = torch.from_numpy(ptr).to(dev), torch.from_numpy(idx).to(dev)
    d_val = torch.from_numpy(synth.make_values(nnz)).to(dev)
    for N in (32, 128):
        d_B = (torch.randn(M, N, device=dev) * 0.1).contiguous()
        ops = {}
        for thr in (0, 4096, 8192, 1 << 20):
            op = SpMMOpt(CSR(M, nnz, d_ptr, d_idx, d_val), N)
            op.set_option("long_row_threshold", thr)
            C = torch.empty(M, N, device=dev)
            op.preprocess(d_B, C)
            for _ in range(2): op.run(d_B, C)
            ops[thr] = (op, C)
        best = {t: 1e9 for t in ops}
        for rnd in range(3):
            for t, (op, C) in ops.items():
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize(); a.record()
                for _ in range(10): op.run(d_B, C)
                b.record(); torch.cuda.synchronize()
                best[t] = min(best[t], a.elapsed_time(b) / 10)
        for t, (op, C) in ops.items():
            print(f"{name} N {N} hub threshold {t} -> {op.get_option('long_row_threshold')}: hubs {op.get_option('n_hub_rows')} strips {op.get_option('n_col_strips')}  {best[t]:.3f} ms", flush=True)
        del ops, d_B
        torch.cuda.empty_cache()
