#!/usr/bin/env python3
"""Column tiles on small-K graphs: does a tile narrow enough to make K x tile x 4 bytes L2-sized pay (rows kernel, short rows)?
    python scripts/debug/tile_small_k_sweep.py arxiv collab yelp youtube --N 128 256"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from hpc_amd import CSR, SpMMOpt, synth

ap = argparse.ArgumentParser()
ap.add_argument("names", nargs="+")
ap.add_argument("--N", type=int, nargs="+", default=[128, 256])
args = ap.parse_args()
dev = torch.device("cuda", 0)
for name in args.names:
    if name == "c2":
        ptr, idx = synth.csr_powerlaw(1 << 20, 32.0, 4096)
    elif name.startswith("rmat"):
        ptr, idx = synth.csr_rmat(int(name[4:]), 32)
    else:
        ptr, idx = synth.csr_dataset_shaped(name)
    M, nnz = ptr.size - 1, idx.size
    d_ptr, d_idx = torch.from_numpy(ptr).to(dev), torch.from_numpy(idx).to(dev)
    d_val = torch.randn(nnz, device=dev) * 0.1
    print(name, M, nnz, flush=True)
    for N in args.N:
        d_B = (torch.randn(M, N, device=dev) * 0.1).contiguous()
        ops = {}
        for tile in (0, 32, 64, 128, 256):
            if tile > N and tile != 256:
                continue
            op = SpMMOpt(CSR(M, nnz, d_ptr, d_idx, d_val), N)
            op.set_option("tile_cols", tile)
            C = torch.empty(M, N, device=dev)
            op.preprocess(d_B, C)
            for _ in range(2): op.run(d_B, C)
            ops[tile] = (op, C)
        best = {t: 1e9 for t in ops}
        for rnd in range(3):
            for t, (op, C) in ops.items():
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize(); a.record()
                for _ in range(10): op.run(d_B, C)
                b.record(); torch.cuda.synchronize()
                best[t] = min(best[t], a.elapsed_time(b) / 10)
        for t, (op, C) in ops.items():
            print(f"  N {N:4d} tile_cols {t:3d} (lanes/row {op.get_option('lanes_per_row'):2d}; B tile {M * min(t or N, N) * 4 / 2**20:7.1f} MiB): {best[t]:8.4f} ms", flush=True)
        del ops, d_B
        torch.cuda.empty_cache()
