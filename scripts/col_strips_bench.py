#!/usr/bin/env python3
"""Column strips of the exact segments ("col_strips") on the dense-ish dataset shapes: strips off / auto / forced counts,
interleaved in one process, every C compared bit for bit with the strips-off C.

    python scripts/col_strips_bench.py protein reddit ddi ppa --N 32 128 256 [--S 1 0 2 4 8]"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from hpc_amd import CSR, SpMMOpt, synth
    from hpc_amd.spmm import count_bitdiff

    ap = argparse.ArgumentParser()
    ap.add_argument("names", nargs="+")
    ap.add_argument("--N", type=int, nargs="+", default=[32, 128, 256])
    ap.add_argument("--S", type=int, nargs="+", default=[1, 0, 2, 4, 8, 16])
    ap.add_argument("--reps", type=int, default=10)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    for name in args.names:
        key = name if name in synth.DATASET_SHAPES else name + ".dgl"
        ptr, idx = synth.csr_dataset_shaped(key)
        M, nnz = ptr.size - 1, idx.size
        d_ptr, d_idx = torch.from_numpy(ptr).to(dev), torch.from_numpy(idx).to(dev)
        d_val = torch.from_numpy(synth.make_values(nnz)).to(dev)
        print(f"{key}: M {M}, nnz {nnz}, mean degree {nnz / M:.1f}", flush=True)
        for N in args.N:
            d_B = (torch.randn(M, N, device=dev) * 0.1).contiguous()
            ops, Cs = {}, {}
            for S in args.S:
                op = SpMMOpt(CSR(M, nnz, d_ptr, d_idx, d_val), N)
                op.set_option("col_strips", S)
                C = torch.full((M, N), float("nan"), device=dev)
                op.preprocess(d_B, C)
                for _ in range(2):
                    op.run(d_B, C)
                ops[S], Cs[S] = op, C
            torch.cuda.synchronize()
            best = {S: 1e9 for S in args.S}
            for rnd in range(3):                      # interleaved rounds
                for S in args.S:
                    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    a.record()
                    for _ in range(args.reps):
                        ops[S].run(d_B, Cs[S])
                    b.record()
                    torch.cuda.synchronize()
                    best[S] = min(best[S], a.elapsed_time(b) / args.reps)
            base = best[args.S[0]]
            for S in args.S:
                nd = count_bitdiff(Cs[S], Cs[args.S[0]])[0] if S != args.S[0] else 0
                o = ops[S]
                print(f"  N {N:4d}  col_strips {S:2d} -> {o.get_option('n_col_strips'):2d} strips  {best[S]:8.3f} ms  ({base / best[S]:4.2f}x)  "
                      f"launches {o.get_option('n_launches')}  segments {o.get_option('n_medium_rows')}  hubs {o.get_option('n_hub_rows')}  "
                      f"hub threshold {o.get_option('long_row_threshold')}  lanes/row {o.get_option('lanes_per_row')}  bits differing from strips off: {nd}", flush=True)
            del ops, Cs, d_B
            torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
