#!/usr/bin/env python3
"""Wide dense widths (N >= 256, BASELINE configs[3]'s one-GPU leg): column-tile width of the rows kernel, A/B in ONE process.

    python scripts/wide_n.py [--N 256 512 1024] [--tiles 256 128 64] [--rounds 3] [--one N TILE --steps K]

C1's CSR (M = K = 2^20, nnz = 33.6 M).  For every N, B is filled on the device, then the tile widths are timed
interleaved (round-robin, `rounds` times, 5 launches each; median reported) -- cdna_hip_programming.md rule 24.
Also: N = 128 columns gathered out of a B whose row pitch is 1024 floats (one 128-column strip of the 4 GiB B):
separates "bytes per gather" from "address footprint".
--one N TILE: a fixed configuration, K launches (for rocprofv3 --pmc passes).
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--N", type=int, nargs="*", default=[256, 512, 1024])
    ap.add_argument("--tiles", type=int, nargs="*", default=[256, 128, 64])
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--one", type=int, nargs=2, default=None)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--structure", default="uniform", help="uniform (C1's CSR) | powerlaw | rmat | banded | denseish")
    args = ap.parse_args()
    import torch
    from hpc_amd import CSR, SpMMOpt, synth
    from hpc_amd.spmm import fill_normal

    dev = torch.device("cuda:0")
    M = 1 << 20
    if args.structure == "uniform":
        ptr, idx = synth.csr_uniform(M, 16, 48)
    elif args.structure == "powerlaw":
        ptr, idx = synth.csr_powerlaw(M, 32.0, 4096)
    elif args.structure == "rmat":
        ptr, idx = synth.csr_rmat(20, 32)
    elif args.structure == "banded":
        ptr, idx = synth.csr_banded(M)
    elif args.structure == "denseish":
        M = 1 << 18
        ptr, idx = synth.csr_uniform(M, 300, 700)
    else:
        raise SystemExit("unknown structure")
    vals = synth.make_values(idx.size)
    nnz = int(idx.size)
    d_ptr, d_idx, d_val = (torch.from_numpy(a).to(dev) for a in (ptr, idx, vals))

    def timed(f, reps=5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            f()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / reps

    if args.one:
        N, tile = args.one
        d_B = torch.empty(M * N, dtype=torch.float32, device=dev)
        fill_normal(d_B, seed=125)
        d_C = torch.empty(M, N, dtype=torch.float32, device=dev)
        op = SpMMOpt(CSR(M, nnz, d_ptr, d_idx, d_val), N)
        op.set_option("tile_cols", tile)
        op.preprocess(d_B, d_C)
        for _ in range(args.steps):
            op.run(d_B, d_C)
        torch.cuda.synchronize()
        print(json.dumps({"one": [N, tile], "steps": args.steps, "lanes_per_row": op.get_option("lanes_per_row")}))
        return

    for N in args.N:
        d_B = torch.empty(M * N, dtype=torch.float32, device=dev)
        fill_normal(d_B, seed=125)
        d_C = torch.empty(M, N, dtype=torch.float32, device=dev)
        ops = {}
        for t in args.tiles:
            if t > N:
                continue
            op = SpMMOpt(CSR(M, nnz, d_ptr, d_idx, d_val), N)
            op.set_option("tile_cols", t)
            op.preprocess(d_B, d_C)
            op.run(d_B, d_C)
            ops[t] = op
        torch.cuda.synchronize()
        res = {t: [] for t in ops}
        for _ in range(args.rounds):
            for t, op in ops.items():
                op.run(d_B, d_C)
                res[t].append(timed(lambda: op.run(d_B, d_C)))
        model = synth.bytes_model(M, M, N, nnz)
        for t in ops:
            ms = float(np.median(res[t]))
            print(json.dumps({"structure": args.structure, "N": N, "tile_cols": t, "lanes_per_row": ops[t].get_option("lanes_per_row"), "ms_median": round(ms, 4),
                              "ms_all": [round(x, 4) for x in res[t]], "GBs_alg": round(model["bytes_alg"] / ms / 1e6, 1),
                              "frac_8TBs": round(model["bytes_alg"] / ms / 1e6 / 8000, 4)}), flush=True)
        if N == 1024 and args.structure == "uniform":
            # one 128-column strip of the wide B (row pitch 1024): N = 128's bytes, N = 1024's address footprint
            op = SpMMOpt(CSR(M, nnz, d_ptr, d_idx, d_val), 128)
            op.preprocess(d_B, d_C)
            f = lambda: op.run_ld(d_B, 1024, d_C, 1024)
            f()
            ms = float(np.median([timed(f) for _ in range(args.rounds)]))
            m128 = synth.bytes_model(M, M, 128, nnz)
            print(json.dumps({"N": 128, "ldb": 1024, "ms_median": round(ms, 4), "GBs_alg": round(m128["bytes_alg"] / ms / 1e6, 1),
                              "note": "one 128-column strip of the 4 GiB B"}), flush=True)
        del ops, d_B, d_C
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
