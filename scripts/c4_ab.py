#!/usr/bin/env python3
"""Block (MFMA) path on BASELINE configs[4] (block-dense rows, M = 2^20, N = 256): interleaved A/B of the item
options in ONE process (cdna_hip_programming.md rule 24).

    python scripts/c4_ab.py [--N 256] [--rounds 4] [--one SHARE PIECES --steps K]

block_share  = most pieces per item (B rows staged once for all of them);
block_max_pieces = most runs a group's column list is cut into (= passes; 1 = the round-1 behaviour: one wave
walks the whole list of one group).  Every variant is checked bit for bit against the first one.
--one: a fixed configuration, K launches (for rocprofv3 passes).
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
    ap.add_argument("--N", type=int, default=256)
    ap.add_argument("--M", type=int, default=1 << 20)
    ap.add_argument("--rounds", type=int, default=4)
    ap.add_argument("--one", type=int, nargs=2, default=None)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--opt", action="append", default=[])
    ap.add_argument("--remap", action="store_true", help="XCD remap of the item list auto / off / on (identical configurations differ by up to 3 %% with the placement of their C buffers: read differences against that)")
    ap.add_argument("--sweep", action="store_true", help="items vs B-stationary sweeps (block_sweep; A/B library: MI_SPMM_LIB=hpc_amd/libmi_spmm_ablate.so) at several segment lengths / track floors")
    args = ap.parse_args()
    import torch
    from hpc_amd import CSR, SpMMOpt, synth
    from hpc_amd.spmm import count_bitdiff, fill_normal

    dev = torch.device("cuda:0")
    M, N = args.M, args.N
    ptr, idx = synth.csr_block_dense_fast(M)
    vals = synth.make_values(idx.size)
    nnz = int(idx.size)
    d_ptr, d_idx, d_val = (torch.from_numpy(a).to(dev) for a in (ptr, idx, vals))
    d_B = torch.empty(M * N, dtype=torch.float32, device=dev)
    fill_normal(d_B, seed=125)

    def make(share, pieces, extra=()):
        d_C = torch.full((M, N), float("nan"), dtype=torch.float32, device=dev)
        op = SpMMOpt(CSR(M, nnz, d_ptr, d_idx, d_val), N)
        op.set_option("block_share", share)
        op.set_option("block_max_pieces", pieces)
        for k, v in extra:
            op.set_option(k, v)
        for kv in args.opt:
            k, v = kv.split("=")
            op.set_option(k, int(v))
        op.preprocess(d_B, d_C)
        op.run(d_B, d_C)
        torch.cuda.synchronize()
        return op, d_C

    def timed(f, reps=10):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            f()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / reps

    if args.one:
        op, d_C = make(*args.one)
        ms = timed(lambda: op.run(d_B, d_C), args.steps)
        print(json.dumps({"one": args.one, "steps": args.steps, "n_launches": op.get_option("n_launches"), "ms": round(ms, 4)}))
        return

    if args.remap or args.sweep:
        if args.remap:
            combos = [(), (("xcd_remap", 0),), (("xcd_remap", 1),), (("xcd_remap", -1),)]
            extra_keys = ()
        else:
            combos = [()] + [(("block_sweep", 1), ("block_sweep_cols", c), ("block_sweep_min_tracks", t))
                             for c, t in ((2048, 4), (1024, 4), (4096, 4), (2048, 2), (512, 3), (8192, 4), (256, 3))]
            extra_keys = ("n_sweep_workgroups", "n_sweep_pieces", "n_sweep_trips", "n_block_residual_items")
        ops = {c: make(2, 4, c) for c in combos}
        ref = ops[()][1]
        res = {c: [] for c in combos}
        for _ in range(args.rounds):
            for c, (op, d_C) in ops.items():
                res[c].append(timed(lambda: op.run(d_B, d_C)))
        flops = 2.0 * nnz * N
        for c, (op, d_C) in ops.items():
            ms = float(np.median(res[c]))
            nd, _ = count_bitdiff(d_C, ref)
            line = {"options": dict(c), "ms_median": round(ms, 4), "ms_min": round(min(res[c]), 4), "TFLOPs": round(flops / ms / 1e9, 2),
                    "n_pieces": op.get_option("n_block_pieces"), "n_launches": op.get_option("n_launches"),
                    "preprocess_us": op.get_option("preprocess_us"), "bitdiff_vs_items": nd, "nan_left": bool(torch.isnan(d_C).any())}
            line.update({k: op.get_option(k) for k in extra_keys})
            print(json.dumps(line), flush=True)
        return

    variants = [(2, 4), (1, 4), (2, 1), (1, 1)]
    ops = {v: make(*v) for v in variants}
    ref = ops[variants[0]][1]
    res = {v: [] for v in variants}
    for _ in range(args.rounds):
        for v, (op, d_C) in ops.items():
            res[v].append(timed(lambda: op.run(d_B, d_C)))
    flops = 2.0 * nnz * N
    for v, (op, d_C) in ops.items():
        ms = float(np.median(res[v]))
        nd, _ = count_bitdiff(d_C, ref)
        print(json.dumps({"block_share": v[0], "block_max_pieces": v[1], "ms_median": round(ms, 4), "ms_min": round(min(res[v]), 4),
                          "TFLOPs": round(flops / ms / 1e9, 2), "frac_mfma_157": round(flops / ms / 1e9 / 157.3, 4),
                          "n_items": op.get_option("n_block_items"), "n_shared_items": op.get_option("n_block_shared_items"),
                          "n_pieces": op.get_option("n_block_pieces"), "n_passes": op.get_option("n_block_passes"),
                          "n_launches": op.get_option("n_launches"), "preprocess_us": op.get_option("preprocess_us"),
                          "bitdiff_vs_first": nd, "nan_left": bool(torch.isnan(d_C).any())}), flush=True)


if __name__ == "__main__":
    main()
