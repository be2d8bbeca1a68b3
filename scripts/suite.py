#!/usr/bin/env python3
"""Measure the hot path over the BASELINE.json configuration families in ONE process (GPU box).

    python scripts/suite.py [--quick] [--ref]  > gpurun_out/suite.jsonl

One JSON line per (structure, N): ours (ms, GFLOP/s, gather-model GB/s and fraction of 8 TB/s) and,
with --ref, the reference's own kernels compiled for gfx950 (oracle/_ref: the student's
SpmmOptKernel and the course's spmm_kernel_ref) as comparators on the same device and data.
Test/measurement infrastructure: the only place besides tests/ and bench.py that loads oracle/.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def timed(f, warm, reps):
    import torch

    for _ in range(warm):
        f()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(reps):
        f()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--ref", action="store_true")
    ap.add_argument("--vendor", action="store_true", help="also time rocSPARSE (the reference's headline comparator)")
    ap.add_argument("--M", type=int, default=1 << 20)
    ap.add_argument("--only", default=None)
    ap.add_argument("--opt", action="append", default=[])
    args = ap.parse_args()
    import torch
    from hpc_amd import CSR, SpMMOpt, synth
    from hpc_amd.spmm import count_bitdiff
    from oracle import oracle

    dev = torch.device("cuda:0")
    M = args.M
    structures = {
        "uniform32": lambda: synth.csr_uniform(M, 16, 48),
        "powerlaw32": lambda: synth.csr_powerlaw(M, 32.0, 4096),
        "blockdense": lambda: synth.csr_block_dense_fast(M),
        "rmat": lambda: synth.csr_rmat(int(np.log2(M)), 32),
        "banded": lambda: synth.csr_banded(M),
        "denseish": lambda: synth.csr_uniform(M // 4, 300, 700),
    }
    plan = [("uniform32", [32, 64, 128, 256, 512, 1024]), ("powerlaw32", [32, 128, 256]), ("blockdense", [32, 64, 128, 256, 512]),
            ("rmat", [32, 128, 256]), ("banded", [32, 128, 256]), ("denseish", [32, 128, 256])]
    if args.quick:
        plan = [("uniform32", [32, 128, 256]), ("powerlaw32", [128]), ("blockdense", [256])]
    for sname, Ns in plan:
        if args.only and args.only not in sname:
            continue
        t = time.time()
        ptr, idx = structures[sname]()
        vals = synth.make_values(idx.size)
        deg = np.diff(ptr)
        nnz = int(idx.size)
        M = ptr.size - 1
        d_ptr, d_idx, d_val = (torch.from_numpy(a).to(dev) for a in (ptr, idx, vals))
        gen_s = time.time() - t
        for N in Ns:
            d_B = torch.randn(M, N, device=dev) * 0.1
            d_C = torch.full((M, N), float("nan"), device=dev)
            op = SpMMOpt(CSR(M, nnz, d_ptr, d_idx, d_val), N)
            for kv in args.opt:
                k, v = kv.split("=")
                op.set_option(k, int(v))
            op.preprocess(d_B, d_C)
            ms = timed(lambda: op.run(d_B, d_C), 3, 10)
            model = synth.bytes_model(M, M, N, nnz)
            row = {"structure": sname, "M": M, "nnz": nnz, "deg_mean": round(float(deg.mean()), 2), "deg_max": int(deg.max()), "N": N,
                   "ours_ms": round(ms, 4), "ours_gflops": round(model["flops"] / ms / 1e6, 1),
                   "ours_GBs_alg": round(model["bytes_alg"] / ms / 1e6, 1),
                   # gather-model rate over 8 TB/s: a ROOFLINE fraction only where B is beyond the caches (uniform / power-law at M = 2^20); above 1 the
                   # structure lets L2 / the Infinity Cache serve the gathers (block-dense, R-MAT, banded, dense-ish) and the ratio is a reuse factor
                   "alg_rate_over_8TBs": round(model["bytes_alg"] / ms / 1e6 / 8000, 4),
                   "GBs_min_model": round(model["bytes_min"] / ms / 1e6, 1),
                   "n_hub_rows": op.get_option("n_hub_rows"), "hub_threshold": op.get_option("long_row_threshold"), "n_segments": op.get_option("n_chunks"),
                   "rows_out_of_stored_order": op.get_option("n_long_rows") if op.get_option("split_long_rows") else 0,
                   "lanes_per_row": op.get_option("lanes_per_row"), "preprocess_us": op.get_option("preprocess_us"), "gen_s": round(gen_s, 1)}
            if args.vendor:
                from hpc_amd.comparator import SpMMRocSparse
                from hpc_amd import valid as _valid
                d_V = torch.full((M, N), float("nan"), device=dev)
                try:
                    vs = SpMMRocSparse(CSR(M, nnz, d_ptr, d_idx, d_val), N)
                    vs.preprocess(d_B, d_V)
                    row["rocsparse_ms"] = round(timed(lambda: vs.run(d_B, d_V), 2, 5), 4)
                    row["speedup_vs_rocsparse"] = round(row["rocsparse_ms"] / ms, 2)
                    row["rocsparse_valid_bad"] = _valid(d_C, d_V, M * N)
                    row["rocsparse_buffer_bytes"] = vs.buffer_bytes()
                    del vs
                except Exception as e:  # comparator trouble must not hide our own numbers
                    row["rocsparse_error"] = str(e)[:100]
                del d_V
            if args.ref and oracle.ref_available():
                d_R = torch.zeros((M, N), device=dev)
                if N <= 1024:
                    ro = oracle.RefOpt(d_ptr, d_idx, d_val, M, N)

                    def run_opt():
                        d_R.zero_()          # the student kernel accumulates (spmm_opt.cu:34); zeroing is part of a correct call
                        ro.run(d_B, d_R)
                    row["ref_opt_ms_incl_memset"] = round(timed(run_opt, 1, 3), 4)
                    row["ref_opt_kernel_ms"] = round(timed(lambda: ro.run(d_B, d_R), 1, 3), 4)
                    d_R.zero_()
                    ro.run(d_B, d_R)
                    torch.cuda.synchronize()
                    from hpc_amd import valid
                    row["ref_opt_valid_bad_vs_ours"] = valid(d_R, d_C, M * N)
                if N <= 128 or args.quick is False:
                    t_ref = timed(lambda: oracle.ref_kernel_run(d_ptr, d_idx, d_val, d_B, d_R, M, N), 0, 1)
                    row["ref_kernel_ms"] = round(t_ref, 3)
                    nd, mx = count_bitdiff(d_C, d_R)
                    row["ref_kernel_bitdiff_vs_ours"] = nd
            print(json.dumps(row), flush=True)
            del d_B, d_C


if __name__ == "__main__":
    main()
