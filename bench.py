#!/usr/bin/env python3
"""bench.py -- the hot path's benchmark (driver contract: one JSON line from rank 0).

    python bench.py --gpus 1 --steps 20 --warmup 10
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one SpMM  C = A_csr * B  over the whole synthetic input, inputs resident in HBM.

  N = 1 : BASELINE.json configs[1] -- random CSR, M = K = 2^20, degree ~ U{16..48} (mean 32),
          dense N = 128, fp32 values / int32 indices.
  N > 1 : the column-sharded configuration (configs[3]) read as WEAK scaling: every GPU owns 128
          dense columns (N_total = 128 * N; N = 8 gives the north star's N = 1024), A replicated,
          B slice resident, RCCL all-gather of the C column blocks INSIDE the timed step, every rank
          ends with row-major C[M][128*N].  `--mode strong` keeps N_total = 1024 instead.

metric  = SpMM GFLOP/s, FLOPs == 2 * nnz * N_total.
roofline: dominant kernel = spmm_rows; achieved = algorithmic bytes (gather model
          8*nnz + 4*(M+1) + 4*N*nnz + 4*M*N per launch) / mean kernel duration from HIP events
          recorded on the launch stream around every timed launch; peak = 8.0 TB/s HBM3E spec.
cpu_baseline: the oracle's OpenMP restatement ("port") timed on this box's host cores on the same
          workload (rank 0, N = 1 only).  The oracle is used here as the baseline and checker only.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
COLS_PER_GPU = 128


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", default="C1", help="C1 (default), C2 power-law, C4 block-dense, C0 tiny")
    ap.add_argument("--M", type=int, default=None, help="override rows (down-sizing for rehearsal)")
    ap.add_argument("--N", type=int, default=None, help="override dense columns per GPU")
    ap.add_argument("--mode", default="weak", choices=["weak", "strong"])
    ap.add_argument("--panels", type=int, default=8)
    ap.add_argument("--exchange", default="allgather", choices=["auto", "allgather", "direct"],
                    help="N>1: how the C blocks travel.  allgather = RCCL's collective (default: the only schedule that could be "
                         "rehearsed over RCCL from a one-GPU box); direct = all-pairs grouped send/recv; auto = time both before "
                         "the warm-up and keep the faster")
    ap.add_argument("--opt", action="append", default=[], help="key=value handle option (repeatable)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-rows", type=int, default=None, help="rows of the workload the CPU baseline runs (default: all for C1)")
    ap.add_argument("--check", action="store_true", help="verify a row sample against the oracle after timing")
    ap.add_argument("--sweep", default=None, help="tuning sweep name: knobs")
    ap.add_argument("--rehearse-multi", action="store_true",
                    help="run the N>1 code path (RCCL group, panel pipeline, breakdown legs) with a world of 1 on one GPU")
    return ap.parse_args()


def dist_env(args):
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
        raise SystemExit(f"WORLD_SIZE={world} but --gpus {args.gpus}")
    return rank, world, local


def build_inputs(args, world, rank):
    from hpc_amd import synth

    if args.mode == "strong":
        n_total = args.N if args.N else 1024
        assert n_total % world == 0
        n_loc = n_total // world
    else:
        n_loc = args.N if args.N else COLS_PER_GPU
        n_total = n_loc * world
    name = args.config.upper()
    M = args.M
    if name == "C0":
        M = M or 1024
        ptr, idx = synth.csr_uniform(M, 0, 32)
        if not args.N and args.mode == "weak":
            n_loc, n_total = 32, 32 * world
    elif name == "C1":
        M = M or (1 << 20)
        ptr, idx = synth.csr_uniform(M, 16, 48)
    elif name == "C2":
        M = M or (1 << 20)
        ptr, idx = synth.csr_powerlaw(M, 32.0, 4096)
    elif name == "C4":
        M = M or (1 << 20)
        ptr, idx = synth.csr_block_dense_fast(M)
        if not args.N and args.mode == "weak":
            n_loc, n_total = 256, 256 * world
    elif name == "RMAT":
        M = M or (1 << 20)
        ptr, idx = synth.csr_rmat(int(np.log2(M)), 32)
    elif name == "DENSEISH":      # reddit/protein/ddi-like: hundreds of nonzeros in every row
        M = M or (1 << 18)
        ptr, idx = synth.csr_uniform(M, 300, 700)
    elif name == "BANDED":
        M = M or (1 << 20)
        ptr, idx = synth.csr_banded(M)
    else:
        raise SystemExit(f"unknown config {name}")
    vals = synth.make_values(idx.size)
    # this rank's column block of B: stream = global column-block index, so the union over ranks is
    # one well-defined K x N_total matrix whatever the world size
    B_loc = synth.normal_f32(M * n_loc, synth.SEED_B, stream=rank).reshape(M, n_loc)
    return name, M, n_loc, n_total, ptr, idx, vals, B_loc


def main():
    args = parse()
    rank, world, local = dist_env(args)
    import torch
    import torch.distributed as dist

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the product path has no CPU fallback")
    # MI_SPMM_SHARE_GPU=1: rehearsal of the N>1 launch line on a one-GPU box -- every rank uses cuda:0 and the
    # group is gloo (RCCL refuses two ranks on one device).  The JSON says so ("rehearsal"); never a result.
    share = os.environ.get("MI_SPMM_SHARE_GPU", "0") == "1" and world > 1
    if share:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    multi = world > 1 or args.rehearse_multi
    if multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        if share:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            # "nccl" IS RCCL on ROCm.  Its kernels run beside a compute kernel that fills every CU: give them the
            # high-priority queue so the exchange -- the link-bound part of the step -- is never the one waiting.
            pg_opts = None
            try:
                pg_opts = dist.ProcessGroupNCCL.Options()
                pg_opts.is_high_priority_stream = True
            except Exception:
                pg_opts = None
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev, pg_options=pg_opts)

    from hpc_amd import CSR, SpMMOpt, synth
    from hpc_amd.dist import ColumnShardedSpMM, ShardLayout
    from hpc_amd.spmm import unpack_gathered

    t_gen = time.time()
    name, M, n_loc, n_total, ptr, idx, vals, B_loc = build_inputs(args, world, rank)
    nnz = int(idx.size)
    t_gen = time.time() - t_gen
    d_ptr, d_idx, d_val, d_B = (torch.from_numpy(a).to(dev) for a in (ptr, idx, vals, B_loc))
    d_Cfull = torch.full((M, n_total), float("nan"), dtype=torch.float32, device=dev)
    d_Cloc = d_Cfull if not multi else torch.empty((M, n_loc), dtype=torch.float32, device=dev)

    op = SpMMOpt(CSR(M, nnz, d_ptr, d_idx, d_val), n_loc)
    for kv in args.opt:
        k, v = kv.split("=")
        op.set_option(k, int(v))
    t_pre = time.time()
    op.preprocess(d_B, d_Cloc)
    torch.cuda.synchronize()
    t_pre_first = time.time() - t_pre          # includes first-use code-object loading
    t_pre = time.time()
    op.preprocess(d_B, d_Cloc)
    torch.cuda.synchronize()
    t_pre = time.time() - t_pre

    sharded = ColumnShardedSpMM(op, ShardLayout(M, n_loc, world, rank), unpack_gathered, n_panels=args.panels,
                                force_collective=args.rehearse_multi,
                                exchange="allgather" if args.exchange == "auto" else args.exchange)
    if world > 1 and args.exchange == "auto":
        try:
            sharded.tune(d_Cloc)      # collective; untimed, before the warm-up
        except Exception as e:        # tuning is an optimisation: never lose the run over it
            print(f"[bench] exchange tuning failed on rank {rank}: {e!r}; using the library all-gather", file=sys.stderr, flush=True)
            sharded.exchange = "allgather"

    def barrier():
        if multi:
            dist.barrier()

    def step():
        sharded.run(d_B, d_Cloc, d_Cfull)

    if args.sweep and not multi:
        sweep(args, op, step, M, n_loc, nnz)
        return

    for _ in range(args.warmup):
        step()
    # timed region: EXACTLY K steps between barrier + synchronize on both sides
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        ev[i][0].record()          # on the stream the kernels are launched on (torch's current stream)
        step()
        ev[i][1].record()
    torch.cuda.synchronize()
    barrier()
    t1 = time.perf_counter()
    elapsed = t1 - t0
    if multi:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ms_per_step = elapsed * 1e3 / max(1, args.steps)
    dev_ms = [a.elapsed_time(b) for a, b in ev]
    dev_ms_mean = float(np.mean(dev_ms)) if dev_ms else float("nan")

    model = synth.bytes_model(M, M, n_loc, nnz)          # per launch = per GPU
    flops_total = 2.0 * nnz * n_total
    value = flops_total / (ms_per_step * 1e-3) / 1e9 if args.steps else float("nan")
    achieved = model["bytes_alg"] / (dev_ms_mean * 1e-3) / 1e9 if not multi else None

    check = None
    if args.check:
        from oracle import oracle
        g = np.random.Generator(np.random.Philox(key=[99, rank]))
        rows = np.unique(g.integers(0, M, 2048))
        deg = np.diff(ptr)[rows]
        sp = np.concatenate([[0], np.cumsum(deg)]).astype(np.int32)
        take = np.concatenate([np.arange(ptr[r], ptr[r + 1]) for r in rows])
        exp = oracle.spmm_omp(sp, idx[take], vals[take], B_loc)
        got = d_Cfull[rows.tolist()][:, rank * n_loc:(rank + 1) * n_loc].cpu().numpy()
        thr = op.get_option("long_row_threshold")  # rows above it are split: compared by tolerance elsewhere
        short = deg <= thr
        check = {"rows": int(rows.size), "bitwise_equal_rows": int((got.view(np.uint32) == exp.view(np.uint32)).all(axis=1).sum()),
                 "short_rows_all_equal": bool((got.view(np.uint32)[short] == exp.view(np.uint32)[short]).all())}

    # N > 1: compute-only and exchange-only legs, outside the timed region (report
    # compute scaling and end-to-end scaling separately; the step is bound by the all-gather)
    breakdown = None
    if multi:
      try:
        def timed_ms(f, reps=5):
            f()
            barrier()
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(reps):
                f()
            b.record()
            torch.cuda.synchronize()
            t = torch.tensor([a.elapsed_time(b) / reps], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return float(t.item())

        compute_ms = timed_ms(lambda: op.run_rows(d_B, n_loc, d_Cloc, n_loc, 0, M))
        stage = torch.empty(world * M * n_loc, dtype=torch.float32, device=dev)

        def exchange():
            sharded._exchange(stage, d_Cloc.view(-1), M)
            unpack_gathered(stage, d_Cfull, M, world, n_loc, n_total)

        exchange_ms = timed_ms(exchange, reps=3)
        del stage
        breakdown = {"compute_only_ms": round(compute_ms, 4), "allgather_plus_unpack_only_ms": round(exchange_ms, 4),
                     "bytes_received_per_gpu": int((world - 1) * M * n_loc * 4),
                     "compute_only_gflops_total": round(flops_total / (compute_ms * 1e-3) / 1e9, 1),
                     "exchange": sharded.exchange,
                     "exchange_tuning_ms_first_panel": sharded.tuning}
      except Exception as e:   # the breakdown is a courtesy: it must never cost the contract line
        breakdown = {"error": repr(e)[:200]}

    roof_ms = dev_ms_mean
    if multi and breakdown and "compute_only_ms" in breakdown:
        # N>1: the dominant kernel is the same per-GPU launch; its duration is the compute-only leg
        # (one launch over all rows, HIP events on the launch stream, max over ranks)
        roof_ms = breakdown["compute_only_ms"]
        achieved = model["bytes_alg"] / (roof_ms * 1e-3) / 1e9

    cpu = None
    if rank == 0 and not multi and not args.no_cpu_baseline:
        cpu = cpu_baseline(args, ptr, idx, vals, B_loc, M, n_loc)

    if rank == 0:
        traffic = None
        tp = os.path.join(ROOT, "profiles", "traffic_latest.json")
        if not multi and name == "C1" and os.path.exists(tp):
            try:
                traffic = json.load(open(tp)).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        line = {
            "metric": "spmm_gflops", "value": round(value, 2), "unit": "GFLOP/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
            "scaling": args.mode, "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {
                "workload": f"{name}: CSR SpMM M=K={M}, nnz={nnz} (deg mean {nnz / max(1, M):.1f}, max {int(np.diff(ptr).max()) if M else 0}), "
                            f"N={n_total} fp32 ({n_loc} columns per GPU), int32 indices",
                "M": M, "K": M, "nnz": nnz, "N": n_total, "cols_per_gpu": n_loc,
                "parallelism": "single GPU" if world == 1 else f"column-sharded x{world}, RCCL all-gather of C blocks ({sharded.exchange} schedule), {args.panels} row panels",
                "options": {k: op.get_option(k) for k in ("kernel", "rows_per_block", "block_threads", "xcd_remap", "nt_store", "nt_stream",
                                                          "medium_row_threshold", "long_row_threshold", "long_row_chunk", "segment_unroll", "n_long_rows", "n_chunks",
                                                          "lanes_per_row", "vector_width", "n_launches")},
                "preprocess_ms": round(t_pre * 1e3, 2), "preprocess_first_call_ms": round(t_pre_first * 1e3, 2), "input_gen_s": round(t_gen, 1),
            },
            "device_ms_per_step": round(dev_ms_mean, 4),
            "roofline": ({
                "bound": "hbm", "kernel": "mi::spmm_rows_v2" if op.get_option("kernel") == 2 else "mi::spmm_rows", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                "bytes_alg_per_launch": model["bytes_alg"], "bytes_min_per_launch": model["bytes_min"],
                "kernel_ms": round(roof_ms, 4),
                "frac_of_measured_copy_6290": round(achieved / 6290.0, 4),
            } if achieved is not None else None),
            "cpu_baseline": cpu,
        }
        if share:
            line["rehearsal"] = f"{world} ranks sharing one GPU over gloo (MI_SPMM_SHARE_GPU=1): launch-line rehearsal, not a result"
        if check is not None:
            line["check"] = check
        if breakdown is not None:
            line["multi_gpu_breakdown"] = breakdown
        print(json.dumps(line), flush=True)
    if multi:
        dist.destroy_process_group()


def cpu_baseline(args, ptr, idx, vals, B, M, N):
    """Oracle OpenMP restatement on the host cores: 1 warm-up + 3 timed runs,
    on a bounded row prefix of the same workload sized for ~10-30 s of CPU work."""
    from oracle import oracle

    rows = args.cpu_rows if args.cpu_rows else M
    rows = min(rows, M)
    out = np.empty((rows, N), dtype=np.float32)
    sub_ptr = ptr[: rows + 1]
    nnz = int(sub_ptr[-1])
    t = time.perf_counter()
    oracle.spmm_omp(sub_ptr, idx, vals, B, out=out)          # warm-up (and first-touch of out)
    first = time.perf_counter() - t
    reps = 5 if first < 4 else (3 if first < 8 else 1)
    times = []
    for _ in range(reps):
        t = time.perf_counter()
        oracle.spmm_omp(sub_ptr, idx, vals, B, out=out)
        times.append(time.perf_counter() - t)
    best = float(np.mean(times))
    return {
        "value": round(2.0 * nnz * N / best / 1e9, 3), "unit": "GFLOP/s", "cores": oracle.num_threads(), "kind": "port",
        "sample": f"rows [0,{rows}) of the same workload ({nnz} nnz, N={N}), mean of {reps} runs after 1 warm-up, "
                  f"{best * 1e3:.1f} ms each; oracle/spmm_oracle.c oracle_spmm_omp, gcc -O3 -march=x86-64-v3 -fopenmp",
        "seconds": round(best, 4), "host_cpus": os.cpu_count(),
    }


def sweep(args, op, step, M, N, nnz):
    """One-process interleaved A/B over the handle's knobs (cdna_hip_programming.md rule 24)."""
    import itertools
    import torch
    from hpc_amd import synth

    model = synth.bytes_model(M, M, N, nnz)
    grids = {
        "knobs": dict(kernel=[2], nt_store=[0, 1], nt_stream=[0, 1], xcd_remap=[0, 1], rows_per_block=[0]),
        "wide": dict(kernel=[1, 2], nt_store=[1], nt_stream=[0], xcd_remap=[0, 1], rpg=[1, 2, 8]),
        "bt": dict(kernel=[2], nt_store=[1], nt_stream=[0], xcd_remap=[0, 1], block_threads=[64, 128, 256], rpg=[1, 2]),
        "rpg": dict(kernel=[2], nt_store=[1], nt_stream=[0, 1], xcd_remap=[1], rpg=[1, 2, 3, 4, 8, 16]),
        "v2": dict(kernel=[1, 2], nt_store=[0, 1], nt_stream=[0, 1], xcd_remap=[0, 1], rpg=[2, 4, 7]),
    }[args.sweep]
    keys = list(grids)
    combos = list(itertools.product(*[grids[k] for k in keys]))
    results = {c: [] for c in combos}
    rounds = 3
    for r in range(rounds):
        for c in combos:
            for k, v in zip(keys, c):
                if k == "rpg":   # rows per lane group -> rows per block
                    bt = dict(zip(keys, c)).get("block_threads", 256)
                    op.set_option("rows_per_block", v * max(1, bt // max(8, min(64, N // 4))))
                else:
                    op.set_option(k, v)
            for _ in range(2):
                step()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(5):
                step()
            b.record()
            torch.cuda.synchronize()
            results[c].append(a.elapsed_time(b) / 5)
    rows = []
    for c in combos:
        ms = float(np.median(results[c]))
        rows.append((ms, c))
    rows.sort()
    for ms, c in rows:
        print(json.dumps({"sweep": dict(zip(keys, c)), "ms_median": round(ms, 4), "ms_min": round(min(results[c]), 4),
                          "GBs_alg": round(model["bytes_alg"] / ms / 1e6, 1), "GFLOPs": round(2.0 * nnz * N / ms / 1e6, 1)}), flush=True)


if __name__ == "__main__":
    main()
