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
          B slice resident, exchange of the C column blocks over RCCL / xGMI INSIDE the timed step
          (include/mi_spmm_dist.h: libmi_spmm_dist.so), every rank ends with row-major C[M][128*N].
          Schedule of the exchange (`--exchange auto`, the default): SAFE FIRST -- warm-up + K timed steps on the RCCL
          all-gather give a complete contract line that is kept in hand; then `direct`, `peer2d` and `peer_store` are each
          set up and timed for two steps under a per-rank watchdog (success and times agreed collectively); if one is
          faster the K steps are timed again on it and THAT line is printed, with the all-gather's figures beside it
          (`exchange_selection`).  A candidate that neither finishes nor throws costs the candidates, not the measurement:
          the watchdog prints the line in hand (`exchange_watchdog`) and the process leaves with os._exit.
          `--mode strong` keeps N_total = 1024 instead.  The line carries `strong_reference_ms`:
          ONE GPU computing all N_total columns, so the 8-GPU-vs-1-GPU ratio at N = 1024 is in the record.

metric  = SpMM GFLOP/s, FLOPs == 2 * nnz * N_total.
roofline: C1/C2/...: dominant kernel = spmm_rows_v2; achieved = algorithmic bytes (gather model
          8*nnz + 4*(M+1) + 4*N*nnz + 4*M*N per launch) / mean step duration from HIP events
          recorded on the launch stream around every timed step; peak = 8.0 TB/s HBM3E spec.  The
          rate is an ALGORITHMIC rate: the counters behind `traffic` sit on the L2's fabric side and
          include Infinity-Cache hits, so it can exceed what HBM alone delivers (6.29 TB/s copy).
          C4 (block-dense): dominant kernel = spmm_block_items on the f32 MFMA; bound "mfma",
          achieved = 2*nnz*N / step, peak 157.3 TFLOP/s, with the byte models beside it.
also:     N = 1 and the default configuration only: the other single-GPU BASELINE configurations -- C2 (power-law rows),
          C4 (block-dense rows, N = 256, MFMA path) and C1's CSR at N = 1024 (configs[3]'s one-GPU leg) -- each timed with
          the same HIP-event protocol AFTER the headline's timed region (so the driver's clock brackets them too); inputs
          regenerated from the seeds (structure) and filled on the device (values, B).  {config, ms_per_step, value, roofline}.
cpu_baseline: the oracle's OpenMP restatement ("port") timed on this box's host cores on the same
          workload (rank 0, N = 1 only), all cores and one thread.  Baseline and checker only.
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
HUB_CLOCK_HZ = 2.4e9         # MI355X peak engine clock: cycles of the `chain` bound (also.AM32)
COLS_PER_GPU = 128

# The contract: rank 0 prints ONE JSON line on stdout.  Libraries write there too (RCCL prints a five-line version banner to stdout at
# communicator set-up on this image), so main() points file descriptor 1 at stderr for the whole run and the line goes out through
# a private duplicate of the original stdout: whatever else is printed, stdout carries the line and nothing else.
_LINE_FD = None


def claim_stdout():
    global _LINE_FD
    if _LINE_FD is None:
        sys.stdout.flush()
        _LINE_FD = os.dup(1)
        os.dup2(2, 1)


def emit_line(line):
    data = (json.dumps(line) + "\n").encode()
    fd = _LINE_FD if _LINE_FD is not None else 1
    sys.stdout.flush()
    while data:
        n = os.write(fd, data)
        data = data[n:]


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
    ap.add_argument("--exchange", default="auto", choices=["auto", "allgather", "direct", "peer2d", "peer_store"],
                    help="N>1: how the C blocks travel (include/mi_spmm_dist.h).  allgather = RCCL's collective into staging + "
                         "re-layout kernel (default); direct = all-pairs grouped ncclSend/ncclRecv into the same staging; peer2d = "
                         "strided 2-D copies straight into the peers' C (HIP IPC), no staging, no re-layout; peer_store = the kernels' epilogues "
                         "store every result into the local C and into every peer's (no copies at all); auto (default) = safe-first: "
                         "the contract line is measured on allgather and kept in hand, then direct / peer2d / peer_store are tried under a "
                         "per-rank watchdog (ranks agree collectively) and the line is re-measured on one of them only if it is faster")
    ap.add_argument("--opt", action="append", default=[], help="key=value handle option (repeatable)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-also", action="store_true", help="N=1, C1: skip the `also` object (C2, C4 and C1 at N=1024, timed after the headline)")
    ap.add_argument("--no-strong-reference", action="store_true", help="N>1: skip the one-GPU-all-columns reference timing")
    ap.add_argument("--no-link-probe", action="store_true", help="N>1: skip the link probe (peer copies after the timed region: multi_gpu_breakdown.links)")
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
    elif name == "LONGROWS":      # the `also` entry LONG_ROWS as a configuration of its own (profiles): built on the device, copied back
        import torch

        M = M or (1 << 17)
        d_p, d_i = synth.csr_long_rows_device(M, torch.device("cuda", torch.cuda.current_device()))
        ptr, idx = d_p.cpu().numpy(), d_i.cpu().numpy()
        del d_p, d_i
    elif name == "BANDED":
        M = M or (1 << 20)
        ptr, idx = synth.csr_banded(M)
    elif name.lower() in synth.DATASET_SHAPES or name.lower() + ".dgl" in synth.DATASET_SHAPES:
        # the shape of one of the reference's 13 course graphs (rows, nonzeros, longest row; synth.DATASET_SHAPES): `--config am`
        key = name.lower() if name.lower() in synth.DATASET_SHAPES else name.lower() + ".dgl"
        if M:
            raise SystemExit("dataset-shaped configurations have their own row count")
        ptr, idx = synth.csr_dataset_shaped(key)
        M = int(ptr.size - 1)
        name = key
    else:
        raise SystemExit(f"unknown config {name}")
    vals = synth.make_values(idx.size)
    # this rank's column block of B: stream = global column-block index, so the union over ranks is
    # one well-defined K x N_total matrix whatever the world size
    B_loc = synth.normal_f32(M * n_loc, synth.SEED_B, stream=rank).reshape(M, n_loc)
    return name, M, n_loc, n_total, ptr, idx, vals, B_loc


class Watchdog:
    """Per-rank guard around the parts of an N > 1 run that may hang instead of throwing (a schedule that deadlocks on first
    contact with real xGMI links).  `arm(label, seconds)` ... `disarm()`: if the armed section neither finishes nor throws
    within its bound, rank 0 prints the contract line it has IN HAND (a complete measurement of the safe schedule, with
    `exchange_watchdog` saying what hung) and every rank leaves with os._exit -- a decision to stop, never a re-exec of a
    process that has initialised the GPU, and never a retry.  Exit code: MI_SPMM_WATCHDOG_EXIT (default 0: the line in hand
    is a complete, valid measurement of the safe schedule and a non-zero code would make a driver discard it with the run; the stop is
    reported at the top level of that line -- "watchdog_fired": <section> -- and on stderr with every thread's traceback; set the variable
    to make the same stop read as a failure.  Non-zero (3) whenever no line is in hand yet)."""

    def __init__(self, rank):
        import threading

        self.rank = rank
        self.t0 = time.monotonic()
        self.lock = threading.Lock()
        self.deadline, self.label, self.line, self.measured, self.printed = None, None, None, False, False
        self.thread = threading.Thread(target=self._loop, daemon=True, name="bench-watchdog")
        self.thread.start()

    def set_line(self, line):
        """rank 0: the contract line as it stands (None once the final line has been printed by the main thread)."""
        with self.lock:
            self.line = dict(line) if line is not None else None
            self.printed = self.printed or (line is None and self.measured)

    def mark_measured(self):
        """every rank: the safe schedule's contract measurement is complete (rank 0 holds its line)."""
        with self.lock:
            self.measured = True

    def arm(self, label, seconds):
        # one stderr line per guarded section and rank: the record of a run that hangs says how far each rank got
        print(f"[bench] rank {self.rank}: +{time.monotonic() - self.t0:.1f}s {label}", file=sys.stderr, flush=True)
        with self.lock:
            self.label, self.deadline = label, time.monotonic() + float(seconds)

    def disarm(self):
        with self.lock:
            self.label, self.deadline = None, None

    def _loop(self):
        while True:
            time.sleep(0.2)
            with self.lock:
                if self.deadline is None or time.monotonic() <= self.deadline:
                    continue
                label, line = self.label, self.line
                print(f"[bench] rank {self.rank}: watchdog: '{label}' neither finished nor threw within its bound; "
                      f"{'printing the line in hand and ' if (line is not None and self.rank == 0) else ''}leaving", file=sys.stderr, flush=True)
                try:                               # where every thread of this rank stands (Python frames), for the record of the run
                    import faulthandler

                    faulthandler.dump_traceback(file=sys.stderr, all_threads=True)
                    sys.stderr.flush()
                except Exception:
                    pass
                code = int(os.environ.get("MI_SPMM_WATCHDOG_EXIT", "0"))
                if not self.measured:
                    code = code or 3               # nothing measured yet: this run has no result
                elif self.printed:
                    code = 0                       # the final line is out already (a hang in tear-down)
                elif self.rank == 0 and line is not None:
                    line["watchdog_fired"] = label        # top level: a reader of the run record sees the stop without looking inside anything
                    line["exchange_watchdog"] = {"fired_in": label, "note": "that section hung (no exception, no completion); this line is the "
                                                 "measurement in hand from before it; the process left with os._exit"}
                    emit_line(line)
                os._exit(code)


def main():
    args = parse()
    rank, world, local = dist_env(args)
    claim_stdout()
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
    wd = Watchdog(rank) if multi else None
    wd_bound = float(os.environ.get("MI_SPMM_WATCHDOG_S", "120"))      # per armed section
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
    from hpc_amd.dist import NativeColumnShardedSpMM, ShardLayout

    t_gen = time.time()
    name, M, n_loc, n_total, ptr, idx, vals, B_loc = build_inputs(args, world, rank)
    nnz = int(idx.size)
    t_gen = time.time() - t_gen
    d_ptr, d_idx, d_val, d_B = (torch.from_numpy(a).to(dev) for a in (ptr, idx, vals, B_loc))
    d_Cfull = None
    if multi:      # peers map this buffer (peer2d / peer_store): allocated at a size HIP IPC can open (C1 on 4 GPUs: 2 GiB -> 4 GiB)
        try:
            from hpc_amd.dist import alloc_c_full
            d_Cfull = alloc_c_full(M, n_total, dev, fill=float("nan"))
        except Exception as e:      # no libmi_spmm_dist.so: the Python schedule below still runs (it maps nothing)
            print(f"[bench] rank {rank}: exportable C_full unavailable ({e!r}); plain allocation", file=sys.stderr, flush=True)
    if d_Cfull is None:
        d_Cfull = torch.full((M, n_total), float("nan"), dtype=torch.float32, device=dev)

    op = SpMMOpt(CSR(M, nnz, d_ptr, d_idx, d_val), n_loc)
    for kv in args.opt:
        k, v = kv.split("=")
        op.set_option(k, int(v))
    t_pre = time.time()
    op.preprocess(d_B, d_Cfull)
    torch.cuda.synchronize()
    t_pre_first = time.time() - t_pre          # includes first-use code-object loading
    t_pre = time.time()
    op.preprocess(d_B, d_Cfull)
    torch.cuda.synchronize()
    t_pre = time.time() - t_pre

    def barrier():
        if multi:
            dist.barrier()

    def trace(what):
        """N > 1: one stderr line per set-up stage and rank (stdout carries the contract line only), so that the record of a run that hangs says how far each rank got."""
        if multi:
            print(f"[bench] rank {rank}: +{time.monotonic() - wd.t0:.1f}s {what}", file=sys.stderr, flush=True)

    def agree(x, red):
        v = torch.tensor([float(x)], dtype=torch.float64, device=dev if not share else "cpu")
        dist.all_reduce(v, op=red)
        return float(v.item())

    # the N > 1 step runs behind the C ABI of include/mi_spmm_dist.h; torch.distributed only bootstraps it
    sharded = None
    step_path = "single GPU"
    py_sharded = None
    d_Cloc = None
    # Exchange schedules: `safe` is the one the contract line is FIRST measured on (warm-up + K timed steps, kept in hand);
    # `candidates` are tried afterwards, each under the watchdog, and the line is re-measured on a candidate only if it is faster.
    #   real ranks over RCCL:  safe = the RCCL all-gather the north star names; candidates = direct, peer2d, peer_store
    #   ranks sharing one GPU (rehearsal): RCCL refuses two ranks on a device -> safe = peer2d over IPC + host barriers, candidate peer_store
    if share:
        safe, candidates = "peer2d", ["peer_store"]
    else:
        safe, candidates = "allgather", ["direct", "peer2d", "peer_store"]
    if args.exchange != "auto":
        if share and args.exchange not in ("peer2d", "peer_store"):
            safe = "peer2d"
        else:
            safe = args.exchange
        candidates = []
    needs_peers = ("peer2d", "peer_store")
    if multi:
        step_path = "libmi_spmm_dist.so (C ABI)"
        native_ok = 1.0
        wd.arm("set-up of the multi-GPU step (communicator, IPC)", wd_bound)
        try:
            if os.environ.get("MI_SPMM_FORCE_PY_DIST") == "1":          # rehearsal of the fallback below (tests)
                raise RuntimeError("MI_SPMM_FORCE_PY_DIST=1")
            sharded = NativeColumnShardedSpMM(op, ShardLayout(M, n_loc, world, rank), n_panels=args.panels,
                                              exchange="allgather", rehearse=args.rehearse_multi)
            trace("step object created (streams, staging)")
            if share:
                sharded.set_option("external_barrier", 1)   # step() below brackets every run with synchronize + dist.barrier
            else:
                sharded.init_comm()    # our own RCCL communicator (unique id broadcast over torch.distributed)
                trace("communicator up")
        except Exception as e:
            print(f"[bench] rank {rank}: native multi-GPU step unavailable: {e!r}", file=sys.stderr, flush=True)
            native_ok = 0.0
        if agree(native_ok, dist.ReduceOp.MIN) < 1.0 and not share:
            # Some rank could not set the C-ABI path up (it has only ever met ONE rank over RCCL on the build boxes): every rank
            # drops to the same schedule stated in Python over torch.distributed's own RCCL group (hpc_amd/dist.py
            # ColumnShardedSpMM: the round-1 driver).  Agreed collectively, reported in the line.
            from hpc_amd.dist import ColumnShardedSpMM
            from hpc_amd.spmm import unpack_gathered
            sharded = None
            step_path = "python schedule over torch.distributed (fallback: the C-ABI step failed to initialise on some rank)"
            d_Cloc = torch.empty((M, n_loc), dtype=torch.float32, device=dev)
            py_sharded = ColumnShardedSpMM(op, ShardLayout(M, n_loc, world, rank), unpack_gathered, n_panels=args.panels,
                                           force_collective=args.rehearse_multi, exchange="allgather")
            safe, candidates = "allgather", []
        if sharded is not None:
            sharded.set_exchange(safe)
            if safe in needs_peers:
                sharded.set_peers(d_Cfull)
                trace("peers' C mapped")
        wd.disarm()
    exchange = safe if multi else None

    def step():
        if not multi:
            op.run(d_B, d_Cfull)
        elif py_sharded is not None:
            py_sharded.run(d_B, d_Cloc, d_Cfull)
        elif share:
            torch.cuda.synchronize(); dist.barrier()
            sharded.run(d_B, d_Cfull)
            torch.cuda.synchronize(); dist.barrier()
        else:
            sharded.run(d_B, d_Cfull)

    if args.sweep and not multi:
        sweep(args, op, step, M, n_loc, nnz)
        return

    def contract_timing():
        """W untimed warm-up steps, then EXACTLY K steps between barrier + synchronize on both sides; max over ranks."""
        for _ in range(args.warmup):
            step()
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
        elapsed = time.perf_counter() - t0
        if multi:
            elapsed = agree(elapsed, dist.ReduceOp.MAX)
        dev_ms = [a.elapsed_time(b) for a, b in ev]
        return {"ms_per_step": elapsed * 1e3 / max(1, args.steps), "dev_ms_mean": float(np.mean(dev_ms)) if dev_ms else float("nan")}

    model = synth.bytes_model(M, M, n_loc, nnz)          # per launch = per GPU
    flops_total = 2.0 * nnz * n_total
    # everything the line is built from; filled in as the run goes, so that a line can be printed at any point after the first timing
    R = {"timing": None, "exchange": exchange, "tuning": None, "ref_protocol_ms": None, "also": None, "check": None, "breakdown": None,
         "strong_ref": None, "cpu": None, "safe_timing": None}

    def build_line():
        ms_per_step, dev_ms_mean = R["timing"]["ms_per_step"], R["timing"]["dev_ms_mean"]
        breakdown, strong_ref = R["breakdown"], R["strong_ref"]
        value = flops_total / (ms_per_step * 1e-3) / 1e9 if args.steps else float("nan")
        achieved = model["bytes_alg"] / (dev_ms_mean * 1e-3) / 1e9 if not multi else None
        roof_ms = dev_ms_mean
        if multi and breakdown and "compute_only_ms" in breakdown:
            # N>1: the dominant kernel is the same per-GPU launch; its duration is the compute-only leg
            # (one launch over all rows, HIP events on the launch stream, max over ranks)
            roof_ms = breakdown["compute_only_ms"]
            achieved = model["bytes_alg"] / (roof_ms * 1e-3) / 1e9
        traffic, traffic_source = None, None
        tp = os.path.join(ROOT, "profiles", "traffic_latest.json")
        if not multi and os.path.exists(tp):
            try:
                tj = json.load(open(tp)).get(name)          # one entry per configuration (C1, C4, ...)
                if tj and tj.get("N") == n_total and M == tj.get("M", 1 << 20):      # the counters of THIS workload, not of a down-sized one
                    traffic = tj.get("hbm_bytes_per_launch")
                    traffic_source = {"file": tj.get("source"), "commit": tj.get("commit"), "note": "measured by rocprofv3 --pmc passes of this "
                                      "command at that commit (scripts/prof.sh), not by this run; FETCH_SIZE x2 + WRITE_SIZE, L2 fabric side",
                                      **traffic_staleness(tj)}
            except Exception:
                traffic = None
        n_long = op.get_option("n_long_rows")
        if op.get_option("split_long_rows") and n_long > 0:
            split_mode = (f"opt-in split: {n_long} rows above {op.get_option('long_row_threshold')} nonzeros summed in pieces of "
                          f"{op.get_option('long_row_chunk')} (bit-exact vs the same piece order; <= 1e-5*sum|a*b| vs the plain chain)")
        else:
            split_mode = ("exact: every row one fma chain in stored order" +
                          (f" ({op.get_option('n_hub_rows')} rows above {op.get_option('long_row_threshold')} nonzeros through the hub kernel)"
                           if op.get_option("n_hub_rows") else ""))
        n_blk = op.get_option("n_block_groups")
        if n_blk * 16 * 2 > M and not multi:
            # block-dense input: the step is the MFMA block kernels (BASELINE configs[4])
            tf = flops_total / (dev_ms_mean * 1e-3) / 1e12
            blk_bytes = 4 * nnz + 4 * nnz // 16 + 4 * n_loc * nnz // 16 + 4 * M * n_loc
            roof = {"bound": "mfma", "kernel": "mi::spmm_block_items (v_mfma_f32_16x16x4_f32)", "achieved": round(tf, 2), "peak": 157.3,
                    "unit": "TFLOP/s", "frac": round(tf / 157.3, 4), "traffic": traffic, "traffic_per": "step (all launches of the block kernels)",
                    "traffic_source": traffic_source, "kernel_ms": round(dev_ms_mean, 4), "launches_per_step": op.get_option("n_launches"),
                    "bytes_min_per_step": model["bytes_min"], "bytes_block_reuse_model_per_step": int(blk_bytes),
                    "bytes_gather_model_per_step": model["bytes_alg"],
                    "GBs_on_bytes_min": round(model["bytes_min"] / (dev_ms_mean * 1e-3) / 1e9, 1),
                    "block_items": {k: op.get_option(k) for k in ("n_block_groups", "n_block_pieces", "n_block_items", "n_block_shared_items",
                                                                  "n_block_passes")}}
        elif achieved is not None and not multi and name != "C1" and 4.0 * M * n_loc <= 256 * 1048576:      # (C1 is the contract line: "hbm"; its B is 512 MiB at the contract's size)
            roof = cache_resident_roofline(model, dev_ms_mean, traffic, traffic_source, M, n_loc, op.get_option("n_launches"))
        elif achieved is not None:
            roof = {"bound": "hbm", "bound_detail": "l2_fabric (gather model): algorithmic bytes over the L2<->fabric path, Infinity-Cache hits "
                    "included; the contract's token stays \"hbm\", the HBM copy ceiling is frac_of_measured_copy_6290's denominator",
                    "kernel": "mi::spmm_rows_v2", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_source,
                    "rate_kind": "algorithmic (gather model) bytes / step time; the L2<->fabric path incl. Infinity-Cache hits carries it, "
                                 "so it may exceed the 6.29 TB/s HBM copy rate",
                    "bytes_alg_per_launch": model["bytes_alg"], "bytes_min_per_launch": model["bytes_min"],
                    "kernel_ms": round(roof_ms, 4), "frac_of_measured_copy_6290": round(achieved / 6290.0, 4),
                    "split_mode": split_mode, "tile_cols_in_force": 4 * op.get_option("lanes_per_row"),
                    "column_locality_pct": op.get_option("column_locality_pct")}
        else:
            roof = None
        exch = R["exchange"]
        line = {
            "metric": "spmm_gflops", "value": round(value, 2), "unit": "GFLOP/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
            "scaling": args.mode, "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {
                "workload": f"{name}: CSR SpMM M=K={M}, nnz={nnz} (deg mean {nnz / max(1, M):.1f}, max {int(np.diff(ptr).max()) if M else 0}), "
                            f"N={n_total} fp32 ({n_loc} columns per GPU), int32 indices",
                "M": M, "K": M, "nnz": nnz, "N": n_total, "cols_per_gpu": n_loc,
                "parallelism": "single GPU" if world == 1 else f"column-sharded x{world}, C blocks exchanged over RCCL/xGMI ({exch} schedule, "
                               f"{step_path}), {args.panels} row panels",
                "options": {k: op.get_option(k) for k in ("kernel", "rows_per_block", "block_threads", "xcd_remap", "nt_store", "nt_stream",
                                                          "medium_row_threshold", "long_row_threshold", "split_long_rows", "n_hub_rows", "hub_overlap", "segment_unroll", "n_long_rows", "n_chunks",
                                                          "lanes_per_row", "vector_width", "n_launches", "tile_cols")},
                "preprocess_ms": round(t_pre * 1e3, 2), "preprocess_first_call_ms": round(t_pre_first * 1e3, 2), "input_gen_s": round(t_gen, 1),
            },
            "device_ms_per_step": round(dev_ms_mean, 4),
            "ms_per_step_ref_protocol": round(R["ref_protocol_ms"], 4) if R["ref_protocol_ms"] is not None else None,
            "roofline": roof,
            "cpu_baseline": R["cpu"],
        }
        if R["also"] is not None:
            line["also"] = R["also"]
        if multi:
            line["exchange"] = exch
            line["exchange_selection"] = {
                "rule": ("safe-first auto: the line is first measured on the safe schedule and kept in hand; every other schedule is then tried under a "
                         "per-rank watchdog (set-up + 2 steps, agreed collectively) and the line is re-measured on one only if it is faster"
                         if args.exchange == "auto" else f"fixed by --exchange {args.exchange}"),
                "safe_schedule": safe, "safe_schedule_ms_per_step": round(R["safe_timing"]["ms_per_step"], 4) if R["safe_timing"] else None,
                "exchange_tuning_ms_per_step": R["tuning"], "watchdog_bound_s": wd_bound}
            line["strong_reference_ms"] = round(strong_ref, 4) if strong_ref else None
            line["speedup_vs_one_gpu"] = round(strong_ref / ms_per_step, 3) if strong_ref else None
            line["strong_reference_note"] = (f"one GPU, all N={n_total} columns, same CSR (rank 0, outside the timed region); "
                                             f"step speed-up vs it = {strong_ref / ms_per_step:.2f}x" if strong_ref else None)
        if share:
            line["rehearsal"] = f"{world} ranks sharing one GPU (MI_SPMM_SHARE_GPU=1): IPC peer copies + gloo barriers; launch-line rehearsal, not a result"
        if R["check"] is not None:
            line["check"] = R["check"]
        if breakdown is not None:
            line["multi_gpu_breakdown"] = breakdown
        return line

    # ---- the contract measurement, on the safe schedule first
    if multi:
        wd.arm(f"warm-up + timed steps on '{exchange}'", wd_bound)
    R["timing"] = contract_timing()
    if multi:
        wd.disarm()
        R["safe_timing"] = dict(R["timing"])
        wd.mark_measured()
        if rank == 0:
            wd.set_line(build_line())          # from here on a hang costs the candidates, not the measurement

    def try_exchange(name_):
        """Collective: set the schedule up and time two steps; (ok on every rank, slowest rank's ms)."""
        ok, ms = 1.0, 0.0
        try:
            if os.environ.get("MI_SPMM_FORCE_HANG_EXCHANGE") == name_:      # tests: a candidate that neither finishes nor throws
                print(f"[bench] rank {rank}: MI_SPMM_FORCE_HANG_EXCHANGE={name_}: hanging on purpose", file=sys.stderr, flush=True)
                while True:
                    time.sleep(3600)
            if os.environ.get("MI_SPMM_FORCE_FAIL_EXCHANGE") == name_:      # tests: a candidate that throws on this rank
                raise RuntimeError(f"MI_SPMM_FORCE_FAIL_EXCHANGE={name_}")
            sharded.set_exchange(name_)
            if name_ in needs_peers:
                sharded.set_peers(d_Cfull)
        except Exception as e:                                    # every rank still takes part in the agreement below
            print(f"[bench] rank {rank}: exchange {name_} unavailable: {e!r}", file=sys.stderr, flush=True)
            ok = 0.0
        # agreed BEFORE the first step: a step holds a cross-rank barrier, which a rank whose set-up failed would never enter
        if agree(ok, dist.ReduceOp.MIN) < 1.0:
            return False, None
        try:
            step()
            torch.cuda.synchronize()
        except Exception as e:
            print(f"[bench] rank {rank}: exchange {name_}: first step failed: {e!r}", file=sys.stderr, flush=True)
            ok = 0.0
        if agree(ok, dist.ReduceOp.MIN) < 1.0:
            return False, None
        dist.barrier()
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(2):
            step()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t) * 1e3 / 2
        return True, agree(ms, dist.ReduceOp.MAX)

    if multi and sharded is not None and candidates:
        tuning = {safe: round(R["safe_timing"]["ms_per_step"], 4)}
        R["tuning"] = tuning
        for cand in candidates:
            wd.arm(f"candidate schedule '{cand}' (set-up + 2 steps)", wd_bound)
            ok, ms = try_exchange(cand)
            wd.disarm()
            tuning[cand] = round(ms, 4) if ok else None
            if rank == 0:
                wd.set_line(build_line())
        best = min((k for k in tuning if tuning[k] is not None), key=lambda k: tuning[k])   # identical on every rank (agreed values)
        if best != safe:
            wd.arm(f"re-measuring on '{best}'", wd_bound)
            sharded.set_exchange(best)
            if best in needs_peers:
                sharded.set_peers(d_Cfull)
            again = contract_timing()
            wd.disarm()
            if again["ms_per_step"] < R["safe_timing"]["ms_per_step"]:      # agreed (max over ranks) on every rank
                R["timing"], R["exchange"] = again, best
            tuning[best + " (contract protocol)"] = round(again["ms_per_step"], 4)
        exchange = R["exchange"]
        wd.arm(f"returning to '{exchange}'", wd_bound)
        sharded.set_exchange(exchange)
        if exchange in needs_peers:
            sharded.set_peers(d_Cfull)
        step()
        torch.cuda.synchronize()
        barrier()
        wd.disarm()
        if rank == 0:
            wd.set_line(build_line())

    # the reference's protocol (util.h:141-151): mean of 20 runs, each bracketed by a device synchronise
    if not multi:
        ts = []
        for _ in range(20):
            torch.cuda.synchronize()
            t = time.perf_counter()
            step()
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t)
        R["ref_protocol_ms"] = float(np.mean(ts)) * 1e3

    if not multi and rank == 0 and name == "C1" and not args.no_also and not args.opt:
        R["also"] = also_configs(args, dev, (d_ptr, d_idx, nnz), M)

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
        R["check"] = {"rows": int(rows.size), "bitwise_equal_rows": int((got.view(np.uint32) == exp.view(np.uint32)).all(axis=1).sum()),
                      "short_rows_all_equal": bool((got.view(np.uint32)[short] == exp.view(np.uint32)[short]).all())}

    # N > 1: compute-only and exchange-only legs, outside the timed region (report
    # compute scaling and end-to-end scaling separately; the step is bound by the exchange)
    if multi:
      wd.arm("breakdown legs (compute only, exchange only)", 2 * wd_bound)
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
            return agree(a.elapsed_time(b) / reps, dist.ReduceOp.MAX)

        d_scratch = torch.empty(M, n_loc, dtype=torch.float32, device=dev)
        compute_ms = timed_ms(lambda: op.run_rows(d_B, n_loc, d_scratch, n_loc, 0, M))     # one launch set over all rows
        del d_scratch
        if py_sharded is not None:
            from hpc_amd.spmm import unpack_gathered
            stage = torch.empty(world * M * n_loc, dtype=torch.float32, device=dev)

            def exchange_leg():
                py_sharded._exchange(stage, d_Cloc.view(-1), M)
                unpack_gathered(stage, d_Cfull, M, world, n_loc, n_total)
        elif share:
            def exchange_leg():
                torch.cuda.synchronize(); dist.barrier()
                sharded.run_exchange_only(d_Cfull)
                torch.cuda.synchronize(); dist.barrier()
        else:
            def exchange_leg():
                sharded.run_exchange_only(d_Cfull)
        exchange_ms = timed_ms(exchange_leg, reps=3)
        moved = (world - 1) * M * n_loc * 4
        R["breakdown"] = {"compute_only_ms": round(compute_ms, 4), "exchange_only_ms": round(exchange_ms, 4),
                          "bytes_received_per_gpu": int(moved),
                          "exchange_GBs_in_per_gpu": round(moved / (exchange_ms * 1e-3) / 1e9, 1) if exchange_ms > 0 else None,
                          "compute_only_gflops_total": round(flops_total / (compute_ms * 1e-3) / 1e9, 1),
                          "exchange": R["exchange"], "exchange_tuning_ms_per_step": R["tuning"], "step_path": step_path,
                          "staging_bytes": sharded.get_option("staging_bytes") if sharded is not None else int(2 * world * M * n_loc * 4 / max(1, args.panels))}
        step()                       # leave a complete C behind (the legs are timing legs)
        torch.cuda.synchronize()
        barrier()
      except Exception as e:   # the breakdown is a courtesy: it must never cost the contract line
        R["breakdown"] = {"error": repr(e)[:200]}
      wd.disarm()
      if rank == 0:
          wd.set_line(build_line())

    # N > 1: what the links deliver (SURVEY.md H3 "measure link bandwidth first") -- outside the timed region, under the watchdog: a probe
    # that hangs costs the probe, not the line.  Plain device-to-device copies into the peers' C (mapped through HIP IPC), one peer at a time, then all at once.
    if multi and sharded is not None and not args.no_link_probe:
        wd.arm("link probe (peer copies)", wd_bound)
        links = None
        ok = 1.0
        try:
            if os.environ.get("MI_SPMM_FORCE_HANG_EXCHANGE") == "link_probe":      # tests: a probe that neither finishes nor throws
                while True:
                    time.sleep(3600)
            if share:       # no communicator between ranks that share a GPU: the probe's barriers go through the host
                sharded.set_host_barrier(lambda: (torch.cuda.synchronize(), dist.barrier()))
            if R["exchange"] not in needs_peers:
                sharded.set_peers(d_Cfull)
        except Exception as e:
            print(f"[bench] rank {rank}: link probe unavailable: {e!r}", file=sys.stderr, flush=True)
            ok = 0.0
        if agree(ok, dist.ReduceOp.MIN) >= 1.0:          # the probe is collective: either every rank enters it or none does
            try:
                # one process per GPU on one node, every device visible: rank q runs on device q (LOCAL_RANK); ranks that share a GPU have no link between them
                pdev = [-1] * world if share or torch.cuda.device_count() < world else list(range(world))
                links = sharded.link_probe(d_Cfull, 256 << 20, pdev)
            except Exception as e:
                links = {"error": repr(e)[:200]}
            if share:
                sharded.set_host_barrier(None)
            step()                   # the probe wrote junk into every C: leave a complete one behind
            torch.cuda.synchronize()
            barrier()
        if links is not None and "error" not in links:
            # the slowest rank's view of every number (a step ends when the slowest rank's exchange does)
            links["all_peers_GBs_min_over_ranks"] = round(agree(links["all_peers_GBs"], dist.ReduceOp.MIN), 1)
            slowest = min((x for q, x in enumerate(links["per_peer_GBs"]) if q != rank), default=0.0)
            links["per_peer_GBs_min_over_ranks"] = round(agree(slowest, dist.ReduceOp.MIN), 1)
            moved = (world - 1) * M * n_loc * 4
            rate = links["all_peers_GBs_min_over_ranks"]
            links["link_bound_ms"] = round(moved / (rate * 1e9) * 1e3, 4) if rate > 0 else None
            links["link_bound_note"] = ("bytes a GPU sends per step (world - 1 column blocks) / the rate its links delivered together in the probe (slowest rank): "
                                        "the exchange cannot finish sooner on these links; compare with ms_per_step and exchange_only_ms")
            if share:
                links["note"] = "ranks share one GPU: the copies never left the device; the numbers exercise the code path, they are not link rates"
        if R["breakdown"] is None:
            R["breakdown"] = {}
        R["breakdown"]["links"] = links
        wd.disarm()
        if rank == 0:
            wd.set_line(build_line())

    # N > 1: what ONE GPU needs for all N_total columns (the north star's ">= 6x at 8 GPUs on N = 1024" is against this)
    if multi and rank == 0 and not args.no_strong_reference:
        wd.arm("one-GPU strong reference", 2 * wd_bound)
        try:
            d_Ball = torch.empty(M * n_total, dtype=torch.float32, device=dev)
            from hpc_amd.spmm import fill_normal
            fill_normal(d_Ball, seed=synth.SEED_B)                     # timing only: any N(0, 0.1) B of that shape
            one = SpMMOpt(CSR(M, nnz, d_ptr, d_idx, d_val), n_total)
            one.preprocess(d_Ball, d_Cfull)
            for _ in range(2):
                one.run(d_Ball, d_Cfull)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(5):
                one.run(d_Ball, d_Cfull)
            b.record()
            torch.cuda.synchronize()
            R["strong_ref"] = a.elapsed_time(b) / 5
            del one, d_Ball
        except Exception as e:
            R["strong_ref"] = None
            print(f"[bench] strong reference skipped: {e!r}", file=sys.stderr, flush=True)
        wd.disarm()
        wd.set_line(build_line())
    if multi:
        wd.arm("final barrier", wd_bound)
        barrier()
        wd.disarm()

    if rank == 0 and not multi and not args.no_cpu_baseline:
        R["cpu"] = cpu_baseline(args, ptr, idx, vals, B_loc, M, n_loc)

    if rank == 0:
        if wd is not None:
            wd.set_line(None)          # the final line is printed here, once
        emit_line(build_line())
    if multi:
        wd.arm("tear-down", wd_bound)   # nothing in hand any more: a hang here leaves with the watchdog's code, the line is out
        del sharded
        dist.destroy_process_group()
        wd.disarm()


def cache_resident_roofline(model, ms, traffic, src, M, n, launches):
    """B fits the Infinity Cache (4 K N <= 256 MiB) -- and, strip by strip, an XCD's L2: the gather model's bytes are NOT what the memory system moves, so a
    fraction of the HBM peak on them can exceed 1 (round 4 printed 2.25 under "hbm").  The honest bound: what crossed the L2 <-> fabric side (PMC) over the
    step, against the fabric's 8 TB/s; without counters, the gather model against the L2-resident gather rate of the guide (18.8 TB/s) -- and no fraction at
    all when even that is exceeded (a B of a few MiB with ascending columns is served in front of the L2)."""
    gbs = model["bytes_alg"] / (ms * 1e-3) / 1e9
    if traffic is not None:
        ach = traffic / (ms * 1e-3) / 1e9
        roof = {"bound": "fabric", "bound_detail": "B is cache-resident (4 K N <= 256 MiB): measured L2<->fabric bytes (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE) "
                "per step over this run's step time, against the fabric's 8 TB/s", "peak": HBM_PEAK_GBS, "achieved": round(ach, 1), "frac": round(ach / HBM_PEAK_GBS, 4)}
    else:
        roof = {"bound": "l2_gather", "bound_detail": "B is cache-resident (4 K N <= 256 MiB) and no counters are at hand: gather-model bytes against the "
                "L2-resident gather rate (MI355X_MICROARCH.md \"Indexed rows\": 16.8-18.8 TB/s chip-wide)", "peak": 18800.0, "achieved": round(gbs, 1),
                "frac": round(gbs / 18800.0, 4)}
        if gbs > 18800.0:
            roof["frac"] = None
            roof["frac_note"] = "achieved exceeds the cited L2 gather rate: the gathers are served in front of the L2 (vector L1); no fraction is claimed"
    roof.update({"kernel": "mi::spmm_chunks (column strips) + mi::spmm_rows_v2", "unit": "GB/s", "achieved_alg": round(gbs, 1),
                 "achieved_alg_note": "gather-model bytes / step time: served mostly by L2 and the Infinity Cache, not a fraction of any memory peak",
                 "b_bytes": int(4 * M * n), "bytes_alg": model["bytes_alg"], "bytes_min": model["bytes_min"], "traffic": traffic, "traffic_source": src,
                 "launches_per_step": launches})
    return roof


def traffic_staleness(entry):
    """Is a profiles/traffic_latest.json entry about the kernels this run executes?  The entry carries the sha256 of the
    kernel sources it was measured on (hpc_amd/_lib.py KERNEL_SOURCES); `traffic_stale` = that hash is absent or differs
    from the tree bench.py runs from -- the counters then describe an earlier kernel and say so in the line."""
    from hpc_amd._lib import kernel_sources_sha256

    here, there = kernel_sources_sha256(), entry.get("kernel_sources_sha256")
    return {"traffic_stale": bool(here is None or there is None or here != there),
            "kernel_sources_sha256": {"measured": there, "this_run": here}}


def also_configs(args, dev, c1_tensors, M):
    """C2, C4 and C1-at-N=1024 on this GPU, one after the other, each: preprocess (untimed), 3 warm-up runs, args.steps
    timed runs back to back between one HIP event pair on the launch stream (the headline's protocol).  Values and B are N(0, 0.1) filled on the device
    (mi_spmm_fill_normal); structures come from the same seeded generators as the tests' configurations."""
    import torch
    from hpc_amd import CSR, SpMMOpt, synth
    from hpc_amd.spmm import fill_normal

    traffic_file = {}
    tp = os.path.join(ROOT, "profiles", "traffic_latest.json")
    if os.path.exists(tp):
        try:
            traffic_file = json.load(open(tp))
        except Exception:
            traffic_file = {}

    M_c1 = M

    def run_one(tag, d_ptr, d_idx, n, nnz, extra, M=None, chain=False):
        M = M_c1 if M is None else M
        d_val = torch.empty(nnz, dtype=torch.float32, device=dev)
        fill_normal(d_val, synth.SEED_VALS)
        d_B = torch.empty(M * n, dtype=torch.float32, device=dev)
        fill_normal(d_B, synth.SEED_B)
        d_B = d_B.view(M, n)
        d_C = torch.empty((M, n), dtype=torch.float32, device=dev)
        op = SpMMOpt(CSR(M, nnz, d_ptr, d_idx, d_val), n)
        op.preprocess(d_B, d_C)
        torch.cuda.synchronize()
        t = time.time()
        op.preprocess(d_B, d_C)
        torch.cuda.synchronize()
        pre_ms = (time.time() - t) * 1e3
        for _ in range(3):
            op.run(d_B, d_C)
        # the headline's protocol: the steps back to back inside ONE bracket (an event pair per run would time every step from an
        # idle GPU: C4's two launches and their side-stream fork then read 5 % longer than in `bench.py --config C4`)
        n_steps = max(1, args.steps)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(n_steps):
            op.run(d_B, d_C)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / n_steps
        model = synth.bytes_model(M, M, n, nnz)
        flops = 2.0 * nnz * n
        tj = traffic_file.get(tag)
        traffic = tj.get("hbm_bytes_per_launch") if tj and tj.get("N") == n and M == tj.get("M", 1 << 20) else None
        src = ({"file": tj.get("source"), "commit": tj.get("commit"), "note": "rocprofv3 --pmc passes at that commit, not this run",
                **traffic_staleness(tj)} if traffic is not None else None)
        n_blk = op.get_option("n_block_groups")
        if n_blk * 16 * 2 > M:
            tf = flops / (ms * 1e-3) / 1e12
            roof = {"bound": "mfma", "bound_detail": "fp32 MFMA (v_mfma_f32_16x16x4_f32) peak 157.3 TFLOP/s; the step is also reported on its bytes",
                    "kernel": "mi::spmm_block_items", "achieved": round(tf, 2), "peak": 157.3, "unit": "TFLOP/s", "frac": round(tf / 157.3, 4),
                    "bytes_min": model["bytes_min"], "GBs_on_bytes_min": round(model["bytes_min"] / (ms * 1e-3) / 1e9, 1),
                    "traffic": traffic, "traffic_source": src, "launches_per_step": op.get_option("n_launches"),
                    "block_items": {k: op.get_option(k) for k in ("n_block_groups", "n_block_pieces", "n_block_items",
                                                                  "n_block_shared_items", "n_block_passes")}}
        elif chain:
            # one row's dependent fma chain is the step (am-shaped kLen 32: a 142 153-nonzero row in a 5.8 M-nonzero matrix): the bound is the chain, not bytes.
            # floor: ~5 cycles per dependent v_fmac_f32 (scripts/experiments/fma_chain_micro.hip); achieved: the WHOLE step's time over the
            # longest row (the hub kernel runs beside the rows kernel on its side stream; launch and fork / join included) -- an upper bound on the chain's cost
            longest = op.get_option("max_row_nnz")
            cyc = ms * 1e-3 * HUB_CLOCK_HZ / max(1, longest)
            gbs = model["bytes_alg"] / (ms * 1e-3) / 1e9
            roof = {"bound": "chain", "bound_detail": "the longest row's stored-order fma chain (spmm_ref.cu:10-14 allows no other order): floor ~5 cycles per nonzero "
                    "of that row at 2.4 GHz; achieved = step time x clock / longest row",
                    "kernel": "mi::spmm_hub" + (" (as the hub role of mi::spmm_small_step: the step is one launch)" if op.get_option("fused_step_in_force") else ""),
                    "achieved": round(cyc, 2), "peak": 5.0,
                    "unit": "cycles per nonzero of the longest row (lower is better; frac = floor / achieved)", "frac": round(5.0 / cyc, 4),
                    "cycles_per_nonzero": round(cyc, 2), "floor_cycles": 5.0, "longest_row": int(longest), "ns_per_nonzero": round(ms * 1e6 / max(1, longest), 3),
                    "chain_floor_ms": round(5.0 * longest / HUB_CLOCK_HZ * 1e3, 4), "bytes_alg": model["bytes_alg"], "bytes_min": model["bytes_min"],
                    "GBs_on_bytes_alg": round(gbs, 1), "bytes_floor_ms": round(model["bytes_alg"] / 8e12 * 1e3, 4),
                    "traffic": traffic, "traffic_source": src, "launches_per_step": op.get_option("n_launches"), "n_hub_rows": op.get_option("n_hub_rows")}
        elif 4.0 * M * n <= 256 * 1048576:
            roof = cache_resident_roofline(model, ms, traffic, src, M, n, op.get_option("n_launches"))
        else:
            gbs = model["bytes_alg"] / (ms * 1e-3) / 1e9
            roof = {"bound": "hbm", "bound_detail": "l2_fabric (gather model): algorithmic bytes over the L2<->fabric path, Infinity-Cache hits included",
                    "kernel": "mi::spmm_rows_v2 (+ segment / hub kernels on their side streams)", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4), "frac_of_measured_copy_6290": round(gbs / 6290.0, 4),
                    "bytes_alg": model["bytes_alg"], "bytes_min": model["bytes_min"], "traffic": traffic, "traffic_source": src,
                    "launches_per_step": op.get_option("n_launches")}
        out = {"config": f"{tag}: {extra}, M=K={M}, nnz={nnz}, N={n} fp32", "ms_per_step": round(ms, 4), "steps": n_steps,
               "value": round(flops / (ms * 1e-3) / 1e9, 2), "unit": "GFLOP/s", "preprocess_ms": round(pre_ms, 2), "roofline": roof,
               "summation_order": ("exact (stored order on every row)" if not op.get_option("split_long_rows") else "split"),
               "options": {k: op.get_option(k) for k in ("long_row_threshold", "n_hub_rows", "n_medium_rows", "lanes_per_row", "tile_cols", "n_col_strips",
                                                          "hub_slice", "max_row_nnz")}}
        del op, d_val, d_B, d_C
        torch.cuda.empty_cache()
        return out

    res = {}
    try:
        ptr, idx = synth.csr_powerlaw(M, 32.0, 4096)
        d_ptr, d_idx = torch.from_numpy(ptr).to(dev), torch.from_numpy(idx).to(dev)
        res["C2"] = run_one("C2", d_ptr, d_idx, 128, int(idx.size), f"power-law rows (max {int(np.diff(ptr).max())})")
        del d_ptr, d_idx, ptr, idx
    except Exception as e:
        res["C2"] = {"error": repr(e)[:200]}
    try:
        d_ptr, d_idx = synth.csr_block_dense_fast_device(M, dev)
        res["C4"] = run_one("C4", d_ptr, d_idx, 256, int(d_idx.numel()), "block-dense rows (16-row groups sharing 1-2 runs of 64-128 columns)")
        del d_ptr, d_idx
    except Exception as e:
        res["C4"] = {"error": repr(e)[:200]}
    try:
        # long rows over few columns (protein- / reddit-like): the segments run strip by strip out of an L2-sized piece of B
        Md = max(4096, M // 8)
        d_ptr, d_idx = synth.csr_long_rows_device(Md, dev)
        res["LONG_ROWS"] = run_one("LONG_ROWS", d_ptr, d_idx, 128, int(d_idx.numel()),
                                   "300-700 nonzeros in every row, columns ascending over all K (generated on the device)", M=Md)
        del d_ptr, d_idx
    except Exception as e:
        res["LONG_ROWS"] = {"error": repr(e)[:200]}
    try:
        # the hub kernel in the driver's line (VERDICT r4 #3a): am-shaped kLen 32 -- one row of 142 153 nonzeros in a 5.8 M-nonzero matrix, chain-bound
        if M == (1 << 20):
            ptr, idx = synth.csr_dataset_shaped("am")
        else:       # down-sized runs (tests): the same shape in proportion
            Ma = max(4096, M * 881_680 // (1 << 20))
            ptr, idx = synth.csr_powerlaw(Ma, 6.43, max(512, min(Ma, Ma * 154_828 // 881_680)), seed=sum(map(ord, "am")) % 1000 + 1, force_max=True)
        d_ptr, d_idx = torch.from_numpy(ptr).to(dev), torch.from_numpy(idx).to(dev)
        res["AM32"] = run_one("AM32", d_ptr, d_idx, 32, int(idx.size), f"am-shaped (synth.DATASET_SHAPES: longest row {int(np.diff(ptr).max())}, the hub kernel's case)",
                              M=ptr.size - 1, chain=True)
        del d_ptr, d_idx, ptr, idx
    except Exception as e:
        res["AM32"] = {"error": repr(e)[:200]}
    try:
        d_ptr, d_idx, nnz = c1_tensors
        res["C1_N1024"] = run_one("C1_N1024", d_ptr, d_idx, 1024, nnz, "C1's CSR, the one-GPU leg of configs[3]")
    except Exception as e:
        res["C1_N1024"] = {"error": repr(e)[:200]}
    return res


def cpu_baseline(args, ptr, idx, vals, B, M, N):
    """Oracle OpenMP restatement on the host cores: 1 warm-up + timed runs, on a bounded row
    prefix of the same workload sized for ~10-30 s of CPU work; all available cores, then ONE thread on a smaller
    prefix.  Rebuilt -march=native on this host when gcc is here (the shipped library is x86-64-v3)."""
    from oracle import oracle

    flags = oracle.try_native()
    rows = args.cpu_rows if args.cpu_rows else M
    rows = min(rows, M)

    def timed(nrows, budget_s):
        out = np.empty((nrows, N), dtype=np.float32)
        sub_ptr = ptr[: nrows + 1]
        nnz = int(sub_ptr[-1])
        t = time.perf_counter()
        oracle.spmm_omp(sub_ptr, idx, vals, B, out=out)          # warm-up (and first-touch of out)
        first = time.perf_counter() - t
        reps = max(1, min(5, int(budget_s / max(first, 1e-3))))
        times = []
        for _ in range(reps):
            t = time.perf_counter()
            oracle.spmm_omp(sub_ptr, idx, vals, B, out=out)
            times.append(time.perf_counter() - t)
        return float(np.mean(times)), reps, nnz

    threads = oracle.num_threads()
    best, reps, nnz = timed(rows, 12.0)
    # one thread: a prefix 1/threads as long, so that it costs about as much wall time as the all-cores point
    rows1 = max(1, min(rows, rows // max(1, threads)))
    oracle.set_threads(1)
    try:
        best1, reps1, nnz1 = timed(rows1, 8.0)
    finally:
        oracle.set_threads(threads)
    return {
        "value": round(2.0 * nnz * N / best / 1e9, 3), "unit": "GFLOP/s", "cores": threads, "kind": "port",
        "sample": f"rows [0,{rows}) of the same workload ({nnz} nnz, N={N}), mean of {reps} runs after 1 warm-up, "
                  f"{best * 1e3:.1f} ms each; oracle/spmm_oracle.c oracle_spmm_omp, {flags}",
        "seconds": round(best, 4), "host_cpus": os.cpu_count(),
        "cores_note": f"{threads} OpenMP threads = the CPUs this job may run on (its cgroup / affinity share of the host's "
                      f"{os.cpu_count()}): BASELINE.md's 'all physical cores' means all cores the job is given",
        "single_thread": {"value": round(2.0 * nnz1 * N / best1 / 1e9, 3), "unit": "GFLOP/s", "cores": 1,
                          "sample": f"rows [0,{rows1}) ({nnz1} nnz), mean of {reps1} runs after 1 warm-up, {best1 * 1e3:.1f} ms each"},
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
        emit_line({"sweep": dict(zip(keys, c)), "ms_median": round(ms, 4), "ms_min": round(min(results[c]), 4),
                   "GBs_alg": round(model["bytes_alg"] / ms / 1e6, 1), "GFLOPs": round(2.0 * nnz * N / ms / 1e6, 1)})


if __name__ == "__main__":
    main()
