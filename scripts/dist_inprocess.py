#!/usr/bin/env python3
"""G ranks of the column-sharded step driven from ONE process on one GPU (mi_spmm_dist_set_peer_pointers): every rank object has
its own operator, B slice and C_full; after a step every C_full must equal the single-operator C bit for bit.  Because it is one
process, rocprofv3's counters see the whole step: scripts/prof.sh r03_dist_<exchange> python3 scripts/dist_inprocess.py --exchange X
gives the bytes a step moves on a shared GPU (profiles/r03_dist_bytes.md).

    python scripts/dist_inprocess.py --world 4 --exchange peer_store [--M 262144 --n-loc 128 --steps 10]
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def build(world, M, n_loc, exchange, dev, kind="uniform"):
    import torch
    from hpc_amd import CSR, SpMMOpt, synth
    from hpc_amd.dist import NativeColumnShardedSpMM, ShardLayout

    ptr, idx = (synth.csr_uniform(M, 16, 48) if kind == "uniform" else synth.csr_powerlaw(M, 24.0, 3000, seed=21, force_max=True))
    vals = synth.make_values(idx.size)
    d_ptr, d_idx, d_val = (torch.from_numpy(a).to(dev) for a in (ptr, idx, vals))
    B = [torch.from_numpy(synth.normal_f32(M * n_loc, synth.SEED_B, stream=r).reshape(M, n_loc)).to(dev) for r in range(world)]
    C = [torch.full((M, n_loc * world), float("nan"), device=dev) for _ in range(world)]
    ops, shs = [], []
    for r in range(world):
        op = SpMMOpt(CSR(M, idx.size, d_ptr, d_idx, d_val), n_loc)
        op.preprocess(B[r], C[r])
        ops.append(op)
        sh = NativeColumnShardedSpMM(op, ShardLayout(M, n_loc, world, r), n_panels=4, exchange=exchange)
        sh.set_option("external_barrier", 1)          # one process, one stream: the step order below IS the barrier
        shs.append(sh)
    for r in range(world):
        shs[r].set_peer_tensors(C[r], C)
    B_all = torch.cat(B, dim=1).contiguous()
    C_one = torch.empty(M, n_loc * world, device=dev)
    one = SpMMOpt(CSR(M, idx.size, d_ptr, d_idx, d_val), n_loc * world)
    one.set_option("long_row_threshold", ops[0].get_option("long_row_threshold"))
    one.preprocess(B_all, C_one)
    one.run(B_all, C_one)
    torch.cuda.synchronize()
    return shs, B, C, C_one, (d_ptr, d_idx, d_val, ops, one, B_all), idx.size


def step(shs, B, C):
    for r, sh in enumerate(shs):
        sh.run(B[r], C[r])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--world", type=int, default=4)
    ap.add_argument("--M", type=int, default=1 << 18)
    ap.add_argument("--n-loc", type=int, default=128)
    ap.add_argument("--exchange", default="peer_store", choices=["peer2d", "peer_store", "none"])
    ap.add_argument("--steps", type=int, default=10)
    args = ap.parse_args()
    import torch

    dev = torch.device("cuda:0")
    ex = "peer2d" if args.exchange == "none" else args.exchange
    shs, B, C, C_one, keep, nnz = build(args.world, args.M, args.n_loc, ex, dev)
    if args.exchange == "none":                     # the compute legs alone: every rank's block into its own C_full only
        def run():
            for r, sh in enumerate(shs):
                sh.run_compute_only(B[r], C[r])
    else:
        def run():
            step(shs, B, C)
    for _ in range(2):
        run()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(args.steps):
        run()
    b.record()
    torch.cuda.synchronize()
    ok = None
    if args.exchange != "none":
        ok = all(bool(torch.equal(c.view(torch.int32), C_one.view(torch.int32))) for c in C)
    print(f"world {args.world} ranks in one process, exchange {args.exchange}: M {args.M}, n_loc {args.n_loc}, nnz {nnz}: "
          f"{a.elapsed_time(b) / args.steps:.3f} ms per step of all ranks; every C_full == single-operator C: {ok}", flush=True)
    if ok is False:
        sys.exit(2)


if __name__ == "__main__":
    main()
