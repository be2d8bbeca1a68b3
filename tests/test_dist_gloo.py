"""The N>1 path on CPU: world_size-2 (3, 4, 8) gloo runs of the column-sharded driver, both exchange schedules.

The host logic under test is hpc_amd/dist.py (column blocks, row panels, all-gather, unpack).
The device pieces are replaced by test doubles: the local operator is the oracle restricted to
a row range, the unpack step is a torch permute -- so what is checked is the schedule and the
layout: every rank must end with the row-major C of the 1-process computation, bit for bit."""
import os
import socket

import numpy as np
import pytest

from hpc_amd import synth
from hpc_amd.dist import ColumnShardedSpMM, ShardLayout, column_block, row_panels


def test_column_blocks_and_panels():
    assert column_block(1024, 8, 3) == (384, 512)
    with pytest.raises(ValueError):
        column_block(100, 8, 0)
    p = row_panels(1 << 20, 8)
    assert p[0][0] == 0 and p[-1][1] == 1 << 20 and len(p) == 8
    assert all(a[1] == b[0] for a, b in zip(p, p[1:]))
    assert row_panels(1000, 8, align=256) == [(0, 256), (256, 512), (512, 768), (768, 1000)]
    assert row_panels(0, 4) == []
    assert row_panels(5, 4, align=256) == [(0, 5)]


class OracleRowsOp:
    """Test double with SpMMOpt's run_rows signature, computing with the oracle on CPU tensors."""

    def __init__(self, ptr, idx, vals):
        self.ptr, self.idx, self.vals = ptr, idx, vals
        self.calls = []

    def run_rows(self, B_loc, ldb, C, ldc, r0, r1):
        from oracle import oracle

        self.calls.append((r0, r1))
        Bn = B_loc.numpy()
        n_loc = Bn.shape[1]
        out = C.numpy().reshape(-1)[: (self.ptr.size - 1) * ldc].reshape(self.ptr.size - 1, ldc)
        tmp = np.empty((self.ptr.size - 1, n_loc), np.float32)
        oracle.spmm_omp(self.ptr, self.idx, self.vals, Bn, out=tmp, row_begin=r0, row_end=r1)
        out[r0:r1, :n_loc] = tmp[r0:r1]


def torch_unpack(staging, C_flat, rows, G, n_loc, ldc):
    st = staging[: G * rows * n_loc].view(G, rows, n_loc)
    C_flat[: rows * ldc].view(rows, ldc)[:, : G * n_loc] = st.permute(1, 0, 2).reshape(rows, G * n_loc)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, M, n_loc, n_panels, q, exchange="allgather"):
    import torch
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["ORACLE_THREADS"] = "2"
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ptr, idx = synth.csr_uniform(M, 0, 24, seed=5)
        vals = synth.normal_f32(idx.size, 6)
        B_loc = torch.from_numpy(synth.normal_f32(M * n_loc, synth.SEED_B, stream=rank).reshape(M, n_loc))
        op = OracleRowsOp(ptr, idx, vals)
        sh = ColumnShardedSpMM(op, ShardLayout(M, n_loc, world, rank), torch_unpack, n_panels=n_panels, use_streams=False,
                               exchange="allgather" if exchange == "tune" else exchange)
        C_loc = torch.zeros(M, n_loc)
        C_full = torch.full((M, n_loc * world), float("nan"))
        if exchange == "tune":
            chosen = sh.tune(C_loc, reps=2)
            assert chosen in sh.EXCHANGES and sh.tuning["allgather"] > 0 and sh.tuning["direct"] > 0
            picks = [None] * world
            dist.all_gather_object(picks, chosen)
            assert len(set(picks)) == 1, picks          # every rank kept the same schedule
        sh.run(B_loc, C_loc, C_full)
        sh.run(B_loc, C_loc, C_full)      # idempotent, staging reuse
        q.put((rank, C_full.numpy().copy(), list(op.calls)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,M,n_loc,n_panels,exchange", [(2, 1000, 16, 3, "allgather"), (2, 777, 5, 1, "allgather"), (3, 600, 8, 4, "allgather"),
                                                              (2, 1000, 16, 3, "direct"), (3, 600, 8, 4, "direct"), (4, 515, 4, 2, "direct"),
                                                              (3, 600, 8, 4, "tune"), (8, 530, 4, 3, "allgather"), (8, 530, 4, 3, "direct")])
def test_gloo_column_sharded_matches_single_process(world, M, n_loc, n_panels, exchange):
    import torch.multiprocessing as mp
    from oracle import oracle

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, M, n_loc, n_panels, q, exchange)) for r in range(world)]
    for p in procs:
        p.start()
    results = {}
    for _ in range(world):
        r, C, calls = q.get(timeout=120)
        results[r] = (C, calls)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # single-process truth: the full B is the concatenation of the per-rank column blocks
    ptr, idx = synth.csr_uniform(M, 0, 24, seed=5)
    vals = synth.normal_f32(idx.size, 6)
    B = np.concatenate([synth.normal_f32(M * n_loc, synth.SEED_B, stream=r).reshape(M, n_loc) for r in range(world)], axis=1)
    exp = oracle.spmm_omp(ptr, idx, vals, np.ascontiguousarray(B))
    for r in range(world):
        C, calls = results[r]
        assert np.array_equal(C.view(np.uint32), exp.view(np.uint32)), f"rank {r}"
        assert calls[: len(calls) // 2] == row_panels(M, n_panels)


def test_world_size_one_writes_c_directly():
    import torch

    M, n_loc = 300, 8
    ptr, idx = synth.csr_uniform(M, 0, 10, seed=9)
    vals = synth.normal_f32(idx.size, 10)
    B = torch.from_numpy(synth.normal_f32(M * n_loc, 11).reshape(M, n_loc))
    op = OracleRowsOp(ptr, idx, vals)
    sh = ColumnShardedSpMM(op, ShardLayout(M, n_loc, 1, 0), torch_unpack, use_streams=False)
    C = torch.full((M, n_loc), float("nan"))
    sh.run(B, C, C)
    from oracle import oracle

    assert np.array_equal(C.numpy().view(np.uint32), oracle.spmm_omp(ptr, idx, vals, B.numpy()).view(np.uint32))
    assert op.calls == [(0, M)]
