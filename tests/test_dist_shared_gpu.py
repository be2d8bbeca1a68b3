"""The N>1 product path on ONE MI355X: two (and three) ranks share cuda:0.

RCCL refuses two ranks on one device, so the process group is gloo carrying device tensors;
everything else is what bench.py --gpus N runs: hpc_amd.SpMMOpt.run_rows for the row panels,
the compute / exchange / unpack streams of hpc_amd/dist.py, the HIP unpack kernel.  Checked:
every rank ends with the row-major C a single operator with N_total columns produces, bit for
bit, on repeated steps (staging reuse across steps and panels)."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, M, n_loc, n_panels, kind, exchange, q):
    import torch
    import torch.distributed as dist

    from hpc_amd import CSR, SpMMOpt, synth
    from hpc_amd.dist import ColumnShardedSpMM, ShardLayout
    from hpc_amd.spmm import unpack_gathered

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        if kind == "powerlaw":
            ptr, idx = synth.csr_powerlaw(M, 12.0, 3000, seed=21, force_max=True)
        else:
            ptr, idx = synth.csr_uniform(M, 0, 40, seed=5)
        vals = synth.normal_f32(idx.size, 6)
        blocks = [synth.normal_f32(M * n_loc, synth.SEED_B, stream=r).reshape(M, n_loc) for r in range(world)]
        d_ptr, d_idx, d_val = (torch.from_numpy(a).to(dev) for a in (ptr, idx, vals))
        B_loc = torch.from_numpy(blocks[rank]).to(dev)
        C_loc = torch.empty(M, n_loc, device=dev)
        C_full = torch.full((M, n_loc * world), float("nan"), device=dev)
        op = SpMMOpt(CSR(M, idx.size, d_ptr, d_idx, d_val), n_loc)
        op.preprocess(B_loc, C_loc)
        sh = ColumnShardedSpMM(op, ShardLayout(M, n_loc, world, rank), unpack_gathered, n_panels=n_panels, exchange=exchange)
        for _ in range(3):
            sh.run(B_loc, C_loc, C_full)
        torch.cuda.synchronize()
        # the 1-GPU answer: one operator over all N_total columns of the concatenated B
        B_all = torch.from_numpy(np.ascontiguousarray(np.concatenate(blocks, axis=1))).to(dev)
        C_one = torch.empty(M, n_loc * world, device=dev)
        one = SpMMOpt(CSR(M, idx.size, d_ptr, d_idx, d_val), n_loc * world)
        one.set_option("long_row_threshold", op.get_option("long_row_threshold"))
        one.preprocess(B_all, C_one)
        one.run(B_all, C_one)
        torch.cuda.synchronize()
        same = bool(torch.equal(C_full.view(torch.int32), C_one.view(torch.int32)))
        q.put((rank, same, bool(sh.use_streams), int(torch.isnan(C_full).sum().item())))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,M,n_loc,n_panels,kind,exchange", [(2, 40000, 128, 8, "uniform", "allgather"), (2, 30011, 32, 5, "powerlaw", "allgather"),
                                                                  (3, 20000, 64, 4, "uniform", "allgather"), (3, 20000, 64, 4, "uniform", "direct")])
def test_ranks_sharing_one_gpu_reproduce_the_single_gpu_result(world, M, n_loc, n_panels, kind, exchange):
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, M, n_loc, n_panels, kind, exchange, q)) for r in range(world)]
    for p in procs:
        p.start()
    try:
        got = [q.get(timeout=240) for _ in range(world)]
    finally:
        for p in procs:
            p.join(timeout=60)
            if p.is_alive():
                p.kill()
    assert all(p.exitcode == 0 for p in procs)
    for rank, same, streams, nans in sorted(got):
        assert streams, "the GPU run must use the three-stream pipeline"
        assert nans == 0, f"rank {rank}: {nans} elements of C never written"
        assert same, f"rank {rank}: gathered C differs from the single-operator C"


# ---- the same check through the C ABI of include/mi_spmm_dist.h (hpc_amd/libmi_spmm_dist.so) ----
def _native_worker(rank, world, port, M, n_loc, n_panels, kind, exchange, q):
    """exchange "peer2d": every rank maps the other ranks' C_full through HIP IPC (here: other processes on the same
    GPU) and pushes its column block into them with strided 2-D copies; no staging, no re-layout kernel, the rank's own
    block is computed straight into its C_full.  exchange "peer_store": no copies either -- the kernels' epilogues store
    every finished row segment into the local C_full and into every peer's (mi_spmm_run_rows_multi).  RCCL cannot put two ranks on one device, so the end-of-step barrier
    is the caller's (gloo) instead of the library's one-element all-reduce."""
    import torch
    import torch.distributed as dist

    from hpc_amd import CSR, SpMMOpt, synth
    from hpc_amd.dist import NativeColumnShardedSpMM, ShardLayout

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        if kind == "powerlaw":
            ptr, idx = synth.csr_powerlaw(M, 12.0, 3000, seed=21, force_max=True)
        else:
            ptr, idx = synth.csr_uniform(M, 0, 40, seed=5)
        vals = synth.normal_f32(idx.size, 6)
        blocks = [synth.normal_f32(M * n_loc, synth.SEED_B, stream=r).reshape(M, n_loc) for r in range(world)]
        d_ptr, d_idx, d_val = (torch.from_numpy(a).to(dev) for a in (ptr, idx, vals))
        B_loc = torch.from_numpy(blocks[rank]).to(dev)
        C_full = torch.full((M, n_loc * world), float("nan"), device=dev)
        op = SpMMOpt(CSR(M, idx.size, d_ptr, d_idx, d_val), n_loc)
        op.preprocess(B_loc, C_full)
        sh = NativeColumnShardedSpMM(op, ShardLayout(M, n_loc, world, rank), n_panels=n_panels, exchange=exchange)
        from hpc_amd.dist import MiSpmmDistError

        def host_barrier():
            torch.cuda.synchronize()
            dist.barrier()

        if exchange == "ipc_pull":
            # the allgather schedule's staging / double buffer / re-layout path with real ranks: blocks pulled out of the peers'
            # staging buffers through HIP IPC, the two barriers of every panel supplied by the host (no RCCL on a shared GPU)
            sh.set_peer_staging()
            try:
                sh.run(B_loc, C_full)
                refused = False
            except MiSpmmDistError as e:
                refused = e.code == -3
            assert refused, "ipc_pull at world > 1 with neither communicator nor host barrier must be refused"
            sh.set_host_barrier(host_barrier)
            assert sh.get_option("has_comm") == 0 and sh.get_option("staging_bytes") > 0
        else:
            sh.set_peers(C_full)
            assert sh.get_option("has_peers") == 1 and sh.get_option("has_comm") == 0
            # no communicator: the step's two device-side barriers do not exist, so the library refuses to run it ...
            try:
                sh.run(B_loc, C_full)
                refused = False
            except MiSpmmDistError as e:
                refused = e.code == -3           # MI_SPMM_ESTATE
            assert refused, "peer2d at world > 1 without a communicator must be refused"
            sh.set_option("external_barrier", 1)     # ... unless the caller says it brackets every step itself (below)

        def barrier():
            torch.cuda.synchronize()
            dist.barrier()

        for _ in range(3):
            barrier()                     # nobody still reads the C_full we are about to write into
            sh.run(B_loc, C_full)
            barrier()                     # every rank's pushes have landed
        if exchange != "ipc_pull":
            # With a host barrier registered the step synchronises ITSELF (both barriers through the callback): no promise from
            # the caller needed, no brackets around the steps.  NaN-poison first: the result compared below is this mode's.
            sh.set_option("external_barrier", 0)
            sh.set_host_barrier(host_barrier)
            C_full.fill_(float("nan"))
            for _ in range(2):
                sh.run(B_loc, C_full)
            barrier()
        B_all = torch.from_numpy(np.ascontiguousarray(np.concatenate(blocks, axis=1))).to(dev)
        C_one = torch.empty(M, n_loc * world, device=dev)
        one = SpMMOpt(CSR(M, idx.size, d_ptr, d_idx, d_val), n_loc * world)
        one.set_option("long_row_threshold", op.get_option("long_row_threshold"))
        one.preprocess(B_all, C_one)
        one.run(B_all, C_one)
        torch.cuda.synchronize()
        same = bool(torch.equal(C_full.view(torch.int32), C_one.view(torch.int32)))
        q.put((rank, same, int(sh.get_option("staging_bytes")), int(torch.isnan(C_full).sum().item())))
        barrier()                         # keep C_full mapped until every rank has compared
        del sh
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("exchange", ["peer2d", "peer_store", "ipc_pull"])
@pytest.mark.parametrize("world,M,n_loc,n_panels,kind", [(2, 40000, 128, 8, "uniform"), (3, 30011, 64, 5, "powerlaw")])
def test_native_peer2d_exchange_between_ranks_sharing_one_gpu(world, M, n_loc, n_panels, kind, exchange):
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_native_worker, args=(r, world, port, M, n_loc, n_panels, kind, exchange, q)) for r in range(world)]
    for p in procs:
        p.start()
    try:
        got = [q.get(timeout=240) for _ in range(world)]
    finally:
        for p in procs:
            p.join(timeout=60)
            if p.is_alive():
                p.kill()
    assert all(p.exitcode == 0 for p in procs)
    for rank, same, staging, nans in sorted(got):
        assert (staging == 0) == (exchange != "ipc_pull"), "peer2d / peer_store must not allocate staging; ipc_pull must"
        assert nans == 0, f"rank {rank}: {nans} elements of C never written"
        assert same, f"rank {rank}: C differs from the single-operator C"


@pytest.mark.parametrize("exchange", ["peer2d", "peer_store"])
@pytest.mark.parametrize("world,kind", [(2, "uniform"), (4, "powerlaw"), (8, "powerlaw")])
def test_ranks_in_one_process_through_peer_pointers(exchange, world, kind):
    """mi_spmm_dist_set_peer_pointers: a host that drives every rank from one process hands the peers' C_full in as plain device
    pointers (no IPC).  Up to eight rank objects on the one GPU, hub rows included: every C_full equals the single-operator C bit for
    bit.  World 8 is the north star's node: seven peers fill the kernels' PeerOut table (peer_store) and the seven push streams
    (peer2d) -- the only place the full table is exercised without an 8-GPU node."""
    import torch

    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import dist_inprocess

    dev = torch.device("cuda:0")
    shs, B, C, C_one, keep, nnz = dist_inprocess.build(world, 20011, 64, exchange, dev, kind=kind)
    for _ in range(2):
        dist_inprocess.step(shs, B, C)
    torch.cuda.synchronize()
    for r, c in enumerate(C):
        assert not torch.isnan(c).any(), r
        assert torch.equal(c.view(torch.int32), C_one.view(torch.int32)), r


def test_export_refuses_allocation_sizes_that_hang_hip_ipc():
    """hipIpcOpenMemHandle never returns for an allocation whose size has bit 31 set (profiles/r04_ipc_open_sizes.txt: the N = 4
    rehearsal of bench.py's launch line hung on its 2 GiB C_full).  The exporter refuses such an allocation at once
    (MI_SPMM_EUNSUPPORTED) instead of leaving its peers in a call that never returns; hpc_amd.dist.alloc_c_full allocates the same
    M x N_total in a block the peers can open; "ipc_any_size" switches the check off."""
    import ctypes

    import torch

    from hpc_amd import CSR, SpMMOpt, synth
    from hpc_amd.dist import IPC_HANDLE_BYTES, NativeColumnShardedSpMM, ShardLayout, alloc_c_full, ipc_exportable_bytes, load_dist

    dev = torch.device("cuda", 0)
    M, n_loc = 4096, 32
    ptr, idx = synth.csr_uniform(M, 0, 20, seed=3)
    vals = synth.normal_f32(idx.size, 4)
    d_ptr, d_idx, d_val = (torch.from_numpy(a).to(dev) for a in (ptr, idx, vals))
    B = torch.zeros(M, n_loc, device=dev)
    C = torch.zeros(M, n_loc, device=dev)
    op = SpMMOpt(CSR(M, idx.size, d_ptr, d_idx, d_val), n_loc)
    op.preprocess(B, C)
    sh = NativeColumnShardedSpMM(op, ShardLayout(M, n_loc, 1, 0), n_panels=2, exchange="peer2d")
    lib = load_dist()
    h = (ctypes.c_char * IPC_HANDLE_BYTES)()
    off = ctypes.c_int64(-1)

    def export(t):
        return lib.mi_spmm_dist_export_c(sh._d, ctypes.c_void_p(t.data_ptr()), h, ctypes.byref(off))

    torch.cuda.empty_cache()
    bad = torch.empty(1 << 29, dtype=torch.float32, device=dev)              # 2 GiB: C1's C_full on four GPUs
    assert export(bad) == -5
    sh.set_option("ipc_any_size", 1)
    assert export(bad) == 0 and off.value == 0                               # the handle alone is harmless; opening it is what hangs
    sh.set_option("ipc_any_size", 0)
    del bad
    torch.cuda.empty_cache()
    good = alloc_c_full(1 << 20, 512, dev)                                   # the same 2 GiB of C inside a 4 GiB block
    assert good.shape == (1 << 20, 512) and good.is_contiguous()
    assert ipc_exportable_bytes(4 * good.numel()) == 1 << 32
    assert export(good) == 0 and off.value == 0
    small = alloc_c_full(1000, 256, dev, fill=float("nan"))                  # sizes that need no padding are allocated as they are
    assert torch.isnan(small).all() and export(small) == 0
    del good, small
    torch.cuda.empty_cache()
