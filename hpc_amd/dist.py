"""Column-sharded multi-GPU SpMM: C[:, g*n : (g+1)*n] = A * B[:, g*n : (g+1)*n] on GPU g,
then an all-gather of the C column blocks (RCCL over xGMI) so every rank holds row-major C.

The reference has no multi-GPU code; what must hold is "same C as one GPU":
column blocks are independent, every element is produced by the same kernel in the same
order, so the gathered C is bit-identical to the 1-GPU C.

One process per GPU (torch.distributed, backend "nccl" = RCCL).  A is replicated, each rank
keeps its B slice resident as a contiguous K x n_loc array.  The exchange is pipelined by row
panels: while panel p's column blocks travel, panel p+1's rows are being computed:

    compute stream :  rows(p0) | rows(p1) | rows(p2) | ...
    comm stream    :           | gather(p0) | gather(p1) | gather(p2) | ...      (back to back: link-bound)
    unpack stream  :                        | unpack(p0) | unpack(p1) | ...

all_gather_into_tensor delivers rank-major blocks staging[G][rows][n_loc] (an all-gather
concatenates contiguous per-rank buffers -- SURVEY.md H4); `unpack` writes them into the
row-major C[rows][G*n_loc] the interface promises (a HIP kernel on the GPU path).

Two exchange schedules fill the same staging layout:
  "allgather"  the library collective (RCCL chooses rings/trees over the xGMI mesh);
  "direct"     all-pairs: one grouped launch of G-1 sends of this rank's block and G-1 receives
               straight into staging[peer] -- on the fully connected 8-GPU node every pair has
               its own link, so all 7 links of a GPU carry one block each at the same time.
`tune()` times both on the first panel's real sizes and keeps the faster (max over ranks, so all
ranks agree); any failure of "direct" leaves "allgather".

The class is device-agnostic host logic: the local operator and the unpack step are passed in.
Product use (bench.py): op = hpc_amd.SpMMOpt, unpack = hpc_amd.spmm.unpack_gathered.  The
world_size-2 gloo tests on CPU drive the same schedule with test doubles.
"""
from dataclasses import dataclass
from typing import Callable, List, Optional, Tuple


def column_block(N_total: int, world: int, rank: int) -> Tuple[int, int]:
    """[col0, col1) owned by `rank`: equal blocks (N_total must divide by world, like the
    north-star configuration N = 1024 over 8 GPUs)."""
    if N_total % world:
        raise ValueError(f"N={N_total} does not divide over {world} ranks")
    n = N_total // world
    return rank * n, (rank + 1) * n


def row_panels(M: int, n_panels: int, align: int = 256) -> List[Tuple[int, int]]:
    """Split [0, M) into at most n_panels contiguous panels whose sizes are multiples of `align`
    rows (except the last).  The first panel is never empty when M > 0."""
    if M <= 0:
        return []
    n_panels = max(1, int(n_panels))
    per = -(-M // n_panels)
    per = -(-per // align) * align
    out = []
    r = 0
    while r < M:
        e = min(M, r + per)
        out.append((r, e))
        r = e
    return out


@dataclass
class ShardLayout:
    M: int
    n_loc: int
    world: int
    rank: int

    @property
    def N_total(self) -> int:
        return self.n_loc * self.world

    @property
    def col0(self) -> int:
        return self.rank * self.n_loc


class ColumnShardedSpMM:
    """run(B_loc, C_loc, C_full): C_loc[M][n_loc] = A * B_loc (this rank's block, scratch the
    caller owns), C_full[M][G*n_loc] = all blocks, row-major, identical on every rank."""

    EXCHANGES = ("allgather", "direct")

    def __init__(self, op, layout: ShardLayout, unpack: Callable, n_panels: int = 8,
                 group=None, use_streams: Optional[bool] = None, force_collective: bool = False,
                 exchange: str = "allgather"):
        if exchange not in self.EXCHANGES:
            raise ValueError(f"exchange must be one of {self.EXCHANGES}")
        self.exchange = exchange
        self.tuning = None       # {"allgather": ms, "direct": ms | None} after tune()
        self.op = op
        self.layout = layout
        self.unpack = unpack
        self.group = group
        self.panels = row_panels(layout.M, n_panels)
        self.use_streams = use_streams
        self.force_collective = force_collective  # run the gather pipeline even at world == 1 (rehearsal)
        self._staging = None
        self._streams = None

    def _ensure_buffers(self, like):
        import torch

        L = self.layout
        rows_max = max((e - b for b, e in self.panels), default=0)
        need = L.world * rows_max * L.n_loc
        if self._staging is None or self._staging[0].numel() < need or self._staging[0].device != like.device:
            self._staging = [torch.empty(need, dtype=like.dtype, device=like.device) for _ in range(2)]
        if self.use_streams is None:
            self.use_streams = like.is_cuda
        if self.use_streams and self._streams is None:
            # exchange and unpack at high priority: they must not queue behind a compute kernel that fills every CU
            self._streams = (torch.cuda.Stream(device=like.device, priority=-1), torch.cuda.Stream(device=like.device, priority=-1))

    def _peer(self, r):
        import torch.distributed as dist

        return r if self.group is None else dist.get_global_rank(self.group, r)

    def _exchange(self, stage, src, rows):
        """Fill stage[G][rows][n_loc] with every rank's `src` block (on the current stream)."""
        import torch.distributed as dist

        L = self.layout
        if self.exchange == "allgather" or L.world == 1:
            dist.all_gather_into_tensor(stage, src, group=self.group)
            return
        blk = rows * L.n_loc
        ops = []
        for d in range(1, L.world):          # rank r sends to r+d while it receives from r-d: every step is a perfect matching
            to, frm = (L.rank + d) % L.world, (L.rank - d) % L.world
            ops.append(dist.P2POp(dist.isend, src, self._peer(to), group=self.group))
            ops.append(dist.P2POp(dist.irecv, stage[frm * blk: (frm + 1) * blk], self._peer(frm), group=self.group))
        stage[L.rank * blk: (L.rank + 1) * blk].copy_(src)
        for w in dist.batch_isend_irecv(ops):
            w.wait()

    def tune(self, C_loc, reps: int = 3):
        """Time both exchange schedules on the first panel's sizes and keep the faster one.
        Collective: every rank must call it.  Leaves self.exchange set identically on all ranks."""
        import time
        import torch
        import torch.distributed as dist

        L = self.layout
        if L.world == 1 or not self.panels:
            return self.exchange
        self._ensure_buffers(C_loc)
        r0, r1 = self.panels[0]
        rows = r1 - r0
        src = C_loc.view(-1)[r0 * L.n_loc: r1 * L.n_loc]
        stage = self._staging[0][: L.world * rows * L.n_loc]
        cuda = C_loc.is_cuda
        result = {}
        where = C_loc.device if cuda else "cpu"

        def agree(x, op):
            v = torch.tensor([x], dtype=torch.float64, device=where)
            dist.all_reduce(v, op=op, group=self.group)
            return float(v.item())

        for name in self.EXCHANGES:
            self.exchange = name
            ok = 1.0
            try:
                self._exchange(stage, src, rows)           # first use: communicator / connection set-up
                if cuda:
                    torch.cuda.synchronize()
            except Exception:                              # e.g. a backend without p2p for this tensor type
                if name == "allgather":
                    raise
                ok = 0.0
            if agree(ok, dist.ReduceOp.MIN) < 1.0:         # some rank could not: nobody times it
                result[name] = None
                continue
            dist.barrier(group=self.group)
            t = time.perf_counter()
            for _ in range(reps):
                self._exchange(stage, src, rows)
            if cuda:
                torch.cuda.synchronize()
            result[name] = agree((time.perf_counter() - t) * 1e3 / reps, dist.ReduceOp.MAX)   # the slowest rank's time
        self.tuning = result
        self.exchange = "direct" if result["direct"] is not None and result["direct"] < 0.97 * result["allgather"] else "allgather"
        return self.exchange

    def run(self, B_loc, C_loc, C_full):
        import torch
        import torch.distributed as dist

        L = self.layout
        if L.world == 1 and not self.force_collective:
            # nothing to exchange: the local block IS C
            self.op.run_rows(B_loc, L.n_loc, C_full, L.N_total, 0, L.M)
            return
        self._ensure_buffers(C_loc)
        if not self.use_streams:
            for p, (r0, r1) in enumerate(self.panels):
                rows = r1 - r0
                self.op.run_rows(B_loc, L.n_loc, C_loc, L.n_loc, r0, r1)
                src = C_loc.view(-1)[r0 * L.n_loc: r1 * L.n_loc]
                stage = self._staging[p & 1][: L.world * rows * L.n_loc]
                self._exchange(stage, src, rows)
                self.unpack(stage, C_full.view(-1)[r0 * L.N_total:], rows, L.world, L.n_loc, L.N_total)
            return
        comm, post = self._streams
        main = torch.cuda.current_stream()
        comm.wait_stream(main)   # earlier work on the caller's stream (previous step's readers of C_full, staging)
        post.wait_stream(main)
        unpacked = [None, None]  # per staging buffer: event after the unpack that last read it
        for p, (r0, r1) in enumerate(self.panels):
            rows = r1 - r0
            self.op.run_rows(B_loc, L.n_loc, C_loc, L.n_loc, r0, r1)
            computed = torch.cuda.Event()
            computed.record(main)
            src = C_loc.view(-1)[r0 * L.n_loc: r1 * L.n_loc]
            stage = self._staging[p & 1][: L.world * rows * L.n_loc]
            with torch.cuda.stream(comm):
                comm.wait_event(computed)
                if unpacked[p & 1] is not None:
                    comm.wait_event(unpacked[p & 1])     # the buffer's previous contents have been consumed
                self._exchange(stage, src, rows)
                gathered = torch.cuda.Event()
                gathered.record(comm)
            with torch.cuda.stream(post):
                post.wait_event(gathered)
                self.unpack(stage, C_full.view(-1)[r0 * L.N_total:], rows, L.world, L.n_loc, L.N_total)
                done = torch.cuda.Event()
                done.record(post)
                unpacked[p & 1] = done
        main.wait_stream(comm)
        main.wait_stream(post)


# ---------------------------------------------------------------------------------------------------
# The product path on GPUs: the same step behind the C ABI of include/mi_spmm_dist.h
# (hpc_amd/libmi_spmm_dist.so: streams, RCCL calls and strided peer copies in C++, no torch in its
# signatures).  torch.distributed is only the bootstrap here -- it carries the 128-byte RCCL unique id
# and the HIP IPC handles between the ranks, as MPI_Bcast / MPI_Allgather would in a C++ host.
# ColumnShardedSpMM above stays as the device-agnostic statement of the schedule that the world_size-2
# gloo tests drive on CPU tensors.
# ---------------------------------------------------------------------------------------------------
import ctypes as _C
import os as _os

_DIST_LIB = None
_DIST_PATH = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "libmi_spmm_dist.so")
_P = _C.c_void_p
_DIST_SIGNATURES = {
    "mi_spmm_dist_create": (_C.c_int, [_C.POINTER(_P), _P, _C.c_int32, _C.c_int32, _C.c_int32, _C.c_int32, _C.c_int32]),
    "mi_spmm_dist_destroy": (_C.c_int, [_P]),
    "mi_spmm_dist_unique_id": (_C.c_int, [_P]),
    "mi_spmm_dist_comm_init": (_C.c_int, [_P, _P]),
    "mi_spmm_dist_set_comm": (_C.c_int, [_P, _P]),
    "mi_spmm_dist_export_c": (_C.c_int, [_P, _P, _P, _C.POINTER(_C.c_int64)]),
    "mi_spmm_dist_ipc_exportable_bytes": (_C.c_int64, [_C.c_int64]),
    "mi_spmm_dist_set_peers": (_C.c_int, [_P, _P, _P, _C.POINTER(_C.c_int64)]),
    "mi_spmm_dist_link_probe": (_C.c_int, [_P, _P, _C.c_int64, _P, _P, _P, _P, _P]),
    "mi_spmm_dist_set_peer_pointers": (_C.c_int, [_P, _P, _P]),
    "mi_spmm_dist_export_staging": (_C.c_int, [_P, _P, _C.POINTER(_C.c_int64)]),
    "mi_spmm_dist_set_peer_staging": (_C.c_int, [_P, _P, _C.POINTER(_C.c_int64)]),
    "mi_spmm_dist_set_host_barrier": (_C.c_int, [_P, _P, _P]),
    "mi_spmm_dist_set_option": (_C.c_int, [_P, _C.c_char_p, _C.c_int64]),
    "mi_spmm_dist_get_option": (_C.c_int, [_P, _C.c_char_p, _C.POINTER(_C.c_int64)]),
    "mi_spmm_dist_run": (_C.c_int, [_P, _P, _P, _P]),
    "mi_spmm_dist_run_compute_only": (_C.c_int, [_P, _P, _P, _P]),
    "mi_spmm_dist_run_exchange_only": (_C.c_int, [_P, _P, _P]),
    "mi_spmm_dist_strerror": (_C.c_char_p, [_C.c_int]),
}
UNIQUE_ID_BYTES = 128
IPC_HANDLE_BYTES = 64
EXCHANGE_CODES = {"allgather": 0, "direct": 1, "peer2d": 2, "peer_store": 3, "ipc_pull": 4}
_BARRIER_FN = _C.CFUNCTYPE(None, _C.c_void_p)


class MiSpmmDistError(RuntimeError):
    def __init__(self, code, where):
        self.code = int(code)
        msg = load_dist().mi_spmm_dist_strerror(self.code)
        super().__init__(f"{where}: [{self.code}] {msg.decode() if msg else 'unknown'}")


def load_dist():
    """libmi_spmm_dist.so, every symbol of include/mi_spmm_dist.h bound; raises when it is not built (no fallback)."""
    global _DIST_LIB
    if _DIST_LIB is not None:
        return _DIST_LIB
    if not _os.path.exists(_DIST_PATH):
        raise RuntimeError(f"{_DIST_PATH} not found: build it with `make -C hpc_amd/csrc` (or __graft_entry__.build())")
    from . import _lib

    _lib.load()                                   # libmi_spmm.so first (the dist library links it)
    lib = _C.CDLL(_DIST_PATH)
    for name, (res, args) in _DIST_SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _DIST_LIB = lib
    return lib


def _dcheck(code, where):
    if code != 0:
        raise MiSpmmDistError(code, where)


def ipc_exportable_bytes(nbytes):
    """mi_spmm_dist_ipc_exportable_bytes: the allocation size to use for a buffer of nbytes that peers will map through HIP IPC
    (hipIpcOpenMemHandle hangs on allocations whose size has bit 31 set: such sizes go up to the next multiple of 4 GiB)."""
    return int(load_dist().mi_spmm_dist_ipc_exportable_bytes(int(nbytes)))


def alloc_c_full(M, n_total, device, fill=None):
    """A row-major M x n_total fp32 C_full whose ALLOCATION the peers can map (peer2d / peer_store): a view of the front of a flat
    buffer of ipc_exportable_bytes(4 * M * n_total) bytes, in a block of its own (the caching allocator may otherwise hand out
    part of a larger cached block of any size).  C1 on four GPUs is exactly 2 GiB: allocated as 4 GiB."""
    import torch

    need = 4 * int(M) * int(n_total)
    nbytes = ipc_exportable_bytes(need)
    if nbytes != need:
        torch.cuda.empty_cache()             # no cached block of another size to be re-used for this request
    flat = torch.empty(nbytes // 4, dtype=torch.float32, device=device)
    C = flat[: int(M) * int(n_total)].view(int(M), int(n_total))
    if fill is not None:
        C.fill_(fill)
    return C


def L_bytes(layout):
    """floats in a rank's C_full"""
    return int(layout.M) * int(layout.N_total)


class NativeColumnShardedSpMM:
    """The column-sharded step through the C ABI.  op: a PREPROCESSED hpc_amd.SpMMOpt for this rank's n_loc columns.

        sh = NativeColumnShardedSpMM(op, ShardLayout(M, n_loc, world, rank), n_panels=8, exchange="allgather")
        sh.init_comm()                    # collective over torch.distributed (any backend): RCCL communicator of our own
        sh.set_peers(C_full)              # collective, only for exchange="peer2d": HIP IPC handles of every rank's C_full
        sh.run(B_loc, C_full)             # asynchronous on torch's current stream
    """

    def __init__(self, op, layout: ShardLayout, n_panels: int = 8, exchange: str = "allgather", group=None, rehearse: bool = False):
        self._lib = load_dist()
        self.op = op
        self.layout = layout
        self.group = group
        self._d = _P(None)
        _dcheck(self._lib.mi_spmm_dist_create(_C.byref(self._d), op._h, layout.M, layout.n_loc, layout.rank, layout.world, int(n_panels)),
                "mi_spmm_dist_create")
        self.set_exchange(exchange)
        if rehearse:
            self.set_option("rehearse", 1)
        self.has_comm = False

    def __del__(self):
        d = getattr(self, "_d", None)
        if d is not None and d.value:
            self._lib.mi_spmm_dist_destroy(d)
            self._d = _P(None)

    def set_option(self, key, value):
        _dcheck(self._lib.mi_spmm_dist_set_option(self._d, key.encode(), int(value)), f"mi_spmm_dist_set_option({key})")

    def get_option(self, key):
        v = _C.c_int64(0)
        _dcheck(self._lib.mi_spmm_dist_get_option(self._d, key.encode(), _C.byref(v)), f"mi_spmm_dist_get_option({key})")
        return v.value

    def set_exchange(self, name):
        if name not in EXCHANGE_CODES:
            raise ValueError(f"exchange must be one of {tuple(EXCHANGE_CODES)}")
        self.exchange = name
        self.set_option("exchange", EXCHANGE_CODES[name])

    def init_comm(self):
        """Collective.  Rank 0 draws the RCCL unique id, torch.distributed broadcasts its 128 bytes, every rank joins."""
        import torch.distributed as dist

        buf = (_C.c_char * UNIQUE_ID_BYTES)()
        if self.layout.world > 1:
            box = [None]
            if self.layout.rank == 0:
                _dcheck(self._lib.mi_spmm_dist_unique_id(buf), "mi_spmm_dist_unique_id")
                box[0] = bytes(buf)
            src = 0 if self.group is None else dist.get_global_rank(self.group, 0)
            dist.broadcast_object_list(box, src=src, group=self.group)
            _C.memmove(buf, box[0], UNIQUE_ID_BYTES)
        else:
            _dcheck(self._lib.mi_spmm_dist_unique_id(buf), "mi_spmm_dist_unique_id")
        _dcheck(self._lib.mi_spmm_dist_comm_init(self._d, buf), "mi_spmm_dist_comm_init")
        self.has_comm = True

    def set_peers(self, C_full):
        """Collective (exchange="peer2d").  Every rank exports its C_full; the handle table goes to the library."""
        import torch.distributed as dist

        L = self.layout
        h = (_C.c_char * IPC_HANDLE_BYTES)()
        off = _C.c_int64(0)
        # a rank that cannot export (e.g. an allocation size HIP IPC cannot open: mi_spmm_dist_ipc_exportable_bytes) still takes part
        # in the all-gather, and then EVERY rank raises: no rank is left waiting in a collective the failing one never entered
        code = self._lib.mi_spmm_dist_export_c(self._d, _P(C_full.data_ptr()), h, _C.byref(off))
        mine = (bytes(h), int(off.value)) if code == 0 else None
        if L.world > 1:
            table = [None] * L.world
            dist.all_gather_object(table, mine, group=self.group)
        else:
            table = [mine]
        _dcheck(code, "mi_spmm_dist_export_c")
        if any(t is None for t in table):
            raise MiSpmmDistError(-5, f"mi_spmm_dist_export_c on rank(s) {[q for q, t in enumerate(table) if t is None]}")
        handles = (_C.c_char * (IPC_HANDLE_BYTES * L.world))()
        offsets = (_C.c_int64 * L.world)()
        for q, (hb, ob) in enumerate(table):
            _C.memmove(_C.addressof(handles) + q * IPC_HANDLE_BYTES, hb, IPC_HANDLE_BYTES)
            offsets[q] = ob
        _dcheck(self._lib.mi_spmm_dist_set_peers(self._d, _P(C_full.data_ptr()), handles, offsets), "mi_spmm_dist_set_peers")
        self._peers_of = C_full           # keep the exported tensor alive

    def set_peer_tensors(self, C_full, all_C_full):
        """exchange "peer2d" / "peer_store" with every rank in THIS process: all_C_full[q] is rank q's C_full tensor."""
        ptrs = (_P * self.layout.world)(*[_P(t.data_ptr()) for t in all_C_full])
        _dcheck(self._lib.mi_spmm_dist_set_peer_pointers(self._d, _P(C_full.data_ptr()), ptrs), "mi_spmm_dist_set_peer_pointers")
        self._peers_of = list(all_C_full)

    LINK_TYPES = {0: "hypertransport", 1: "qpi", 2: "pcie", 3: "infiniband", 4: "xgmi", -1: "unknown"}

    def link_probe(self, C_full, nbytes=256 << 20, peer_devices=None):
        """Collective, after set_peers / set_peer_tensors, outside any timed region: what the links to the peers deliver (mi_spmm_dist_link_probe).
        C_full holds junk afterwards.  Returns {"per_peer_GBs": [...], "all_peers_GBs": x, "link_type": [...], "hops": [...], "bytes": n}."""
        W = self.layout.world
        per = (_C.c_double * W)()
        allp = _C.c_double(0.0)
        lt = (_C.c_int32 * W)()
        hp = (_C.c_int32 * W)()
        pd = (_C.c_int32 * W)(*[int(x) for x in peer_devices]) if peer_devices is not None else None
        _dcheck(self._lib.mi_spmm_dist_link_probe(self._d, _P(C_full.data_ptr()), int(nbytes), pd, per, _C.byref(allp), lt, hp), "mi_spmm_dist_link_probe")
        return {"per_peer_GBs": [round(float(x), 1) for x in per], "all_peers_GBs": round(float(allp.value), 1),
                "link_type": [self.LINK_TYPES.get(int(x), str(int(x))) for x in lt], "hops": [int(x) for x in hp],
                "bytes": int(min(nbytes, 4 * L_bytes(self.layout)))}

    def set_peer_staging(self):
        """Collective (exchange="ipc_pull").  Every rank exports its two staging buffers; the table goes to the library."""
        import torch.distributed as dist

        L = self.layout
        h = (_C.c_char * (2 * IPC_HANDLE_BYTES))()
        off = (_C.c_int64 * 2)()
        code = self._lib.mi_spmm_dist_export_staging(self._d, h, off)
        mine = (bytes(h), [int(off[0]), int(off[1])]) if code == 0 else None
        if L.world > 1:
            table = [None] * L.world
            dist.all_gather_object(table, mine, group=self.group)       # every rank takes part, whatever its own export did (see set_peers)
        else:
            table = [mine]
        _dcheck(code, "mi_spmm_dist_export_staging")
        if any(t is None for t in table):
            raise MiSpmmDistError(-5, f"mi_spmm_dist_export_staging on rank(s) {[q for q, t in enumerate(table) if t is None]}")
        handles = (_C.c_char * (2 * IPC_HANDLE_BYTES * L.world))()
        offsets = (_C.c_int64 * (2 * L.world))()
        for q, (hb, ob) in enumerate(table):
            _C.memmove(_C.addressof(handles) + q * 2 * IPC_HANDLE_BYTES, hb, 2 * IPC_HANDLE_BYTES)
            offsets[2 * q], offsets[2 * q + 1] = ob
        _dcheck(self._lib.mi_spmm_dist_set_peer_staging(self._d, handles, offsets), "mi_spmm_dist_set_peer_staging")

    def set_host_barrier(self, fn):
        """fn(): a cross-rank barrier of the host's (e.g. torch.cuda.synchronize(); dist.barrier()), called by the library
        where a step needs one and there is no communicator.  None removes it."""
        self._barrier_cb = _BARRIER_FN(lambda ctx: fn()) if fn is not None else None      # keep the thunk alive
        _dcheck(self._lib.mi_spmm_dist_set_host_barrier(self._d, _C.cast(self._barrier_cb, _P) if self._barrier_cb else None, None),
                "mi_spmm_dist_set_host_barrier")

    @staticmethod
    def _stream():
        import torch

        return _P(torch.cuda.current_stream().cuda_stream)

    def run(self, B_loc, C_full):
        _dcheck(self._lib.mi_spmm_dist_run(self._d, _P(B_loc.data_ptr()), _P(C_full.data_ptr()), self._stream()), "mi_spmm_dist_run")

    def run_compute_only(self, B_loc, C_full):
        _dcheck(self._lib.mi_spmm_dist_run_compute_only(self._d, _P(B_loc.data_ptr()), _P(C_full.data_ptr()), self._stream()),
                "mi_spmm_dist_run_compute_only")

    def run_exchange_only(self, C_full):
        _dcheck(self._lib.mi_spmm_dist_run_exchange_only(self._d, _P(C_full.data_ptr()), self._stream()), "mi_spmm_dist_run_exchange_only")
