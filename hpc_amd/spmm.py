"""Host-side mirror of the reference's PA4 operator interface, over the C ABI.

Same names, argument meaning and error behaviour as the reference so that a
test written against it reads like PA4/workspace/test/test_spmm.cu:

    g = CSR(num_v, num_e, ptr, idx, val)            # include/util.h:120-129
    spmmer = SpMMOpt(g, feat_in)                    # include/spmm_opt.h:12-17
    spmmer.preprocess(vin, vout)                    # src/spmm_opt.cu:37-69
    spmmer.run(vin, vout)                           # src/spmm_opt.cu:71-75
    bad = valid(vout, vout_ref, num_v * feat_in)    # src/valid.cu:36-51

All arrays are torch tensors on the HIP device ("cuda" in torch-ROCm): torch is
only the owner of device memory and streams here.  Pointers, sizes and the
current stream handle are what cross into libmi_spmm.so.
"""
import ctypes as C

from . import _lib


class MiSpmmError(RuntimeError):
    """A non-zero status from the C ABI.  The reference aborts the process on
    any device error (include/util.h:63-84); a Python host raises instead."""

    def __init__(self, code, where):
        self.code = int(code)
        msg = _lib.load().mi_spmm_strerror(self.code)
        super().__init__(f"{where}: [{self.code}] {msg.decode() if msg else 'unknown'}")


def _check(code, where):
    if code != 0:
        raise MiSpmmError(code, where)


def _ptr(t):
    return C.c_void_p(0 if t is None else t.data_ptr())


def _stream():
    import torch

    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _require_device(name, t, dtype):
    import torch

    if t is None:
        return
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise TypeError(f"{name} must be a device tensor (the reference passes device pointers)")
    if t.dtype != dtype:
        raise TypeError(f"{name} must be {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise ValueError(f"{name} must be contiguous")


class CSR:
    """Non-owning view of device CSR arrays -- `struct CSR`, include/util.h:120-129.

    ptr: int32[num_v+1], idx: int32[num_e], val: float32[num_e]; 0-based."""

    def __init__(self, out_num_v, out_num_e, outptr, outidx, outval):
        import torch

        _require_device("ptr", outptr, torch.int32)
        _require_device("idx", outidx, torch.int32)
        _require_device("val", outval, torch.float32)
        self.num_v = int(out_num_v)
        self.num_e = int(out_num_e)
        self.ptr = outptr
        self.idx = outidx
        self.val = outval


class SpMM:
    """Abstract operator -- `class SpMM`, include/spmm_base.h:8-46."""

    def __init__(self, g_or_ptr, *rest):
        # SpMM(CSR *g, int feat_in)  or  SpMM(ptr, idx, num_v, num_e, feat_in) (d_val stays NULL)
        if isinstance(g_or_ptr, CSR):
            (feat_in,) = rest
            g = g_or_ptr
            self.d_ptr, self.d_idx, self.d_val = g.ptr, g.idx, g.val
            self.num_v, self.num_e = g.num_v, g.num_e
        else:
            d_idx, num_v, num_e, feat_in = rest
            self.d_ptr, self.d_idx, self.d_val = g_or_ptr, d_idx, None
            self.num_v, self.num_e = int(num_v), int(num_e)
        self.feat_in = int(feat_in)

    def set_feat(self, given_feat):
        self.feat_in = int(given_feat)

    def preprocess(self, vin, vout):
        raise NotImplementedError

    def run(self, vin, vout):
        raise NotImplementedError


class SpMMOpt(SpMM):
    """The MI355X replacement for the reference's `SpMMOpt`
    (include/spmm_opt.h:12-29, src/spmm_opt.cu:37-75): same constructor and
    preprocess/run contract, backed by the hand-written gfx950 kernels.

    Differences that are deliberate: run() OVERWRITES vout with
    A*vin (the student kernel accumulates with atomicAdd and is only right on
    the first call after preprocess), and is idempotent."""

    def __init__(self, g_or_ptr, *rest, num_cols=None):
        super().__init__(g_or_ptr, *rest)
        self._lib = _lib.load()  # raises if the HIP library is not built: no fallback
        self._h = C.c_void_p(None)
        self._num_cols = int(num_cols) if num_cols is not None else self.num_v  # reference: square
        _check(
            self._lib.mi_spmm_create(
                C.byref(self._h), _ptr(self.d_ptr), _ptr(self.d_idx), _ptr(self.d_val),
                self.num_v, self._num_cols, self.num_e, self.feat_in,
            ),
            "mi_spmm_create",
        )

    def __del__(self):
        h = getattr(self, "_h", None)
        if h is not None and h.value:
            self._lib.mi_spmm_destroy(h)
            self._h = C.c_void_p(None)

    def set_feat(self, given_feat):
        super().set_feat(given_feat)
        _check(self._lib.mi_spmm_set_feat(self._h, self.feat_in), "mi_spmm_set_feat")

    def set_option(self, key, value):
        _check(self._lib.mi_spmm_set_option(self._h, key.encode(), int(value)), f"mi_spmm_set_option({key})")

    def get_option(self, key):
        v = C.c_int64(0)
        _check(self._lib.mi_spmm_get_option(self._h, key.encode(), C.byref(v)), f"mi_spmm_get_option({key})")
        return v.value

    def _check_dense(self, vin, vout, ldb=None, ldc=None):
        import torch

        _require_device("vin", vin, torch.float32)
        _require_device("vout", vout, torch.float32)
        ldb = self.feat_in if ldb is None else int(ldb)
        ldc = self.feat_in if ldc is None else int(ldc)
        if self._num_cols and self.feat_in:
            need_b = (self._num_cols - 1) * ldb + self.feat_in
            if vin is None or vin.numel() < need_b:
                raise ValueError(f"vin holds {0 if vin is None else vin.numel()} floats, need {need_b}")
        if self.num_v and self.feat_in:
            need_c = (self.num_v - 1) * ldc + self.feat_in
            if vout is None or vout.numel() < need_c:
                raise ValueError(f"vout holds {0 if vout is None else vout.numel()} floats, need {need_c}")
        return ldb, ldc

    def preprocess(self, vin, vout):
        self._check_dense(vin, vout)
        _check(self._lib.mi_spmm_preprocess(self._h, _ptr(vin), _ptr(vout)), "mi_spmm_preprocess")

    def run(self, vin, vout):
        self._check_dense(vin, vout)
        _check(self._lib.mi_spmm_run(self._h, _ptr(vin), _ptr(vout), _stream()), "mi_spmm_run")

    def run_ld(self, vin, ldb, vout, ldc):
        """run() on a column slice of wider row-major B / C (row pitches in floats)."""
        ldb, ldc = self._check_dense(vin, vout, ldb, ldc)
        _check(self._lib.mi_spmm_run_ld(self._h, _ptr(vin), ldb, _ptr(vout), ldc, _stream()), "mi_spmm_run_ld")

    def run_rows(self, vin, ldb, vout, ldc, row_begin, row_end):
        """run_ld() restricted to rows [row_begin, row_end) (vout = base of the full-height C)."""
        ldb, ldc = self._check_dense(vin, vout, ldb, ldc)
        _check(
            self._lib.mi_spmm_run_rows(self._h, _ptr(vin), ldb, _ptr(vout), ldc, int(row_begin), int(row_end), _stream()),
            "mi_spmm_run_rows",
        )


def valid(y, y2, num):
    """`valid(y, y2, num)` of src/valid.cu:22-51: the number of elements the
    reference's validator counts as wrong (float: |(y-y2)/y| > 1e-2; int: y != y2)."""
    import torch

    lib = _lib.load()
    bad = C.c_int64(-1)
    if y.dtype == torch.float32:
        _require_device("y", y, torch.float32)
        _require_device("y2", y2, torch.float32)
        _check(lib.mi_spmm_valid_float(_ptr(y), _ptr(y2), int(num), C.byref(bad), _stream()), "mi_spmm_valid_float")
    else:
        _require_device("y", y, torch.int32)
        _require_device("y2", y2, torch.int32)
        _check(lib.mi_spmm_valid_int(_ptr(y), _ptr(y2), int(num), C.byref(bad), _stream()), "mi_spmm_valid_int")
    return bad.value


def count_bitdiff(a, b):
    """(number of fp32 elements whose bit patterns differ, max |a-b|), computed on the device."""
    import torch

    lib = _lib.load()
    _require_device("a", a, torch.float32)
    _require_device("b", b, torch.float32)
    if a.numel() != b.numel():
        raise ValueError("size mismatch")
    n = C.c_int64(-1)
    m = C.c_float(0.0)
    _check(lib.mi_spmm_count_bitdiff(_ptr(a), _ptr(b), a.numel(), C.byref(n), C.byref(m), _stream()), "mi_spmm_count_bitdiff")
    return n.value, m.value


def unpack_gathered(staging, C_out, rows, n_ranks, n_loc, ldc):
    """staging[G][rows][n_loc] (what an all-gather delivers) -> row-major C[rows][ldc]."""
    lib = _lib.load()
    _check(
        lib.mi_spmm_unpack_gathered(_ptr(staging), _ptr(C_out), int(rows), int(n_ranks), int(n_loc), int(ldc), _stream()),
        "mi_spmm_unpack_gathered",
    )


def allocate(num, tensor_ptr=None, random=True, device="cuda", seed=123, subsequence=0):
    """`allocate<float>(num, &tensor_ptr, random)` of include/data.h:24-37: a device buffer of
    roundup512(num) floats, filled with N(0, 0.1) when `random` (seed 123 as test/main.cpp:20)."""
    import torch

    n = (int(num) + 511) // 512 * 512
    t = torch.empty(n, dtype=torch.float32, device=device)
    if random:
        fill_normal(t, seed=seed, subsequence=subsequence, mean=0.0, stddev=0.1)
    if tensor_ptr is not None:
        tensor_ptr.append(t)
    return t


def fill_normal(t, seed=123, subsequence=0, mean=0.0, stddev=0.1):
    import torch

    _require_device("t", t, torch.float32)
    _check(_lib.load().mi_spmm_fill_normal(_ptr(t), t.numel(), int(seed), int(subsequence), float(mean), float(stddev), _stream()),
           "mi_spmm_fill_normal")
    return t


def fill_philox_u32(t, seed=123, subsequence=0):
    import torch

    _require_device("t", t, torch.int32)
    _check(_lib.load().mi_spmm_fill_philox_u32(_ptr(t), t.numel(), int(seed), int(subsequence), _stream()), "mi_spmm_fill_philox_u32")
    return t
