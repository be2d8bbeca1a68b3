"""ctypes front end of the oracle -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import
this module.  hpc_amd/ never does (tests/test_abi_symbols.py::test_product_never_touches_the_oracle checks).

  liboracle.so                      CPU restatement (oracle/spmm_oracle.c)
  _ref/libspmm_ref_gfx950.so        the reference's own kernels compiled by hipcc from
                                    /root/reference in place (oracle/Makefile `_ref`);
                                    needs a GPU to run; prebuilt file travels to the GPU box
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(_HERE, "liboracle.so")
REF_LIB = os.path.join(_HERE, "_ref", "libspmm_ref_gfx950.so")
REF_LIB_FTZ = os.path.join(_HERE, "_ref", "libspmm_ref_gfx950_ftz.so")     # the same reference kernels, hipcc -fgpu-flush-denormals-to-zero (nvcc -ftz=true)

_lib = None
_ref = None


def build(ref=True):
    """Compile the C restatement and (when /root/reference is present) oracle/_ref."""
    subprocess.check_call(["make", "-s", "-C", _HERE, "all"])
    if ref:
        subprocess.check_call(["make", "-s", "-C", _HERE, "_ref"])


def available_cores():
    """CPUs this process may actually use: min(affinity mask, cgroup quota)."""
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                if q > 0:
                    per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                    n = min(n, max(1, q // per))
        except Exception:
            pass
    env = os.environ.get("ORACLE_THREADS")
    if env:
        n = max(1, int(env))
    return n


def set_threads(n):
    lib().oracle_set_threads(int(n))


def try_native():
    """bench.py's cpu_baseline leg: rebuild the restatement with -march=native ON THE BOX IT RUNS ON
     and switch to it.  The shipped liboracle.so is built -march=x86-64-v3 in the CPU
    container so that it also loads on the GPU box's host, which understates a Zen 5 host (no AVX-512).
    Returns the compiler flags in force (the portable ones when gcc is missing or the build fails)."""
    global _lib
    portable = "gcc -O3 -march=x86-64-v3 -ffp-contract=off -fopenmp"
    out = os.path.join(_HERE, "_native", "liboracle_native.so")
    try:
        os.makedirs(os.path.dirname(out), exist_ok=True)
        subprocess.check_call(["gcc", "-O3", "-march=native", "-ffp-contract=off", "-fopenmp", "-fPIC", "-shared", "-o", out,
                               os.path.join(_HERE, "spmm_oracle.c"), "-lm"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    except Exception:
        return portable
    global LIB
    keep_lib, keep_path = _lib, LIB
    try:
        _lib, LIB = None, out
        lib()
        return "gcc -O3 -march=native -ffp-contract=off -fopenmp (built on this host)"
    except Exception:
        _lib, LIB = keep_lib, keep_path
        return portable


def lib():
    global _lib
    if _lib is None:
        os.environ.setdefault("OMP_WAIT_POLICY", "passive")
        if not os.path.exists(LIB):
            build(ref=False)
        L = C.CDLL(LIB)
        P = C.c_void_p
        L.oracle_spmm_ref.argtypes = [P, P, P, P, P, C.c_int32, C.c_int32]
        L.oracle_spmm_ref.restype = None
        L.oracle_spmm_nofma.argtypes = [P, P, P, P, P, C.c_int32, C.c_int32]
        L.oracle_spmm_nofma.restype = None
        L.oracle_spmm_omp.argtypes = [P, P, P, P, C.c_int64, P, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32]
        L.oracle_spmm_omp.restype = None
        L.oracle_spmm_ftz.argtypes = [P, P, P, P, C.c_int64, P, C.c_int64, C.c_int32, C.c_int32]
        L.oracle_spmm_ftz.restype = None
        L.oracle_spmm_f64.argtypes = [P, P, P, P, P, P, C.c_int32, C.c_int32]
        L.oracle_spmm_f64.restype = None
        L.oracle_valid_float.argtypes = [P, P, C.c_int64]
        L.oracle_valid_float.restype = C.c_int64
        L.oracle_valid_int.argtypes = [P, P, C.c_int64]
        L.oracle_valid_int.restype = C.c_int64
        L.oracle_validation_passes.argtypes = [C.c_int64, C.c_int64, C.c_int64]
        L.oracle_validation_passes.restype = C.c_int
        L.oracle_num_threads.restype = C.c_int
        L.oracle_set_threads.argtypes = [C.c_int]
        L.oracle_set_threads.restype = None
        L.oracle_set_threads(min(available_cores(), 64))
        L.oracle_fnv1a64.argtypes = [P, C.c_int64]
        L.oracle_fnv1a64.restype = C.c_uint64
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _csr(ptr, idx, val):
    ptr = np.ascontiguousarray(ptr, dtype=np.int32)
    idx = np.ascontiguousarray(idx, dtype=np.int32)
    val = np.ascontiguousarray(val, dtype=np.float32)
    return ptr, idx, val


def spmm_ref(ptr, idx, val, vin, feat=None):
    """spmm_kernel_ref restated (serial, reference loop order)."""
    ptr, idx, val = _csr(ptr, idx, val)
    vin = np.ascontiguousarray(vin, dtype=np.float32)
    num_v = ptr.size - 1
    feat = vin.shape[1] if feat is None else feat
    out = np.empty((num_v, feat), dtype=np.float32)
    lib().oracle_spmm_ref(_p(ptr), _p(idx), _p(val), _p(vin), _p(out), num_v, feat)
    return out


def spmm_nofma(ptr, idx, val, vin):
    ptr, idx, val = _csr(ptr, idx, val)
    vin = np.ascontiguousarray(vin, dtype=np.float32)
    num_v = ptr.size - 1
    out = np.empty((num_v, vin.shape[1]), dtype=np.float32)
    lib().oracle_spmm_nofma(_p(ptr), _p(idx), _p(val), _p(vin), _p(out), num_v, vin.shape[1])
    return out


def spmm_omp(ptr, idx, val, vin, out=None, row_begin=0, row_end=-1, feat=None, ldb=None):
    """Same arithmetic, OpenMP over rows, vectorised over columns (the cpu_baseline 'port')."""
    ptr, idx, val = _csr(ptr, idx, val)
    assert vin.dtype == np.float32 and vin.flags.c_contiguous
    num_v = ptr.size - 1
    feat = vin.shape[1] if feat is None else feat
    ldb = vin.shape[1] if ldb is None else ldb
    if out is None:
        out = np.empty((num_v, feat), dtype=np.float32)
    lib().oracle_spmm_omp(_p(ptr), _p(idx), _p(val), _p(vin), ldb, _p(out), out.shape[1], num_v, feat, row_begin, row_end)
    return out


def spmm_ftz(ptr, idx, val, vin):
    """The reference's BUILD semantics (nvcc --use_fast_math => fma.rn.ftz.f32): subnormal inputs and results flushed to
    sign-preserving zeros around every fused multiply-add; stored order, +0 start."""
    ptr, idx, val = _csr(ptr, idx, val)
    vin = np.ascontiguousarray(vin, dtype=np.float32)
    num_v = ptr.size - 1
    out = np.empty((num_v, vin.shape[1]), dtype=np.float32)
    lib().oracle_spmm_ftz(_p(ptr), _p(idx), _p(val), _p(vin), vin.shape[1], _p(out), vin.shape[1], num_v, vin.shape[1])
    return out


def spmm_f64(ptr, idx, val, vin, with_abs=True):
    ptr, idx, val = _csr(ptr, idx, val)
    vin = np.ascontiguousarray(vin, dtype=np.float32)
    num_v = ptr.size - 1
    out = np.empty((num_v, vin.shape[1]), dtype=np.float64)
    ab = np.empty_like(out) if with_abs else None
    lib().oracle_spmm_f64(_p(ptr), _p(idx), _p(val), _p(vin), _p(out), _p(ab) if with_abs else None, num_v, vin.shape[1])
    return out, ab


def spmm_chunked(ptr, idx, val, vin, threshold, chunk):
    """What the device computes for rows longer than `threshold`: per-chunk fma chains
    (each exactly the oracle on the chunk) summed left to right in chunk order with fp32
    adds.  Rows at or below the threshold are the plain oracle.  Built FROM the oracle: the
    chunks are presented to oracle_spmm_omp as rows of an expanded CSR."""
    ptr, idx, val = _csr(ptr, idx, val)
    vin = np.ascontiguousarray(vin, dtype=np.float32)
    out = spmm_omp(ptr, idx, val, vin)
    deg = np.diff(ptr)
    for r in np.nonzero(deg > threshold)[0]:
        b, e = int(ptr[r]), int(ptr[r + 1])
        cuts = list(range(b, e, chunk)) + [e]
        sub_ptr = np.asarray(cuts, dtype=np.int32) - b
        parts = spmm_omp(sub_ptr, idx[b:e], val[b:e], vin)
        acc = parts[0].copy()
        for i in range(1, parts.shape[0]):
            acc = (acc + parts[i]).astype(np.float32)
        out[r] = acc
    return out


def valid_float(y, y2):
    y = np.ascontiguousarray(y, dtype=np.float32).ravel()
    y2 = np.ascontiguousarray(y2, dtype=np.float32).ravel()
    return int(lib().oracle_valid_float(_p(y), _p(y2), y.size))


def valid_int(y, y2):
    y = np.ascontiguousarray(y, dtype=np.int32).ravel()
    y2 = np.ascontiguousarray(y2, dtype=np.int32).ravel()
    return int(lib().oracle_valid_int(_p(y), _p(y2), y.size))


def validation_passes(bad, num_v, feat):
    return bool(lib().oracle_validation_passes(int(bad), int(num_v), int(feat)))


def num_threads():
    return int(lib().oracle_num_threads())


def philox_block(ctr, key):
    c = (C.c_uint32 * 4)(*ctr)
    k = (C.c_uint32 * 2)(*key)
    o = (C.c_uint32 * 4)()
    lib().oracle_philox_block(c, k, o)
    return [int(x) for x in o]


def fill_philox_u32(n, seed, subseq=0):
    out = np.empty(int(n), dtype=np.uint32)
    f = lib().oracle_fill_philox_u32
    f.argtypes = [C.c_void_p, C.c_int64, C.c_uint64, C.c_uint64]
    f.restype = None
    f(_p(out), int(n), int(seed), int(subseq))
    return out


def fill_normal(n, seed, subseq=0, mean=0.0, stddev=0.1):
    out = np.empty(int(n), dtype=np.float32)
    f = lib().oracle_fill_normal
    f.argtypes = [C.c_void_p, C.c_int64, C.c_uint64, C.c_uint64, C.c_float, C.c_float]
    f.restype = None
    f(_p(out), int(n), int(seed), int(subseq), float(mean), float(stddev))
    return out


def fnv1a64(a):
    a = np.ascontiguousarray(a)
    return int(lib().oracle_fnv1a64(_p(a), a.nbytes))


# ---- the reference's own kernels on the GPU (oracle/_ref) ---------------------------------
def ref_available():
    return os.path.exists(REF_LIB)


def ref_lib():
    global _ref
    if _ref is None:
        L = C.CDLL(REF_LIB)
        P = C.c_void_p
        L.ref_spmm_ref_run.argtypes = [P, P, P, P, P, C.c_int, C.c_int, P]
        L.ref_spmm_ref_run.restype = C.c_int
        L.ref_spmm_opt_preprocess.argtypes = [P, C.c_int, C.POINTER(P), C.POINTER(C.c_int)]
        L.ref_spmm_opt_preprocess.restype = C.c_int
        L.ref_spmm_opt_run.argtypes = [P, C.c_int, P, P, P, P, C.c_int, P]
        L.ref_spmm_opt_run.restype = C.c_int
        L.ref_free.argtypes = [P]
        L.ref_free.restype = C.c_int
        L.ref_valid_float.argtypes = [P, P, C.c_int, C.POINTER(C.c_int)]
        L.ref_valid_float.restype = C.c_int
        L.ref_valid_int.argtypes = [P, P, C.c_int, C.POINTER(C.c_int)]
        L.ref_valid_int.restype = C.c_int
        _ref = L
    return _ref


_ref_ftz = None


def ref_ftz_available():
    return os.path.exists(REF_LIB_FTZ)


def ref_kernel_run_ftz(d_ptr, d_idx, d_val, d_vin, d_vout, num_v, feat, stream=0):
    """spmm_kernel_ref itself built the way the reference builds it with respect to subnormals: hipcc -fgpu-flush-denormals-to-zero
    (= nvcc -ftz=true, implied by --use_fast_math, CMakeLists.txt:46)."""
    global _ref_ftz
    if _ref_ftz is None:
        L = C.CDLL(REF_LIB_FTZ)
        P = C.c_void_p
        L.ref_spmm_ref_run.argtypes = [P, P, P, P, P, C.c_int, C.c_int, P]
        L.ref_spmm_ref_run.restype = C.c_int
        _ref_ftz = L
    rc = _ref_ftz.ref_spmm_ref_run(d_ptr.data_ptr(), d_idx.data_ptr(), d_val.data_ptr(), d_vin.data_ptr(),
                                   d_vout.data_ptr(), int(num_v), int(feat), stream)
    if rc:
        raise RuntimeError(f"ref_spmm_ref_run (ftz build) -> hip error {rc}")


def ref_kernel_run(d_ptr, d_idx, d_val, d_vin, d_vout, num_v, feat, stream=0):
    """spmm_kernel_ref itself (reference source, hipcc) on torch device tensors."""
    rc = ref_lib().ref_spmm_ref_run(d_ptr.data_ptr(), d_idx.data_ptr(), d_val.data_ptr(), d_vin.data_ptr(),
                                    d_vout.data_ptr(), int(num_v), int(feat), stream)
    if rc:
        raise RuntimeError(f"ref_spmm_ref_run -> hip error {rc}")


class RefOpt:
    """The student's SpmmOptKernel itself (reference source, hipcc): preprocess + run."""

    def __init__(self, d_ptr, d_idx, d_val, num_v, feat):
        self.d_idx, self.d_val, self.feat, self.num_v = d_idx, d_val, int(feat), int(num_v)
        self.tasks = C.c_void_p(None)
        n = C.c_int(0)
        rc = ref_lib().ref_spmm_opt_preprocess(d_ptr.data_ptr(), self.num_v, C.byref(self.tasks), C.byref(n))
        if rc:
            raise RuntimeError(f"ref_spmm_opt_preprocess -> {rc}")
        self.n_tasks = n.value

    def run(self, d_vin, d_vout, stream=0):
        rc = ref_lib().ref_spmm_opt_run(self.tasks, self.n_tasks, self.d_idx.data_ptr(), self.d_val.data_ptr(),
                                        d_vin.data_ptr(), d_vout.data_ptr(), self.feat, stream)
        if rc:
            raise RuntimeError(f"ref_spmm_opt_run -> {rc}")

    def __del__(self):
        if getattr(self, "tasks", None) is not None and self.tasks.value:
            ref_lib().ref_free(self.tasks)
            self.tasks = C.c_void_p(None)


def ref_valid(d_y, d_y2, num):
    import torch

    bad = C.c_int(-1)
    if d_y.dtype == torch.float32:
        rc = ref_lib().ref_valid_float(d_y.data_ptr(), d_y2.data_ptr(), int(num), C.byref(bad))
    else:
        rc = ref_lib().ref_valid_int(d_y.data_ptr(), d_y2.data_ptr(), int(num), C.byref(bad))
    if rc:
        raise RuntimeError(f"ref_valid -> {rc}")
    return bad.value
