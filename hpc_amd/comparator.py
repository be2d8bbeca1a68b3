"""Vendor comparator: `SpMMRocSparse`, the MI355X counterpart of the reference's `SpMMCuSparse`
(PA4/workspace/include/spmm_cusparse.h:6-24, src/spmm_cusparse.cu:3-34) over
include/mi_spmm_comparator.h.  Same constructor and preprocess/run contract as every `SpMM`.
It reproduces the reference's headline metric (speed-up over the vendor library,
PA4/report.md:41-73) on this hardware and is an independent GPU-side value check.
Not on the product path."""
import ctypes as C
import os

from .spmm import SpMM, _ptr, _stream

_LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libmi_spmm_rocsparse.so")
_lib = None

ALG_DEFAULT, ALG_CSR, ALG_CSR_ROW_SPLIT, ALG_CSR_MERGE = 0, 1, 4, 5


def _load():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            raise RuntimeError(f"{_LIB_PATH} not built (make -C hpc_amd/csrc)")
        L = C.CDLL(_LIB_PATH)
        P = C.c_void_p
        L.mi_rocsparse_spmm_create.argtypes = [C.POINTER(P), P, P, P, C.c_int32, C.c_int32, C.c_int64, C.c_int32, C.c_int32]
        L.mi_rocsparse_spmm_preprocess.argtypes = [P, P, P, P]
        L.mi_rocsparse_spmm_run.argtypes = [P, P, P, P]
        L.mi_rocsparse_spmm_destroy.argtypes = [P]
        L.mi_rocsparse_spmm_buffer_bytes.argtypes = [P]
        L.mi_rocsparse_spmm_buffer_bytes.restype = C.c_int64
        _lib = L
    return _lib


class SpMMRocSparse(SpMM):
    def __init__(self, g_or_ptr, *rest, alg=ALG_DEFAULT, num_cols=None):
        super().__init__(g_or_ptr, *rest)
        self._lib = _load()
        self._h = C.c_void_p(None)
        ncols = self.num_v if num_cols is None else int(num_cols)
        rc = self._lib.mi_rocsparse_spmm_create(C.byref(self._h), _ptr(self.d_ptr), _ptr(self.d_idx), _ptr(self.d_val),
                                                self.num_v, ncols, self.num_e, self.feat_in, int(alg))
        if rc:
            raise RuntimeError(f"mi_rocsparse_spmm_create -> {rc}")

    def preprocess(self, vin, vout):
        rc = self._lib.mi_rocsparse_spmm_preprocess(self._h, _ptr(vin), _ptr(vout), _stream())
        if rc:
            raise RuntimeError(f"mi_rocsparse_spmm_preprocess -> {rc} (rocsparse_status {rc - 1000})")

    def run(self, vin, vout):
        rc = self._lib.mi_rocsparse_spmm_run(self._h, _ptr(vin), _ptr(vout), _stream())
        if rc:
            raise RuntimeError(f"mi_rocsparse_spmm_run -> {rc} (rocsparse_status {rc - 1000})")

    def buffer_bytes(self):
        return int(self._lib.mi_rocsparse_spmm_buffer_bytes(self._h))

    def __del__(self):
        h = getattr(self, "_h", None)
        if h is not None and h.value:
            self._lib.mi_rocsparse_spmm_destroy(h)
            self._h = C.c_void_p(None)
