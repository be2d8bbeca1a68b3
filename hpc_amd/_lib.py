"""ctypes binding of include/mi_spmm.h (the C-ABI drop-in boundary).

The product path has no CPU fallback: if hpc_amd/libmi_spmm.so is missing or
lacks a symbol this module raises, and every operator in hpc_amd.spmm fails
with it.  Build with `python -c "import __graft_entry__ as g; g.build()"` or
`make -C hpc_amd/csrc`.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# Exactly one library is ever loaded: the in-tree product build (no environment override).
LIB_PATH = os.path.join(_HERE, "libmi_spmm.so")

# name -> (restype, argtypes); kept in the order of include/mi_spmm.h.
# tests/test_abi_symbols.py parses the header and checks this table covers it.
_P = C.c_void_p
SIGNATURES = {
    "mi_spmm_create": (C.c_int, [C.POINTER(_P), _P, _P, _P, C.c_int32, C.c_int32, C.c_int64, C.c_int32]),
    "mi_spmm_set_feat": (C.c_int, [_P, C.c_int32]),
    "mi_spmm_preprocess": (C.c_int, [_P, _P, _P]),
    "mi_spmm_run": (C.c_int, [_P, _P, _P, _P]),
    "mi_spmm_run_ld": (C.c_int, [_P, _P, C.c_int64, _P, C.c_int64, _P]),
    "mi_spmm_run_rows": (C.c_int, [_P, _P, C.c_int64, _P, C.c_int64, C.c_int32, C.c_int32, _P]),
    "mi_spmm_run_rows_multi": (C.c_int, [_P, _P, C.c_int64, _P, C.c_int64, C.c_int32, C.c_int32, C.c_int32, _P, _P]),
    "mi_spmm_destroy": (C.c_int, [_P]),
    "mi_spmm_strerror": (C.c_char_p, [C.c_int]),
    "mi_spmm_set_option": (C.c_int, [_P, C.c_char_p, C.c_int64]),
    "mi_spmm_get_option": (C.c_int, [_P, C.c_char_p, C.POINTER(C.c_int64)]),
    "mi_spmm_valid_float": (C.c_int, [_P, _P, C.c_int64, C.POINTER(C.c_int64), _P]),
    "mi_spmm_valid_int": (C.c_int, [_P, _P, C.c_int64, C.POINTER(C.c_int64), _P]),
    "mi_spmm_count_bitdiff": (C.c_int, [_P, _P, C.c_int64, C.POINTER(C.c_int64), C.POINTER(C.c_float), _P]),
    "mi_spmm_fill_normal": (C.c_int, [_P, C.c_int64, C.c_uint64, C.c_uint64, C.c_float, C.c_float, _P]),
    "mi_spmm_fill_philox_u32": (C.c_int, [_P, C.c_int64, C.c_uint64, C.c_uint64, _P]),
    "mi_spmm_unpack_gathered": (C.c_int, [_P, _P, C.c_int64, C.c_int32, C.c_int32, C.c_int64, _P]),
    "mi_spmm_stream_create_concurrent": (C.c_int, [C.POINTER(_P), C.c_int, C.POINTER(C.c_int)]),
    "mi_spmm_abi_version": (C.c_int, []),
    "mi_spmm_build_info": (C.c_char_p, []),
}

_lib = None

# The kernel sources a measured profile describes: profiles/traffic_latest.json stores this hash beside the counters
# (scripts/summarize_prof.py) and bench.py compares it with the tree it runs from ("traffic_stale").
# (round 5: plan.hpp / plan_types.hpp / preprocess_gpu.hip decide strip counts, thresholds and launch sets -- re-tuning a rule changes the measured traffic too)
KERNEL_SOURCES = ("spmm_kernels.hpp", "mi_spmm.hip", "hub_chain_asm.inc", "plan.hpp", "plan_types.hpp", "preprocess_gpu.hip")


def kernel_sources_sha256():
    """sha256 over the kernel sources (names and contents, in KERNEL_SOURCES order); None when the tree has no sources."""
    import hashlib

    h = hashlib.sha256()
    for name in KERNEL_SOURCES:
        path = os.path.join(_HERE, "csrc", name)
        if not os.path.exists(path):
            return None
        h.update(name.encode() + b"\0")
        with open(path, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


class MiSpmmLibraryMissing(RuntimeError):
    pass


def load():
    """Load libmi_spmm.so once and bind every declared symbol; raise if anything is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MiSpmmLibraryMissing(
            f"{LIB_PATH} not found: the HIP extension is not built. There is no CPU fallback; "
            "run `make -C hpc_amd/csrc` (or __graft_entry__.build()) first."
        )
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise MiSpmmLibraryMissing(f"{LIB_PATH} does not export {name}") from e
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib
