"""CPU tests of the oracle (test infrastructure): internal consistency, the reference's own
comparison rules (valid.cu), and the golden fixtures made by the reference kernel on MI355X."""
import glob
import os

import numpy as np
import pytest

from hpc_amd import synth
from conftest import bits

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def test_ref_and_omp_forms_are_bitwise_equal(oracle):
    ptr, idx, vals, B, _ = synth.config("C0")
    a = oracle.spmm_ref(ptr, idx, vals, B)
    b = oracle.spmm_omp(ptr, idx, vals, B)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))


def test_hand_checkable_case(oracle):
    # 3x3: row0 = {(0,2.0),(2,-1.0)}, row1 empty, row2 = {(1,0.5)}
    ptr = np.array([0, 2, 2, 3], np.int32)
    idx = np.array([0, 2, 1], np.int32)
    val = np.array([2.0, -1.0, 0.5], np.float32)
    B = np.array([[1, 2], [3, 4], [5, 6]], np.float32)
    C = oracle.spmm_ref(ptr, idx, val, B)
    assert np.array_equal(C, np.array([[2 * 1 - 5, 2 * 2 - 6], [0, 0], [1.5, 2.0]], np.float32))


def test_empty_rows_write_zero_and_duplicates_accumulate(oracle):
    ptr = np.array([0, 0, 3, 3], np.int32)
    idx = np.array([1, 1, 0], np.int32)           # duplicate column in a row: both terms count
    val = np.array([1.0, 2.0, 4.0], np.float32)
    B = np.arange(6, dtype=np.float32).reshape(3, 2)
    C = oracle.spmm_ref(ptr, idx, val, B)
    assert np.array_equal(C[0], [0, 0]) and np.array_equal(C[2], [0, 0])
    assert np.array_equal(C[1], 3 * B[1] + 4 * B[0])


def test_fma_matters(oracle):
    """the fused and unfused builds of the same loop differ in about half the
    elements -- the oracle pins contraction to fma (nvcc --use_fast_math => fmad)."""
    ptr, idx, vals, B, _ = synth.config("C0")
    a = oracle.spmm_ref(ptr, idx, vals, B)
    c = oracle.spmm_nofma(ptr, idx, vals, B)
    assert (a != c).sum() > a.size // 4
    assert np.abs(a - c).max() < 1e-6


def test_against_fp64_and_scipy(oracle):
    import scipy.sparse as sp

    ptr, idx, vals, B, meta = synth.config("C0")
    f64, sabs = oracle.spmm_f64(ptr, idx, vals, B)
    A = sp.csr_matrix((vals.astype(np.float64), idx, ptr), shape=(meta["M"], meta["K"]))
    assert np.abs(A @ B.astype(np.float64) - f64).max() < 1e-12
    a = oracle.spmm_ref(ptr, idx, vals, B)
    err = np.abs(a - f64)
    assert (err <= 1e-6 * sabs + 1e-30).all()


def test_valid_float_rules(oracle):
    # valid.cu:6: |(y - y2)/y| > 1e-2, quotient against the FIRST argument; 0/0 not counted, x/0 counted
    y = np.array([1.0, 1.0, 0.0, 0.0, 100.0, -2.0], np.float32)
    y2 = np.array([1.0, 1.02, 0.0, 1e-9, 100.9, -2.03], np.float32)
    #             ok    bad   nan->ok  inf->bad  ok(0.9%)  bad(1.5%)
    assert oracle.valid_float(y, y2) == 3
    assert oracle.valid_int(np.array([1, 2, 3]), np.array([1, 5, 3])) == 1
    # test_spmm.cu:43  bad < M*N/10000 + 1
    assert oracle.validation_passes(0, 10, 10)
    assert not oracle.validation_passes(1, 10, 10)
    assert oracle.validation_passes(3, 1024, 32)
    assert not oracle.validation_passes(4, 1024, 32)


def test_chunked_order_helper_matches_oracle_below_threshold(oracle):
    ptr, idx, vals, B, _ = synth.config("C0")
    a = oracle.spmm_omp(ptr, idx, vals, B)
    b = oracle.spmm_chunked(ptr, idx, vals, B, threshold=1 << 30, chunk=256)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    c = oracle.spmm_chunked(ptr, idx, vals, B, threshold=8, chunk=8)
    f64, sabs = oracle.spmm_f64(ptr, idx, vals, B)
    assert (np.abs(c - f64) <= 1e-6 * sabs + 1e-30).all()


def _golden_files():
    return sorted(glob.glob(os.path.join(GOLDEN, "*.npz")))


def test_golden_fixtures_exist():
    assert _golden_files(), "tests/golden/*.npz missing: run tests/golden/make_golden.py on the GPU box"


@pytest.mark.parametrize("path", _golden_files(), ids=lambda p: os.path.basename(p))
def test_oracle_reproduces_reference_kernel_goldens(oracle, path):
    """The fixtures hold outputs of the REFERENCE kernel (spmm_ref.cu:3-17 compiled by hipcc,
    run on an MI355X).  The CPU restatement must reproduce them bit for bit."""
    z = np.load(path)
    out = oracle.spmm_ref(z["row_ptr"], z["col_idx"], z["vals"], z["B"])
    exp = z["C_ref_kernel"]
    assert out.shape == exp.shape
    assert np.array_equal(out.view(np.uint32), exp.view(np.uint32)), (
        f"{(out.view(np.uint32) != exp.view(np.uint32)).sum()} of {out.size} elements differ"
    )
    out2 = oracle.spmm_omp(z["row_ptr"], z["col_idx"], z["vals"], z["B"])
    assert np.array_equal(out2.view(np.uint32), exp.view(np.uint32))


@pytest.mark.parametrize("path", _golden_files(), ids=lambda p: os.path.basename(p))
def test_reference_kernel_body_built_for_the_host_reproduces_the_goldens(path):
    """Provenance of the fixtures without a GPU: the reference's own 15 lines (spmm_ref.cu:3-17), compiled for the host by
    oracle/Makefile `_ref_host` (g++ -mfma -ffp-contract=fast; the launch loop of SpMMRef::run around them), give the
    fixtures' C_ref_kernel bit for bit.  The build reads the reference tree, so it happens only where that tree is (this
    container); a prebuilt oracle/_ref/libspmm_ref_host.so is used as it stands; with neither the test is skipped."""
    import ctypes
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    so = os.path.join(root, "oracle", "_ref", "libspmm_ref_host.so")
    if not os.path.exists(so):
        subprocess.check_call(["make", "-s", "-C", os.path.join(root, "oracle"), "_ref_host"])
    if not os.path.exists(so):
        pytest.skip("no reference tree and no prebuilt oracle/_ref/libspmm_ref_host.so here")
    lib = ctypes.CDLL(so)
    z = np.load(path)
    ptr, idx = np.ascontiguousarray(z["row_ptr"], np.int32), np.ascontiguousarray(z["col_idx"], np.int32)
    vals, B = np.ascontiguousarray(z["vals"], np.float32), np.ascontiguousarray(z["B"], np.float32)
    M, N = ptr.size - 1, B.shape[1]
    out = np.full((M, N), np.nan, np.float32)
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    lib.ref_host_spmm.argtypes = [ctypes.c_void_p] * 5 + [ctypes.c_int, ctypes.c_int]
    assert lib.ref_host_spmm(p(ptr), p(idx), p(vals), p(B), p(out), M, N) == 0
    exp = z["C_ref_kernel"]
    assert np.array_equal(out.view(np.uint32), exp.view(np.uint32)), f"{(out.view(np.uint32) != exp.view(np.uint32)).sum()} of {out.size} differ"


def test_philox4x32_10_known_answers(oracle):
    """Random123 kat_vectors for philox4x32-10: pins the generator behind mi_spmm_fill_normal."""
    assert oracle.philox_block([0, 0, 0, 0], [0, 0]) == [0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8]
    assert oracle.philox_block([0xFFFFFFFF] * 4, [0xFFFFFFFF] * 2) == [0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD]
    assert oracle.philox_block([0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344], [0xA4093822, 0x299F31D0]) == [
        0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1]


def test_fill_normal_restatement_statistics(oracle):
    z = oracle.fill_normal(1 << 20, 123, 0, 0.0, 0.1).astype(np.float64)
    assert abs(z.mean()) < 5e-4 and abs(z.std() - 0.1) < 5e-4        # N(0, 0.1): data.h:31
    assert abs((z ** 3).mean()) < 2e-5 and abs((z ** 4).mean() / 0.1 ** 4 - 3.0) < 0.05
    assert np.isfinite(z).all() and np.abs(z).max() < 0.62            # |z| <= 0.1*sqrt(2*24*ln2)
    a = oracle.fill_normal(1001, 5, 7)
    b = oracle.fill_normal(4000, 5, 7)
    assert np.array_equal(a, b[:1001])                                # element i depends on (seed, subseq, i) only
    assert not np.array_equal(oracle.fill_normal(64, 5, 8), b[:64])


def test_flush_to_zero_restatement_of_the_reference_build(oracle):
    """The reference is BUILT with nvcc --use_fast_math (W/CMakeLists.txt:46) => -ftz=true => fma.rn.ftz.f32: subnormal inputs and results
    are flushed to sign-preserving zeros.  oracle.spmm_ftz restates that; on data without subnormals it is the canonical oracle bit for bit."""
    f32 = np.float32
    sub = np.frombuffer(np.uint32(0x00012345).tobytes(), f32)[0]          # a subnormal (~1.07e-40)
    assert 0 < sub < np.finfo(f32).tiny
    ptr = np.array([0, 1, 2, 4, 6, 7], np.int32)
    idx = np.array([0, 1, 2, 2, 0, 1, 3], np.int32)
    val = np.array([sub, f32(1e-20), f32(1.0), f32(1.0), f32(-1e-20), f32(3.0), -sub], f32)
    B = np.array([[1e10], [1e-20], [0.5], [4.0]], f32)
    plain = oracle.spmm_omp(ptr, idx, val, B)
    ftz = oracle.spmm_ftz(ptr, idx, val, B)
    # row 0: subnormal a times 1e10 -- a normal number in IEEE arithmetic, nothing once the input is flushed
    assert plain[0, 0] > 0 and ftz[0, 0] == 0.0 and not np.signbit(ftz[0, 0])
    # row 1: 1e-20 * 1e-20 = 1e-40: a subnormal RESULT, flushed
    assert 0 < plain[1, 0] < np.finfo(f32).tiny and ftz[1, 0] == 0.0
    # row 2: normal terms are untouched
    assert ftz[2, 0] == plain[2, 0] == f32(1.0)
    # row 3: -1e-20 * 1e10 + 3 * 1e-20: normal, untouched;  row 4: negative subnormal input: -0 * 4 = -0, added to +0 = +0
    assert ftz[3, 0] == plain[3, 0] and ftz[4, 0] == 0.0 and not np.signbit(ftz[4, 0])
    # a subnormal result keeps its sign as a zero: (-1e-20) * (1e-20) from a -0 accumulator ... start is +0, so the sum is +0; check the product alone
    pn = oracle.spmm_ftz(np.array([0, 1], np.int32), np.array([0], np.int32), np.array([-1e-20], f32), np.array([[1e-20], ], f32))
    assert pn[0, 0] == 0.0       # fma(-1e-20, 1e-20, +0) = -1e-40 -> flushed to -0; -0 == 0, and the stored bits are the sign-preserving zero:
    assert bits(pn)[0, 0] == 0x80000000
    # the reference's own kind of data (N(0, 0.1), BASELINE configs[0]) holds no subnormals and makes none: both forms agree on every bit
    p0, i0, v0, B0, _ = synth.config("C0")
    assert np.array_equal(bits(oracle.spmm_ftz(p0, i0, v0, B0)), bits(oracle.spmm_omp(p0, i0, v0, B0)))
