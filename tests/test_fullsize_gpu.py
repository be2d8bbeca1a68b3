"""BASELINE.json full sizes: size-independent properties + oracle on a row sample."""
import numpy as np
import pytest

from conftest import bits, to_dev
from hpc_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def c1(device):
    import torch
    from hpc_amd import CSR, SpMMOpt

    ptr, idx, vals, B, meta = synth.config("C1")
    d_ptr, d_idx, d_val, d_B = to_dev(device, ptr, idx, vals, B)
    d_C = torch.full((meta["M"], meta["N"]), float("nan"), dtype=torch.float32, device=device)
    op = SpMMOpt(CSR(meta["M"], meta["nnz"], d_ptr, d_idx, d_val), meta["N"])
    op.preprocess(d_B, d_C)
    op.run(d_B, d_C)
    torch.cuda.synchronize()
    return dict(ptr=ptr, idx=idx, vals=vals, B=B, meta=meta, d=(d_ptr, d_idx, d_val, d_B, d_C), op=op)


def _sample_rows(ptr, idx, vals, rows):
    deg = np.diff(ptr)[rows]
    sp = np.concatenate([[0], np.cumsum(deg)]).astype(np.int32)
    take = np.concatenate([np.arange(ptr[r], ptr[r + 1]) for r in rows]) if len(rows) else np.zeros(0, np.int64)
    return sp, idx[take], vals[take]


def test_c1_rows_sample_bitwise(c1, oracle):
    """M = 2^20, nnz ~ 2^25, N = 128: 8192 sampled rows (first, last, random) against the oracle."""
    M = c1["meta"]["M"]
    g = np.random.Generator(np.random.Philox(key=[99, 0]))
    rows = np.unique(np.concatenate([[0, 1, M - 2, M - 1], g.integers(0, M, 8192)]))
    sp, si, sv = _sample_rows(c1["ptr"], c1["idx"], c1["vals"], rows)
    exp = oracle.spmm_omp(sp, si, sv, c1["B"])
    got = c1["d"][4][rows.tolist()].cpu().numpy()
    assert not np.isnan(got).any()
    assert np.array_equal(bits(got), bits(exp))


def test_c1_no_nan_left_and_power_of_two_scaling(c1):
    """Every element overwritten; scaling A by 2 scales C by exactly 2 (exact in fp32)."""
    import torch

    d_ptr, d_idx, d_val, d_B, d_C = c1["d"]
    assert not torch.isnan(d_C).any()
    from hpc_amd import CSR, SpMMOpt

    d_val2 = d_val * 2
    d_C2 = torch.empty_like(d_C)
    op = SpMMOpt(CSR(c1["meta"]["M"], c1["meta"]["nnz"], d_ptr, d_idx, d_val2), c1["meta"]["N"])
    op.preprocess(d_B, d_C2)
    op.run(d_B, d_C2)
    torch.cuda.synchronize()
    assert torch.equal(d_C2, d_C * 2)


def test_c1_reference_validator_vs_reference_kernel(c1, oracle):
    """The reference's acceptance test at full size: SpMMRef's kernel (hipcc) vs ours, valid() < M*N/10000+1;
    in fact bitwise equal."""
    import torch
    from hpc_amd import valid
    from hpc_amd.spmm import count_bitdiff

    if not oracle.ref_available():
        pytest.fail("oracle/_ref missing")
    d_ptr, d_idx, d_val, d_B, d_C = c1["d"]
    M, N = c1["meta"]["M"], c1["meta"]["N"]
    d_R = torch.full((M, N), float("nan"), dtype=torch.float32, device=d_C.device)
    oracle.ref_kernel_run(d_ptr, d_idx, d_val, d_B, d_R, M, N)
    torch.cuda.synchronize()
    bad = valid(d_C, d_R, M * N)
    assert oracle.validation_passes(bad, M, N) and bad == 0
    ndiff, maxabs = count_bitdiff(d_C, d_R)
    assert ndiff == 0 and maxabs == 0.0


def test_permutation_matrix_is_a_row_gather(device):
    """A = permutation (one 1.0 per row): C must be B[perm] exactly, at M = 2^20, N = 128."""
    import torch
    from hpc_amd import CSR, SpMMOpt

    M, N = 1 << 20, 128
    g = torch.Generator(device="cpu").manual_seed(5)
    perm = torch.randperm(M, generator=g, dtype=torch.int64).to(torch.int32).to(device)
    d_ptr = torch.arange(M + 1, dtype=torch.int32, device=device)
    d_val = torch.ones(M, dtype=torch.float32, device=device)
    d_B = torch.randn(M, N, device=device)
    d_C = torch.full((M, N), float("nan"), device=device)
    op = SpMMOpt(CSR(M, M, d_ptr, perm, d_val), N)
    op.preprocess(d_B, d_C)
    op.run(d_B, d_C)
    torch.cuda.synchronize()
    assert torch.equal(d_C, d_B[perm.long()])


# ---- the other BASELINE configurations at full size (round-1 gap: only C1 was checked at M = 2^20) ----
def _run_full(device, ptr, idx, vals, d_B, N, K, options=None):
    import torch
    from hpc_amd import CSR, SpMMOpt

    M = ptr.size - 1
    d_ptr, d_idx, d_val = to_dev(device, ptr, idx, vals)
    d_C = torch.full((M, N), float("nan"), dtype=torch.float32, device=device)
    op = SpMMOpt(CSR(M, idx.size, d_ptr, d_idx, d_val), N, num_cols=K)
    for k, v in (options or {}).items():
        op.set_option(k, v)
    op.preprocess(d_B, d_C)
    op.run(d_B, d_C)
    torch.cuda.synchronize()
    return d_ptr, d_idx, d_val, d_C, op


def test_c2_power_law_full_size(device, oracle):
    """BASELINE configs[2] at M = 2^20 (max 4096 nnz/row, N = 128): every split row and 4096 random rows against
    the oracle evaluated in the documented piece order (bit-exact); unsplit rows bit-exact against the plain
    oracle; the exact-order setting bit-identical to the reference's own kernel on the whole C."""
    import torch
    from hpc_amd.spmm import count_bitdiff

    ptr, idx, vals, B, meta = synth.config("C2")
    M, N = meta["M"], meta["N"]
    assert M == 1 << 20 and 4000 <= meta["deg_max"] <= 4096      # duplicate (row, col) draws are dropped
    (d_B,) = to_dev(device, B)
    d_ptr, d_idx, d_val, d_C, op = _run_full(device, ptr, idx, vals, d_B, N, M)
    thr = op.get_option("long_row_threshold")
    deg = np.diff(ptr)
    split = np.nonzero(deg > thr)[0]
    assert thr == 2048 and op.get_option("n_long_rows") == split.size > 0 and op.get_option("n_medium_rows") > 0
    assert not torch.isnan(d_C).any()
    g = np.random.Generator(np.random.Philox(key=[99, 2]))
    rows = np.unique(np.concatenate([split, [0, M - 1], g.integers(0, M, 4096)]))
    sp, si, sv = _sample_rows(ptr, idx, vals, rows)
    got = d_C[rows.tolist()].cpu().numpy()
    exp = oracle.spmm_chunked(sp, si, sv, B, thr, 256)
    assert np.array_equal(bits(got), bits(exp))
    plain = oracle.spmm_omp(sp, si, sv, B)
    unsplit = deg[rows] <= thr
    assert np.array_equal(bits(got[unsplit]), bits(plain[unsplit]))
    _, sabs = oracle.spmm_f64(sp, si, sv, B)
    assert (np.abs(got.astype(np.float64) - plain) <= 1e-5 * sabs + 1e-30).all()      # split rows: documented tolerance
    # exact-order setting == the reference kernel, whole matrix
    if not oracle.ref_available():
        pytest.fail("oracle/_ref missing")
    d_R = torch.full((M, N), float("nan"), dtype=torch.float32, device=device)
    oracle.ref_kernel_run(d_ptr, d_idx, d_val, d_B, d_R, M, N)
    _, _, _, d_E, ope = _run_full(device, ptr, idx, vals, d_B, N, M, {"long_row_threshold": 1 << 30})
    assert ope.get_option("n_long_rows") == 0
    ndiff, maxabs = count_bitdiff(d_E, d_R)
    assert ndiff == 0 and maxabs == 0.0
    ndiff_auto, _ = count_bitdiff(d_C, d_R)                    # auto mode: only elements of split rows may differ
    assert ndiff_auto <= split.size * N


def test_c4_block_dense_full_size(device, oracle):
    """BASELINE configs[4] at M = 2^20, N = 256 (151 M nonzeros, every 16-row group on the MFMA block path):
    whole C bit-identical to the reference's own kernel; 4096 sampled rows bit-identical to the oracle."""
    import torch
    from hpc_amd.spmm import count_bitdiff

    ptr, idx, vals, B, meta = synth.config("C4")
    M, N = meta["M"], meta["N"]
    assert M == 1 << 20 and N == 256
    (d_B,) = to_dev(device, B)
    d_ptr, d_idx, d_val, d_C, op = _run_full(device, ptr, idx, vals, d_B, N, M)
    assert op.get_option("n_block_groups") == M // 16
    assert not torch.isnan(d_C).any()
    g = np.random.Generator(np.random.Philox(key=[99, 4]))
    rows = np.unique(np.concatenate([[0, 15, 16, M - 1], g.integers(0, M, 4096)]))
    sp, si, sv = _sample_rows(ptr, idx, vals, rows)
    assert np.array_equal(bits(d_C[rows.tolist()].cpu().numpy()), bits(oracle.spmm_omp(sp, si, sv, B)))
    if not oracle.ref_available():
        pytest.fail("oracle/_ref missing")
    d_R = torch.full((M, N), float("nan"), dtype=torch.float32, device=device)
    oracle.ref_kernel_run(d_ptr, d_idx, d_val, d_B, d_R, M, N)
    torch.cuda.synchronize()
    ndiff, maxabs = count_bitdiff(d_C, d_R)
    assert ndiff == 0 and maxabs == 0.0
    # the rows kernel alone (block path off) gives the same bits
    _, _, _, d_C0, op0 = _run_full(device, ptr, idx, vals, d_B, N, M, {"block_path": 0})
    assert op0.get_option("n_block_groups") == 0
    assert count_bitdiff(d_C0, d_R)[0] == 0


def test_c3_one_gpu_leg_n1024_at_the_narrow_address_boundary(device, oracle):
    """BASELINE configs[3]'s one-GPU leg: C1's CSR with N = 1024.  B is exactly 4 GiB, so the last byte of B sits
    exactly at 2^32: the largest case the 32-bit-offset ("narrow") kernels take.  B is filled on the device
    (4 GiB of host normals would take longer than the test); whole C against the reference's own kernel,
    sampled rows against the oracle on the gathered B rows."""
    import torch
    from hpc_amd.spmm import count_bitdiff, fill_normal

    ptr, idx = synth.csr_uniform(1 << 20, 16, 48)
    vals = synth.make_values(idx.size)
    M = K = 1 << 20
    N = 1024
    d_B = torch.empty(K * N, dtype=torch.float32, device=device)
    fill_normal(d_B, seed=125, subsequence=3)
    d_ptr, d_idx, d_val, d_C, op = _run_full(device, ptr, idx, vals, d_B, N, K)
    # narrow (32-bit offset) addressing at the exact boundary; random columns and N >= 256 -> 64-column tiles (16 lanes per row)
    assert op.get_option("wide_addressing") == 0 and op.get_option("lanes_per_row") == 16 and op.get_option("column_locality_pct") < 10
    assert not torch.isnan(d_C).any()
    g = np.random.Generator(np.random.Philox(key=[99, 3]))
    rows = np.unique(np.concatenate([[0, M - 1], g.integers(0, M, 1024)]))
    # rows whose columns include the very last B rows (the bytes next to 2^32)
    last_users = np.nonzero(np.isin(idx, [K - 1, K - 2]))[0]
    rows = np.unique(np.concatenate([rows, np.searchsorted(ptr, last_users, side="right") - 1]))
    sp, si, sv = _sample_rows(ptr, idx, vals, rows)
    cols, inv = np.unique(si, return_inverse=True)
    Bsub = d_B.view(K, N)[torch.from_numpy(cols.astype(np.int64)).to(device)].cpu().numpy()
    exp = oracle.spmm_omp(sp, inv.astype(np.int32), sv, Bsub)
    assert np.array_equal(bits(d_C[rows.tolist()].cpu().numpy()), bits(exp))
    if not oracle.ref_available():
        pytest.fail("oracle/_ref missing")
    d_R = torch.full((M, N), float("nan"), dtype=torch.float32, device=device)
    oracle.ref_kernel_run(d_ptr, d_idx, d_val, d_B, d_R, M, N)
    torch.cuda.synchronize()
    ndiff, maxabs = count_bitdiff(d_C, d_R)
    assert ndiff == 0 and maxabs == 0.0
