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
