"""Structured graph generators (hpc_amd/synth.py, round 5): the inputs scripts/regret.py puts in front of the auto rules.  CPU (torch on the host)."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")

from hpc_amd import synth  # noqa: E402


def _rows(ptr):
    return np.repeat(np.arange(ptr.size - 1), np.diff(ptr))


def _check_csr(ptr, idx, M, sorted_cols=True):
    assert ptr.dtype == np.int32 and idx.dtype == np.int32
    assert ptr[0] == 0 and ptr[-1] == idx.size and (np.diff(ptr) >= 0).all()
    assert idx.min() >= 0 and idx.max() < M
    d = np.diff(idx.astype(np.int64))
    inner = np.ones(idx.size - 1, dtype=bool)
    inner[ptr[1:-1][(ptr[1:-1] > 0) & (ptr[1:-1] < idx.size)] - 1] = False     # pairs that straddle a row boundary
    if sorted_cols:
        assert (d[inner] > 0).all()          # ascending and distinct inside a row


def test_dcsbm_is_symmetric_sorted_and_local():
    M = 1 << 14
    p, i = synth.csr_dcsbm_device(M, 24 * M, 2000, "cpu", mean_comm=512, seed=5)
    ptr, idx = p.numpy(), i.numpy()
    _check_csr(ptr, idx, M)
    rows = _rows(ptr)
    fwd = set(zip(rows.tolist(), idx.tolist()))
    assert all((c, r) in fwd for r, c in list(fwd)[:20000])          # A = A^T: a hub row is a hub column
    deg = np.diff(ptr)
    col_deg = np.bincount(idx, minlength=M)
    assert (deg == col_deg).all()
    assert deg.max() > 20 * deg.mean()                                # a power-law profile with hubs
    near = (np.abs(idx - rows) < 1024).mean()
    ps, is_ = synth.csr_dcsbm_device(M, 24 * M, 2000, "cpu", mean_comm=512, seed=5, order="shuffled")
    near_shuffled = (np.abs(is_.numpy() - _rows(ps.numpy())) < 1024).mean()
    assert near > 0.5 and near_shuffled < 0.2                         # community order = locality; the same graph shuffled has none
    assert abs(is_.numel() - idx.size) == 0                           # the same edges under another labelling


def test_dcsbm_degree_order_puts_hubs_first():
    M = 1 << 13
    p, _ = synth.csr_dcsbm_device(M, 16 * M, 1000, "cpu", mean_comm=256, seed=9, order="degree")
    deg = np.diff(p.numpy())
    assert deg[:8].mean() > 10 * deg.mean() and deg[0] == deg.max()


def test_unsorted_variant_keeps_every_row_as_a_set():
    M = 1 << 12
    p, i = synth.csr_dcsbm_device(M, 40 * M, 800, "cpu", mean_comm=256, seed=3)
    pu, iu = synth.csr_dcsbm_device(M, 40 * M, 800, "cpu", mean_comm=256, seed=3, sort_cols=False)
    assert torch.equal(p, pu)
    ptr, a, b = p.numpy(), i.numpy(), iu.numpy()
    _check_csr(ptr, b, M, sorted_cols=False)
    assert not np.array_equal(a, b)
    key = _rows(ptr).astype(np.int64) * M
    assert np.array_equal(np.sort(key + a), np.sort(key + b))
    unsorted_rows = sum(1 for r in range(M) if ptr[r + 1] - ptr[r] > 1 and (np.diff(b[ptr[r]:ptr[r + 1]]) < 0).any())
    assert unsorted_rows > M // 2


def test_plain_sbm_has_even_degrees_and_dense_diagonal_blocks():
    M = 1 << 13
    p, i = synth.csr_dcsbm_device(M, 32 * M, 64, "cpu", alpha=0, mean_comm=256, p_in=0.9, seed=2)
    ptr, idx = p.numpy(), i.numpy()
    _check_csr(ptr, idx, M)
    deg = np.diff(ptr)
    assert deg.max() < 3 * deg.mean()
    assert (np.abs(idx - _rows(ptr)) < 2048).mean() > 0.8


def test_rmat_device_is_unpermuted():
    p, i = synth.csr_rmat_device(12, "cpu", edge_factor=16)
    ptr, idx = p.numpy(), i.numpy()
    _check_csr(ptr, idx, 1 << 12)
    deg = np.diff(ptr)
    assert deg[0] == deg.max()                                        # vertex 0 is the largest hub ...
    assert np.bincount(idx, minlength=1 << 12)[0] == np.bincount(idx).max()   # ... as a column too


def test_rcm_reordering_is_a_relabelling_that_gathers_the_columns():
    M = 1 << 12
    # (a plain block model: on a power-law graph with global hubs RCM has little to gather -- small world)
    p, i = synth.csr_dcsbm_device(M, 12 * M, 300, "cpu", alpha=0, mean_comm=128, p_in=0.97, seed=4, order="shuffled")
    ptr, idx = p.numpy(), i.numpy()
    rp, ri = synth.csr_reorder_rcm(ptr, idx)
    _check_csr(rp, ri, M)
    assert ri.size == idx.size and sorted(np.diff(rp).tolist()) == sorted(np.diff(ptr).tolist())
    bw = lambda pp, ii: np.abs(ii - _rows(pp)).mean()  # noqa: E731
    assert bw(rp, ri) < 0.8 * bw(ptr, idx)          # (3 % global edges make a small world: RCM gathers what it can)


def test_dataset_structured_shapes():
    p, i = synth.csr_dataset_structured_device("ddi", "cpu")
    M, nnz, mx = synth.DATASET_SHAPES["ddi"]
    assert p.numel() == M + 1 and 0.7 * nnz < i.numel() <= nnz
    assert int(torch.diff(p).max()) > 0.5 * mx


def test_banded_long_rows_stay_inside_their_band():
    p, i = synth.csr_banded_long_rows_device(4096, "cpu", width=256, lo=30, hi=70)
    ptr, idx = p.numpy(), i.numpy()
    _check_csr(ptr, idx, 4096)
    assert (np.abs(idx - _rows(ptr)) <= 256).all()
    deg = np.diff(ptr)
    assert deg.min() >= 8 and deg.max() <= 70           # (duplicates and the clamp at the matrix edges dropped: under the drawn 30 .. 70)
