"""BASELINE.json full sizes: size-independent properties + oracle on a row sample."""
import numpy as np
import pytest

from conftest import auto_hub_threshold, bits, to_dev
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
    """BASELINE configs[2] at M = 2^20 (max 4096 nnz/row, N = 128) with DEFAULT options: the whole C bit-identical to the
    reference's own kernel (spmm_kernel_ref compiled for this GPU) -- hub rows included, they keep their stored order
    through the hub kernel.  The opt-in split mode: every split row and 4096 random rows against the oracle evaluated in
    the documented piece order (bit-exact), within 1e-5 * sum|a*b| of the plain chain, plain relative error printed."""
    import torch
    from hpc_amd.spmm import count_bitdiff

    ptr, idx, vals, B, meta = synth.config("C2")
    M, N = meta["M"], meta["N"]
    assert M == 1 << 20 and 4000 <= meta["deg_max"] <= 4096      # duplicate (row, col) draws are dropped
    (d_B,) = to_dev(device, B)
    d_ptr, d_idx, d_val, d_C, op = _run_full(device, ptr, idx, vals, d_B, N, M)
    deg = np.diff(ptr)
    # auto threshold at this size (plan.hpp resolve_hub_threshold): the step hides a segment of up to 5 625 nonzeros -- C2's longest row (4 095)
    # included, so no row needs the hub kernel and the rule returns its largest candidate (round 5; it was 4096, the largest power of two below 5 625:
    # the same plan here) -- every row of C2 is a short row or ONE exact segment; the hub kernel gets its full-size run below (threshold 2048)
    assert op.get_option("long_row_threshold") == auto_hub_threshold(M, N, ptr) == 8192
    assert op.get_option("split_long_rows") == 0 and op.get_option("n_partial_slots") == 0
    assert op.get_option("n_hub_rows") == 0 and op.get_option("n_medium_rows") > 0
    assert not torch.isnan(d_C).any()
    thr = 2048
    hubs = np.nonzero(deg > thr)[0]
    if not oracle.ref_available():
        pytest.fail("oracle/_ref missing")
    d_R = torch.full((M, N), float("nan"), dtype=torch.float32, device=device)
    oracle.ref_kernel_run(d_ptr, d_idx, d_val, d_B, d_R, M, N)
    ndiff, maxabs = count_bitdiff(d_C, d_R)
    assert ndiff == 0 and maxabs == 0.0, "default options must reproduce the reference kernel bit for bit on every row"
    # ---- the 422 rows above 2048 nonzeros through the hub kernel: still the reference kernel's bits on the whole C
    _, _, _, d_H, oph = _run_full(device, ptr, idx, vals, d_B, N, M, {"long_row_threshold": thr})
    assert oph.get_option("n_hub_rows") == hubs.size > 0 and oph.get_option("n_partial_slots") == 0
    ndiff, maxabs = count_bitdiff(d_H, d_R)
    assert ndiff == 0 and maxabs == 0.0
    # the hub rows once more against the CPU restatement (three-way agreement on the rows that changed path this round)
    sp, si, sv = _sample_rows(ptr, idx, vals, hubs)
    assert np.array_equal(bits(d_H[hubs.tolist()].cpu().numpy()), bits(oracle.spmm_omp(sp, si, sv, B)))
    del d_H, oph
    # ---- opt-in split mode
    _, _, _, d_S, ops = _run_full(device, ptr, idx, vals, d_B, N, M, {"split_long_rows": 1, "long_row_threshold": thr})
    assert ops.get_option("n_long_rows") == hubs.size and ops.get_option("n_hub_rows") == 0
    g = np.random.Generator(np.random.Philox(key=[99, 2]))
    rows = np.unique(np.concatenate([hubs, [0, M - 1], g.integers(0, M, 4096)]))
    sp, si, sv = _sample_rows(ptr, idx, vals, rows)
    got = d_S[rows.tolist()].cpu().numpy()
    assert np.array_equal(bits(got), bits(oracle.spmm_chunked(sp, si, sv, B, thr, 256)))
    plain = oracle.spmm_omp(sp, si, sv, B)
    _, sabs = oracle.spmm_f64(sp, si, sv, B)
    assert (np.abs(got.astype(np.float64) - plain) <= 1e-5 * sabs + 1e-30).all()      # split rows: documented tolerance
    split_sel = deg[rows] > thr
    rel = np.abs(got[split_sel].astype(np.float64) - plain[split_sel]) / np.maximum(np.abs(plain[split_sel].astype(np.float64)), 1e-300)
    print(f"C2 split mode, {int(split_sel.sum())} split rows: plain relative error max {rel.max():.3e}, p99.9 {np.quantile(rel, 0.999):.3e}, "
          f"share > 1e-5: {(rel > 1e-5).mean():.5f}")
    ndiff_split, _ = count_bitdiff(d_S, d_R)                   # only elements of split rows may differ -- and some of them must:
    assert 0 < ndiff_split <= hubs.size * N                    # a counter stuck at 0 would pass every `== 0` above
    # the device counter against numpy on the split rows themselves (every other row is bit-identical, asserted by the bound above)
    ref_rows = d_R[hubs.tolist()].cpu().numpy()
    assert ndiff_split == int((bits(d_S[hubs.tolist()].cpu().numpy()) != bits(ref_rows)).sum())


# name: (rows, nonzeros, longest row) -- shapes of the reference's datasets (scripts/report_table.py), N
_HUB_GRAPHS = {
    "rmat20": (None, None, None, 128),
    "ddi": (4_267, 2_135_822, 2_234, 128),
    "reddit": (232_965, 114_615_892, 21_657, 32),
    "am": (881_680, 5_668_682, 154_828, 128),
}


@pytest.mark.parametrize("name", list(_HUB_GRAPHS))
def test_hub_graphs_default_options_bit_identical_to_reference_kernel(device, oracle, name):
    """Graphs whose longest rows are far beyond the split threshold (R-MAT scale 20: a 64 K-nonzero hub; ddi-, reddit- and
    am-shaped: 2 K ... 155 K), default options (the reddit-shaped graph's segments in column strips): count_bitdiff(ours, spmm_kernel_ref) == 0
    over the whole C."""
    import torch
    from hpc_amd.spmm import count_bitdiff

    rows, nnz_t, mx, N = _HUB_GRAPHS[name]
    if name == "rmat20":
        ptr, idx = synth.csr_rmat(20, 32)
    else:
        ptr, idx = synth.csr_powerlaw(rows, nnz_t / rows, min(mx, rows), seed=sum(map(ord, name)) % 1000 + 1, force_max=True)
    M = ptr.size - 1
    vals = synth.make_values(idx.size)
    d_B = torch.empty(M * N, dtype=torch.float32, device=device)
    from hpc_amd.spmm import fill_normal
    fill_normal(d_B, 125, 0, 0.0, 0.1)
    d_B = d_B.view(M, N)
    d_ptr, d_idx, d_val, d_C, op = _run_full(device, ptr, idx, vals, d_B, N, M)
    deg = np.diff(ptr)
    thr = op.get_option("long_row_threshold")
    assert thr == auto_hub_threshold(M, N, ptr)
    assert op.get_option("n_hub_rows") == int((deg > thr).sum()) and op.get_option("n_partial_slots") == 0
    assert op.get_option("n_hub_rows") > 0 or name == "ddi"     # ddi-shaped: the auto threshold sits near its longest row
    if name == "reddit":        # long rows over few columns: its segments run in column strips -- checked here against the reference kernel itself
        assert op.get_option("n_col_strips") >= 2 and op.get_option("segments_unsorted") == 0
    if not oracle.ref_available():
        pytest.fail("oracle/_ref missing")
    d_R = torch.full((M, N), float("nan"), dtype=torch.float32, device=device)
    oracle.ref_kernel_run(d_ptr, d_idx, d_val, d_B, d_R, M, N)
    ndiff, maxabs = count_bitdiff(d_C, d_R)
    assert ndiff == 0 and maxabs == 0.0, (name, int(deg.max()), thr, ndiff)


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


def test_count_bitdiff_sees_planted_differences(device):
    """Every whole-C equality claim of this file rests on `count_bitdiff` (mi_spmm_count_bitdiff -> compare_kernel mode 2, the
    exact counterpart of the reference's validate_int, W/src/valid.cu:13-20).  Here the judge is judged: two 2^26-element
    buffers that differ in k planted places -- first and last element, a +0 / -0 pair (equal as floats, different bits), a
    pair of NaNs with different payloads, a denormal against zero, neighbours in one wave, places 2^24 apart (beyond one
    grid stride) -- must give exactly k, and the largest |a - b| numpy finds."""
    import torch
    from hpc_amd.spmm import count_bitdiff, valid

    n = 1 << 26
    g = np.random.Generator(np.random.Philox(key=[404, 1]))
    a = torch.empty(n, dtype=torch.float32, device=device).normal_(0.0, 0.1)
    b = a.clone()
    assert count_bitdiff(a, b) == (0, 0.0)
    places = np.unique(np.concatenate([[0, 1, 63, 64, 65, 255, 256, n - 1, n - 2, 1 << 24, (1 << 24) + 1, (1 << 25) + 7],
                                       g.integers(0, n, 1000)])).astype(np.int64)
    k = int(places.size)
    hb = b[places.tolist()].cpu().numpy().copy()
    ha = a[places.tolist()].cpu().numpy().copy()
    hb = hb + np.float32(0.25) * (1 + (np.arange(k) % 7)).astype(np.float32)     # finite, exactly representable offsets: every place differs
    # +0 against -0 at place index 3: a bit difference with |a - b| == 0
    ha[3], hb[3] = np.float32(0.0), np.float32(-0.0)
    # a denormal against zero at place index 5
    ha[5], hb[5] = np.float32(0.0), np.frombuffer(np.uint32(1).tobytes(), np.float32)[0]
    a[places.tolist()] = torch.from_numpy(ha).to(device)
    b[places.tolist()] = torch.from_numpy(hb).to(device)
    assert (bits(ha) != bits(hb)).all()
    ndiff, maxabs = count_bitdiff(a, b)
    assert ndiff == k, (ndiff, k)
    assert maxabs == float(np.abs(ha - hb).max()) > 0.25       # fp32 subtraction on both sides
    assert count_bitdiff(b, a) == (ndiff, maxabs)
    # the reference's float rule on the same buffers (valid.cu:6: |(ref - ans) / ref| > 1e-2) against numpy
    with np.errstate(divide="ignore", invalid="ignore"):
        q = np.abs((ha - hb) / ha)
    assert valid(a, b, n) == int((q.astype(np.float64) > 1e-2).sum())
    # a NaN pair with different payloads: counted, and the maximum is reported as inf (a NaN took part)
    nan_a = np.frombuffer(np.uint32(0x7FC00001).tobytes(), np.float32)[0]
    nan_b = np.frombuffer(np.uint32(0x7FC00002).tobytes(), np.float32)[0]
    a2, b2 = a.clone(), a.clone()
    a2[n - 1], b2[n - 1] = float("nan"), float("nan")
    a2.view(torch.int32)[n - 1], b2.view(torch.int32)[n - 1] = 0x7FC00001, 0x7FC00002
    assert np.isnan(nan_a) and np.isnan(nan_b)
    nd, mx = count_bitdiff(a2, b2)
    assert nd == 1 and mx == float("inf")
    a2.view(torch.int32)[n - 1] = 0x7FC00002            # same NaN bits on both sides: no difference
    assert count_bitdiff(a2, b2) == (0, 0.0)


def test_long_rows_device_generator_full_size_against_the_reference_kernel(device, oracle):
    """`bench.py`'s `also.LONG_ROWS` input (synth.csr_long_rows_device: 300-700 nonzeros in every row, columns ascending over all K, built on the device)
    had no parity test of its own: the column strips were checked at full size only through the reddit-shaped graph at N = 32.
    Here at the bench's size and width: whole C against spmm_kernel_ref itself, strips in force, both strip builders, and the tables' hash equal."""
    import torch
    from hpc_amd import CSR, SpMMOpt
    from hpc_amd.spmm import count_bitdiff, fill_normal

    M, N = 1 << 17, 128
    d_ptr, d_idx = synth.csr_long_rows_device(M, device)
    nnz = int(d_idx.numel())
    deg = torch.diff(d_ptr)
    assert int(deg.min()) >= 300 and int(deg.max()) <= 700
    d_val = torch.empty(nnz, dtype=torch.float32, device=device)
    fill_normal(d_val, 124, 0, 0.0, 0.1)
    d_B = torch.empty(M * N, dtype=torch.float32, device=device)
    fill_normal(d_B, 125, 0, 0.0, 0.1)
    d_B = d_B.view(M, N)
    if not oracle.ref_available():
        pytest.fail("oracle/_ref missing")
    d_R = torch.full((M, N), float("nan"), dtype=torch.float32, device=device)
    oracle.ref_kernel_run(d_ptr, d_idx, d_val, d_B, d_R, M, N)
    hashes = []
    for builder in (0, 1):
        d_C = torch.full((M, N), float("nan"), dtype=torch.float32, device=device)
        op = SpMMOpt(CSR(M, nnz, d_ptr, d_idx, d_val), N)
        op.set_option("col_strips_builder", builder)
        op.preprocess(d_B, d_C)
        assert op.get_option("n_col_strips") >= 8 and op.get_option("segments_unsorted") == 0 and op.get_option("n_chunks") == M
        assert op.get_option("segment_nnz") == nnz
        op.run(d_B, d_C)
        op.run(d_B, d_C)
        ndiff, maxabs = count_bitdiff(d_C, d_R)
        assert ndiff == 0 and maxabs == 0.0, (builder, ndiff)
        hashes.append(op.get_option("col_strips_table_hash"))
        print(f"LONG_ROWS builder {builder}: preprocess {op.get_option('preprocess_us')} us, strips {op.get_option('n_col_strips')}")
    assert hashes[0] == hashes[1] != 0


# Round 5: the graphs the auto rules were worst on (profiles/r05_regret_before_rule_fixes.md / r05_regret.md), at the regret suite's sizes, default options,
# whole C against spmm_kernel_ref -- with the plan the fixed rules now choose pinned beside the bits, so that a rule that drifts back shows up here.
_STRUCTURED = {
    # name: (builder, N, what the plan must look like)
    "reddit-community": (lambda dev: synth.csr_dataset_structured_device("reddit.dgl", dev), 256,
                         lambda o: o("n_col_strips") >= 8 and o("segments_unsorted") == 0 and o("column_locality_pct") >= 50),          # 60 % local: round 4's gate kept strips off (+77 %)
    "banded-long-rows": (lambda dev: synth.csr_banded_long_rows_device(1 << 17, dev), 128,
                         lambda o: o("column_locality_pct") >= 95 and o("medium_row_threshold") >= 512),                              # rows stay with their neighbours (+89 % before)
    "protein-unsorted": (lambda dev: synth.csr_dataset_structured_device("protein", dev, sort_cols=False), 128,
                         lambda o: o("n_col_strips") == 1 and (o("segments_unsorted") > 0 or o("n_chunks") == 0 or o("segments_unsorted") == -1)),   # columns in random order: no strips, still every bit
    "rmat20-unpermuted": (lambda dev: synth.csr_rmat_device(20, dev), 256,
                          lambda o: o("column_front_pct") >= 50 and o("n_col_strips") == 4 and o("n_hub_rows") > 0),                     # hubs-first order: four wide strips (+16 % before)
    "ddi-community": (lambda dev: synth.csr_dataset_structured_device("ddi", dev), 256,
                      lambda o: o("n_hub_rows") == 0),                                                                                 # L2-resident B: the longest row hides as a segment (+33 % before)
}


@pytest.mark.parametrize("name", list(_STRUCTURED))
def test_structured_graphs_default_options_bit_identical_to_reference_kernel(device, oracle, name):
    import torch
    from hpc_amd import CSR, SpMMOpt
    from hpc_amd.spmm import count_bitdiff, fill_normal

    build, N, plan_ok = _STRUCTURED[name]
    d_ptr, d_idx = build(device)
    M, nnz = d_ptr.numel() - 1, int(d_idx.numel())
    d_val = torch.empty(nnz, dtype=torch.float32, device=device)
    fill_normal(d_val, 124)
    d_B = torch.empty(M * N, dtype=torch.float32, device=device)
    fill_normal(d_B, 125)
    d_B = d_B.view(M, N)
    d_C = torch.full((M, N), float("nan"), dtype=torch.float32, device=device)
    op = SpMMOpt(CSR(M, nnz, d_ptr, d_idx, d_val), N)
    op.set_option("fused_step", 2)              # the library's defaults throughout (conftest's session default is 0)
    op.preprocess(d_B, d_C)
    assert plan_ok(op.get_option), {k: op.get_option(k) for k in ("n_col_strips", "segments_unsorted", "column_locality_pct", "column_front_pct", "medium_row_threshold",
                                                                  "long_row_threshold", "n_hub_rows", "n_chunks")}
    op.run(d_B, d_C)
    op.run(d_B, d_C)
    if not oracle.ref_available():
        pytest.fail("oracle/_ref missing")
    d_R = torch.full((M, N), float("nan"), dtype=torch.float32, device=device)
    oracle.ref_kernel_run(d_ptr, d_idx, d_val, d_B, d_R, M, N)
    assert count_bitdiff(d_C, d_R) == (0, 0.0), name


@pytest.mark.parametrize("N,deep", [(256, 16), (200, 16), (128, 8), (512, 8)])
def test_banded_long_rows_rows_kernel_depth_rule_against_the_reference_kernel(device, oracle, N, deep):
    """ "rows_unroll" auto: banded columns keep rows of hundreds of nonzeros in the rows kernel (medium threshold 1 024); with ONE whole-wave column tile a row is
    one wave's chain of round trips and the rule keeps 16 gathers in flight (profiles/r05_banded_unroll_ab.jsonl: 0.89 - 0.93 of the time), elsewhere 8.
    Scheduling only: whole C against spmm_kernel_ref either way, and the caller's explicit 8 is left alone."""
    import torch
    from hpc_amd import CSR, SpMMOpt
    from hpc_amd.spmm import count_bitdiff, fill_normal

    M = 1 << 16                  # (the banded medium-threshold rule starts at 65 536 rows)
    d_ptr, d_idx = synth.csr_banded_long_rows_device(M, device, width=1024, lo=300, hi=700, seed=21)
    nnz = int(d_idx.numel())
    d_val = torch.empty(nnz, dtype=torch.float32, device=device)
    fill_normal(d_val, 124, 0, 0.0, 0.1)
    d_B = torch.empty(M * N, dtype=torch.float32, device=device)
    fill_normal(d_B, 125, 0, 0.0, 0.1)
    d_B = d_B.view(M, N)
    if not oracle.ref_available():
        pytest.fail("oracle/_ref missing")
    d_R = torch.full((M, N), float("nan"), dtype=torch.float32, device=device)
    oracle.ref_kernel_run(d_ptr, d_idx, d_val, d_B, d_R, M, N)
    for forced in (0, 8):
        d_C = torch.full((M, N), float("nan"), dtype=torch.float32, device=device)
        op = SpMMOpt(CSR(M, nnz, d_ptr, d_idx, d_val), N)
        op.set_option("rows_unroll", forced)
        op.preprocess(d_B, d_C)
        assert op.get_option("column_locality_pct") >= 95 and op.get_option("n_chunks") == 0 and op.get_option("medium_row_threshold") == 1024
        op.run(d_B, d_C)
        torch.cuda.synchronize()
        assert op.get_option("rows_unroll_in_force") == (deep if forced == 0 else 8), (N, forced, op.get_option("rows_unroll_in_force"))
        assert count_bitdiff(d_C, d_R) == (0, 0.0)
