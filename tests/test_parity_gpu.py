"""Parity tests proper: the HIP path, called through the C ABI, against the oracle.

Bar (BASELINE.json north_star): bit-exact against the reference's stored-order fma chain on EVERY row with default
options (short rows, medium segments, hub rows through the hub kernel, block groups through the f32 MFMA).  Only the
opt-in "split_long_rows" = 1 changes a summation order: those rows are bit-exact against the oracle evaluated in the
same piece order and within 1e-5 * sum|a_k b_k| of the plain oracle."""
import glob
import os

import numpy as np
import pytest

from conftest import auto_hub_threshold, expected, bits, run_spmm, to_dev
from hpc_amd import synth

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
TOL_SPLIT = 1e-5  # |c - c_oracle| <= TOL_SPLIT * sum_k |a_k b_k|  (order-changing path only)


def _rand_case(M, K, N, deg_lo, deg_hi, seed):
    ptr, idx = synth.csr_uniform(M, deg_lo, deg_hi, K=K, seed=seed)
    return ptr, idx, synth.normal_f32(idx.size, seed + 1), synth.normal_f32(K * N, seed + 2).reshape(K, N)


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLDEN, "*.npz"))), ids=lambda p: os.path.basename(p))
def test_goldens_bitwise(device, path):
    z = np.load(path)
    C, _ = run_spmm(device, z["row_ptr"], z["col_idx"], z["vals"], z["B"], options={"long_row_threshold": 1 << 30})
    assert np.array_equal(bits(C), bits(z["C_ref_kernel"]))


@pytest.mark.parametrize("N", [1, 2, 3, 4, 5, 8, 12, 31, 32, 33, 64, 100, 128, 132, 256, 257, 260, 300, 512, 602, 1024, 1433])
def test_bitwise_vs_oracle_over_feature_widths(device, oracle, N):
    M, K = 777, 513
    ptr, idx, vals, B = _rand_case(M, K, N, 0, 70, seed=100 + N)
    ref = oracle.spmm_omp(ptr, idx, vals, B)
    C, op = run_spmm(device, ptr, idx, vals, B)
    assert not np.isnan(C).any(), "output not fully overwritten"
    assert np.array_equal(bits(C), bits(ref)), f"{(bits(C) != bits(ref)).sum()} elements differ"
    assert op.get_option("vector_width") == (4 if N >= 4 else 1)   # 16 B per lane at any width >= 4 (dword-aligned dwordx4)


@pytest.mark.parametrize("block_threads", [64, 128, 256])
@pytest.mark.parametrize("pol", [0, 1, 2, 3])
@pytest.mark.parametrize("N", [32, 128, 256])
def test_bitwise_over_tuning_knobs(device, oracle, block_threads, pol, N):
    ptr, idx, vals, B = _rand_case(2000, 2000, N, 0, 90, seed=7)
    ref = oracle.spmm_omp(ptr, idx, vals, B)
    for rpb, xcd in ((0, 1), (8, 0), (16, 1), (1000, 0), (1000, 1)):
        C, _ = run_spmm(device, ptr, idx, vals, B, options={
            "block_threads": block_threads, "nt_store": pol & 1, "nt_stream": (pol >> 1) & 1, "rows_per_block": rpb, "xcd_remap": xcd})
        assert np.array_equal(bits(C), bits(ref)), (rpb, xcd)


@pytest.mark.parametrize("split", [0, 1])
@pytest.mark.parametrize("N", [64, 128, 256, 300, 1024])
def test_bitwise_over_column_tile_widths(device, oracle, N, split):
    """"tile_cols" (widest column tile of the rows / segment kernels; auto picks 64 for wide B with random columns and
    whole-wave tiles for banded structure) is scheduling only: every width gives the oracle's bits -- short rows, medium
    rows (one exact segment), hub rows (stored order through the hub kernel; with "split_long_rows" the documented piece order)."""
    ptr, idx = synth.csr_powerlaw(3000, 30.0, 1200, K=20000, seed=77)
    vals = synth.normal_f32(idx.size, 78)
    B = synth.normal_f32(20000 * N, 79).reshape(20000, N)
    exp = expected(oracle, ptr, idx, vals, B, split, 300, 64)
    seen = set()
    for tile in (0, 32, 64, 128, 256):
        C, op = run_spmm(device, ptr, idx, vals, B, options={"tile_cols": tile, "long_row_threshold": 300, "long_row_chunk": 64, "split_long_rows": split})
        assert np.array_equal(bits(C), bits(exp)), tile
        seen.add(op.get_option("lanes_per_row"))
    assert len(seen) >= (2 if N == 64 else 3)
    assert op.get_option("n_long_rows") > 0 and op.get_option("n_medium_rows") > 0
    # the auto rule: random columns over K = 20000 are not "local"; a banded matrix is
    assert 0 <= op.get_option("column_locality_pct") < 50
    bp, bi = synth.csr_banded(30000, 4, 12, width=64, seed=5)
    Cb, opb = run_spmm(device, bp, bi, synth.normal_f32(bi.size, 1), synth.normal_f32(30000 * 256, 2).reshape(30000, 256))
    assert opb.get_option("column_locality_pct") > 90 and opb.get_option("lanes_per_row") == 64


def test_edge_shapes(device, oracle):
    # M = 1; all rows empty; a single nonzero; K != M; nnz = 0
    for (M, K, N, lo, hi, seed) in [(1, 1, 4, 1, 1, 1), (5, 9, 8, 0, 0, 2), (1, 300, 128, 200, 200, 3), (300, 7, 16, 0, 7, 4), (64, 64, 128, 64, 64, 5)]:
        ptr, idx, vals, B = _rand_case(M, K, N, lo, hi, seed)
        C, _ = run_spmm(device, ptr, idx, vals, B)
        assert np.array_equal(bits(C), bits(oracle.spmm_omp(ptr, idx, vals, B))), (M, K, N)


def test_empty_matrix_and_zero_width(device):
    import torch
    from hpc_amd import CSR, SpMMOpt

    d_ptr = torch.zeros(1, dtype=torch.int32, device=device)
    d_idx = torch.zeros(0, dtype=torch.int32, device=device)
    d_val = torch.zeros(0, dtype=torch.float32, device=device)
    d_B = torch.zeros(0, dtype=torch.float32, device=device)
    op = SpMMOpt(CSR(0, 0, d_ptr, d_idx, d_val), 16)
    op.preprocess(d_B, d_B)
    op.run(d_B, d_B)
    torch.cuda.synchronize()


@pytest.mark.parametrize("split", [0, 1])
def test_overwrite_and_idempotent(device, oracle, split):
    """run() leaves vout = A*vin whatever vout held (spmm_ref.cu:15, cuSPARSE beta=0) and may be
    called back to back (util.h:143-149) -- unlike the student kernel, which accumulates (H7)."""
    import torch
    from hpc_amd import CSR, SpMMOpt

    ptr, idx, vals, B = _rand_case(3000, 3000, 64, 0, 600, seed=9)   # includes hub rows (> 512)
    assert ptr[-1] == idx.size
    d_ptr, d_idx, d_val, d_B = to_dev(device, ptr, idx, vals, B)
    d_C = torch.full((3000, 64), 1e30, dtype=torch.float32, device=device)
    op = SpMMOpt(CSR(3000, idx.size, d_ptr, d_idx, d_val), 64)
    op.set_option("long_row_threshold", 512)
    op.set_option("split_long_rows", split)
    op.preprocess(d_B, d_C)
    assert op.get_option("n_long_rows") > 0 and op.get_option("n_medium_rows") > 0
    op.run(d_B, d_C)
    first = d_C.clone()
    for _ in range(3):
        op.run(d_B, d_C)
    torch.cuda.synchronize()
    assert torch.equal(first.view(torch.int32), d_C.view(torch.int32))
    exp = expected(oracle, ptr, idx, vals, B, split, 512, 256)
    assert np.array_equal(bits(d_C.cpu().numpy()), bits(exp))


@pytest.mark.parametrize("thr,chunk", [(8, 8), (16, 5), (64, 64), (100, 256)])
@pytest.mark.parametrize("N", [5, 32, 128, 256])
def test_split_rows_chunk_order_and_tolerance(device, oracle, thr, chunk, N):
    """The OPT-IN order-changing path ("split_long_rows" = 1): rows longer than the threshold are summed piece by piece in
    piece order -- bit-exact against the oracle evaluated in that order, within 1e-5 * sum|a*b| of the plain chain; the
    plain relative error north_star names is printed as statistics (it is unbounded near cancellation).
    The DEFAULT path on the same input keeps every row in stored order: bit-identical to the plain oracle."""
    ptr, idx, vals, B = _rand_case(400, 900, N, 0, 300, seed=31)
    C, op = run_spmm(device, ptr, idx, vals, B, options={"long_row_threshold": thr, "long_row_chunk": chunk, "split_long_rows": 1})
    assert op.get_option("n_long_rows") == int((np.diff(ptr) > thr).sum())
    exp = oracle.spmm_chunked(ptr, idx, vals, B, thr, chunk)
    assert np.array_equal(bits(C), bits(exp)), "device chunk order differs from the documented one"
    plain = oracle.spmm_omp(ptr, idx, vals, B)
    _, sabs = oracle.spmm_f64(ptr, idx, vals, B)
    assert (np.abs(C.astype(np.float64) - plain) <= TOL_SPLIT * sabs + 1e-30).all()
    short = np.diff(ptr) <= thr
    assert np.array_equal(bits(C[short]), bits(plain[short])), "rows below the threshold must stay bit-exact"
    rel = np.abs(C[~short].astype(np.float64) - plain[~short]) / np.maximum(np.abs(plain[~short].astype(np.float64)), 1e-300)
    if rel.size:
        print(f"split rows thr={thr} chunk={chunk} N={N}: plain relative error max {rel.max():.3e}, p99.9 {np.quantile(rel, 0.999):.3e}, "
              f"share > 1e-5: {(rel > 1e-5).mean():.4f}")
    C0, op0 = run_spmm(device, ptr, idx, vals, B, options={"long_row_threshold": thr, "long_row_chunk": chunk})
    assert op0.get_option("split_long_rows") == 0 and op0.get_option("n_partial_slots") == 0
    assert op0.get_option("n_hub_rows") == (int((np.diff(ptr) > thr).sum()) if N >= 4 else 0)
    assert np.array_equal(bits(C0), bits(plain)), "the default path must keep the stored order on every row"


@pytest.mark.parametrize("N", [4, 5, 31, 32, 33, 64, 100, 128, 260])
def test_hub_kernel_stage_and_ring_boundaries(device, oracle, N):
    """Hub rows whose lengths sit on and around the hub kernel's internal boundaries -- the 64-nonzero stage (63 / 64 / 65), the
    half stage (31 / 33), the three loaders' round (191 / 192 / 193), the six-slot ring wrapping (383 / 384 / 385, 449), many
    trips (1 000, 5 000) -- for every slice width, with unsorted and repeated columns, next to short and medium rows:
    always the plain oracle's bits."""
    lens = [63, 64, 65, 31, 33, 127, 128, 129, 191, 192, 193, 383, 384, 385, 449, 1000, 5000, 40, 3, 0, 70, 256, 257]
    K = 3000
    g = np.random.Generator(np.random.Philox(key=[515, N]))
    ptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    idx = g.integers(0, K, size=int(ptr[-1])).astype(np.int32)          # unsorted, with repeats
    vals = synth.normal_f32(idx.size, 516)
    B = synth.normal_f32(K * N, 517).reshape(K, N)
    exp = oracle.spmm_omp(ptr, idx, vals, B)
    for opts in ({"hub_slice": 0}, {"hub_slice": 16}, {"hub_slice": 32}, {"hub_slice": 64}, {"hub_overlap": 2}):
        o = {"long_row_threshold": 30, "medium_row_threshold": 8}
        o.update(opts)
        C, op = run_spmm(device, ptr, idx, vals, B, options=o, num_cols=K)
        assert op.get_option("n_hub_rows") == sum(1 for x in lens if x > 30) and op.get_option("n_partial_slots") == 0
        assert np.array_equal(bits(C), bits(exp)), (opts, [lens[i] for i in np.nonzero((bits(C) != bits(exp)).any(axis=1))[0]])


def test_row_panels_without_hub_rows_skip_the_hub_launch(device, oracle):
    """run_rows on a range that holds no hub row launches no hub grid (and forks no side stream for it): the handle keeps a
    row-ordered host copy of the hub list.  The multi-GPU step calls run_rows once per row panel."""
    import torch
    from hpc_amd import CSR, SpMMOpt

    lens = [5] * 400
    for r in (7, 130, 131):
        lens[r] = 900
    K, N = 2000, 64
    g = np.random.Generator(np.random.Philox(key=[77, 4]))
    ptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    idx = g.integers(0, K, size=int(ptr[-1])).astype(np.int32)
    vals = synth.normal_f32(idx.size, 78)
    B = synth.normal_f32(K * N, 79).reshape(K, N)
    exp = oracle.spmm_omp(ptr, idx, vals, B)
    d_ptr, d_idx, d_val, d_B = to_dev(device, ptr, idx, vals, B)
    for overlap in (0, 2):
        d_C = torch.full((400, N), float("nan"), dtype=torch.float32, device=device)
        op = SpMMOpt(CSR(400, idx.size, d_ptr, d_idx, d_val), N, num_cols=K)
        op.set_option("long_row_threshold", 256)
        op.set_option("hub_overlap", overlap)
        op.preprocess(d_B, d_C)
        assert op.get_option("n_hub_rows") == 3 and op.get_option("n_chunks") == 0
        launches = {}
        for r0, r1 in ((0, 7), (7, 8), (8, 130), (130, 132), (132, 400)):
            op.run_rows(d_B, N, d_C, N, r0, r1)
            launches[(r0, r1)] = op.get_option("n_launches")
        torch.cuda.synchronize()
        assert launches == {(0, 7): 1, (7, 8): 2, (8, 130): 1, (130, 132): 2, (132, 400): 1}, launches
        assert np.array_equal(bits(d_C.cpu().numpy()), bits(exp))


def test_hub_rows_of_random_lengths(device, oracle):
    """Seeded random hub lengths (0 .. 30 000, with the multiples of the 64-nonzero stage and of the six-slot ring over-represented),
    random widths and slice widths, row-range calls: the chain wave's assembly loop (hub_chain_asm.inc) enters, wraps and leaves its
    six unrolled copies at every residue -- always the plain oracle's bits."""
    import torch
    from hpc_amd import CSR, SpMMOpt

    n_cases = int(os.environ.get("MI_SPMM_HUB_FUZZ_CASES", "6"))          # soak: MI_SPMM_HUB_FUZZ_CASES=300
    g = np.random.Generator(np.random.Philox(key=[6464, int(os.environ.get("MI_SPMM_FUZZ_SEED", "1"))]))
    for case in range(n_cases):
        K = int(g.integers(2000, 9000))
        N = int(g.choice([4, 8, 17, 32, 48, 64, 100, 128, 200, 256]))
        lens = []
        for _ in range(int(g.integers(4, 14))):
            kind = int(g.integers(0, 4))
            if kind == 0:
                lens.append(int(g.integers(0, 30000)))
            elif kind == 1:
                lens.append(64 * int(g.integers(0, 200)) + int(g.choice([0, 0, 1, 63, 31, 32, 33])))
            elif kind == 2:
                lens.append(384 * int(g.integers(0, 40)) + int(g.choice([0, 64, 65, 383, 1])))
            else:
                lens.append(int(g.integers(0, 80)))
        ptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
        idx = g.integers(0, K, size=int(ptr[-1])).astype(np.int32)
        vals = synth.normal_f32(idx.size, 6500 + case)
        B = synth.normal_f32(K * N, 6600 + case).reshape(K, N)
        exp = oracle.spmm_omp(ptr, idx, vals, B)
        M = len(lens)
        opts = {"long_row_threshold": int(g.choice([30, 64, 200])), "medium_row_threshold": 8, "hub_slice": (0, 16, 32, 64)[case % 4], "hub_overlap": (1, 2, 0)[case % 3], "segment_overlap": case % 2}
        d_ptr, d_idx, d_val, d_B = to_dev(device, ptr, idx, vals, B)
        d_C = torch.full((M, N), float("nan"), dtype=torch.float32, device=device)
        op = SpMMOpt(CSR(M, idx.size, d_ptr, d_idx, d_val), N, num_cols=K)
        for k, v in opts.items():
            op.set_option(k, v)
        op.preprocess(d_B, d_C)
        if case % 2 and M > 2:
            cut = int(g.integers(1, M))
            op.run_rows(d_B, N, d_C, N, 0, cut)
            op.run_rows(d_B, N, d_C, N, cut, M)
        else:
            op.run(d_B, d_C)
        torch.cuda.synchronize()
        got = d_C.cpu().numpy()
        assert op.get_option("n_hub_rows") == (sum(1 for x in lens if x > opts["long_row_threshold"]) if N >= 4 else 0)
        bad = np.nonzero((bits(got) != bits(exp)).any(axis=1))[0]
        assert bad.size == 0, (case, N, opts, [lens[i] for i in bad])


def test_hub_segment_and_short_rows_with_special_values(device, oracle):
    """inf / NaN / -0 / subnormals / near-overflow values in A and B on hub rows (the assembly chain loop), medium rows (segments)
    and short rows: the same bits as the plain oracle wherever it is a number, NaN wherever it is NaN -- no flush to zero, no
    reordering that would turn inf - inf into something else."""
    lens = [5000, 1000, 449, 384, 200, 129, 64, 40, 12, 3, 0, 700, 65]
    K, g = 3000, np.random.Generator(np.random.Philox(key=[818, 1]))
    ptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    idx = g.integers(0, K, size=int(ptr[-1])).astype(np.int32)
    special = np.array([np.inf, -np.inf, np.nan, -0.0, 0.0, 1e-40, -3e-42, 1.1754942e-38, 3.4e38, -3.4e38, 1e-30, -1e-30], np.float32)
    for N in (32, 100, 256):
        vals = synth.normal_f32(idx.size, 819)
        B = synth.normal_f32(K * N, 820).reshape(K, N).copy()
        vals[g.integers(0, vals.size, 300)] = special[g.integers(0, special.size, 300)]
        B.reshape(-1)[g.integers(0, B.size, 3000)] = special[g.integers(0, special.size, 3000)]
        ref = oracle.spmm_omp(ptr, idx, vals, B)
        assert np.isnan(ref).any() and np.isinf(ref).any()
        for opts in ({"hub_slice": 0}, {"hub_slice": 16}, {"hub_slice": 64}):
            o = {"long_row_threshold": 128, "medium_row_threshold": 8}
            o.update(opts)
            C, op = run_spmm(device, ptr, idx, vals, B, options=o, num_cols=K)
            assert op.get_option("n_hub_rows") == sum(1 for x in lens if x > 128) and op.get_option("n_medium_rows") > 0
            same = (bits(C) == bits(ref)) | (np.isnan(C) & np.isnan(ref))
            assert same.all(), (N, opts, int((~same).sum()))


def test_power_law_rows(device, oracle):
    ptr, idx = synth.csr_powerlaw(20000, 32.0, 4096, seed=5)
    vals = synth.normal_f32(idx.size, 6)
    B = synth.normal_f32(20000 * 128, 7).reshape(20000, 128)
    plain = oracle.spmm_omp(ptr, idx, vals, B)
    C, op = run_spmm(device, ptr, idx, vals, B)
    thr = op.get_option("long_row_threshold")          # auto (exact mode): plan.hpp resolve_hub_threshold
    assert thr == auto_hub_threshold(20000, 128, ptr) < 4000 and op.get_option("n_hub_rows") == op.get_option("n_long_rows") > 0
    assert np.array_equal(bits(C), bits(plain)), "default options: hubs through the hub kernel, stored order"
    Cs, ops = run_spmm(device, ptr, idx, vals, B, options={"split_long_rows": 1})
    thr_s = ops.get_option("long_row_threshold")       # auto (split mode): clamp(nnz / 8192, 256, 2048)
    assert thr_s == min(2048, max(256, idx.size // 8192))
    assert np.array_equal(bits(Cs), bits(oracle.spmm_chunked(ptr, idx, vals, B, thr_s, 256)))
    C5, op5 = run_spmm(device, ptr, idx, vals, B, options={"long_row_threshold": 100, "long_row_chunk": 70, "split_long_rows": 1})
    assert op5.get_option("n_long_rows") > op.get_option("n_long_rows")
    assert np.array_equal(bits(C5), bits(oracle.spmm_chunked(ptr, idx, vals, B, 100, 70)))
    C6, op6 = run_spmm(device, ptr, idx, vals, B, options={"long_row_threshold": 100})
    assert op6.get_option("n_hub_rows") == op5.get_option("n_long_rows") and np.array_equal(bits(C6), bits(plain))
    C2, _ = run_spmm(device, ptr, idx, vals, B, options={"long_row_threshold": 1 << 30})
    assert np.array_equal(bits(C2), bits(oracle.spmm_omp(ptr, idx, vals, B)))


def test_block_dense_rows(device, oracle):
    ptr, idx = synth.csr_block_dense_fast(4096)
    vals = synth.normal_f32(idx.size, 6)
    B = synth.normal_f32(4096 * 256, 7).reshape(4096, 256)
    C, _ = run_spmm(device, ptr, idx, vals, B)
    assert np.array_equal(bits(C), bits(oracle.spmm_omp(ptr, idx, vals, B)))


def test_column_slices_with_pitches(device, oracle):
    """run_ld: a shard reads its column block of a wider B and writes its block of a wider C."""
    import torch
    from hpc_amd import CSR, SpMMOpt

    M, N, G = 500, 256, 4
    ptr, idx, vals, B = _rand_case(M, M, N, 0, 50, seed=41)
    d_ptr, d_idx, d_val, d_B = to_dev(device, ptr, idx, vals, B)
    d_C = torch.full((M, N), float("nan"), dtype=torch.float32, device=device)
    n_loc = N // G
    op = SpMMOpt(CSR(M, idx.size, d_ptr, d_idx, d_val), n_loc)
    op.preprocess(d_B, d_C)
    for g in range(G):
        op.run_ld(d_B.view(-1)[g * n_loc:], N, d_C.view(-1)[g * n_loc:], N)
    torch.cuda.synchronize()
    assert np.array_equal(bits(d_C.cpu().numpy()), bits(oracle.spmm_omp(ptr, idx, vals, B)))


def test_unpack_gathered(device):
    import torch
    from hpc_amd.spmm import unpack_gathered

    for rows, G, n_loc in [(37, 4, 32), (5, 3, 6), (64, 8, 128)]:
        st = torch.randn(G, rows, n_loc, device=device)
        C = torch.full((rows, G * n_loc), float("nan"), device=device)
        unpack_gathered(st, C, rows, G, n_loc, G * n_loc)
        torch.cuda.synchronize()
        assert torch.equal(C, st.permute(1, 0, 2).reshape(rows, G * n_loc))


def test_validator_matches_reference_rules(device, oracle):
    import torch
    from hpc_amd import valid

    g = np.random.Generator(np.random.Philox(key=[3, 3]))
    y = g.standard_normal(100000).astype(np.float32)
    y2 = (y * (1 + g.standard_normal(100000) * 0.008)).astype(np.float32)
    y[:50] = 0.0
    y2[:25] = 0.0       # 0/0 -> NaN, not counted ; x/0 -> inf, counted
    d_y, d_y2 = to_dev(device, y, y2)
    assert valid(d_y, d_y2, y.size) == oracle.valid_float(y, y2)
    a = g.integers(0, 5, 10000).astype(np.int32)
    b = g.integers(0, 5, 10000).astype(np.int32)
    d_a, d_b = to_dev(device, a, b)
    assert valid(d_a, d_b, a.size) == oracle.valid_int(a, b) == int((a != b).sum())
    if oracle.ref_available():   # the reference's own validator kernels, compiled by hipcc
        assert oracle.ref_valid(d_y, d_y2, y.size) == oracle.valid_float(y, y2)
        assert oracle.ref_valid(d_a, d_b, a.size) == oracle.valid_int(a, b)


@pytest.mark.parametrize("gpu_pre", [1, 0])
def test_malformed_csr_is_rejected_not_faulted(device, gpu_pre):
    import torch
    from hpc_amd import CSR, SpMMOpt, MiSpmmError

    M, N = 16, 8
    B = torch.zeros(M, N, device=device)
    Cc = torch.zeros(M, N, device=device)
    val = torch.ones(4, device=device)
    bad_cases = {
        "col out of range": (np.array([0, 1, 2, 3, 4] + [4] * 12, np.int32), np.array([0, 1, 2, 99], np.int32)),
        "negative col": (np.array([0, 1, 2, 3, 4] + [4] * 12, np.int32), np.array([0, -1, 2, 3], np.int32)),
        "non-monotone ptr": (np.array([0, 3, 2, 3, 4] + [4] * 12, np.int32), np.array([0, 1, 2, 3], np.int32)),
        "ptr[M] != nnz": (np.array([0, 1, 2, 3, 3] + [3] * 12, np.int32), np.array([0, 1, 2, 3], np.int32)),
    }
    for why, (ptr, idx) in bad_cases.items():
        d_ptr, d_idx = to_dev(device, ptr, idx)
        op = SpMMOpt(CSR(M, 4, d_ptr, d_idx, val), N)
        op.set_option("gpu_preprocess", gpu_pre)
        with pytest.raises(MiSpmmError) as e:
            op.preprocess(B, Cc)
        assert e.value.code == -4, why
        with pytest.raises(MiSpmmError):
            op.run(B, Cc)   # never prepared -> ESTATE, no launch
    # shaped like a block group (16 rows of equal length >= 8, N a multiple of 32) but pointing far outside
    # col_idx: block detection runs before row_ptr is validated on the device path and must not follow it
    B32 = torch.zeros(M, 32, device=device)
    C32 = torch.zeros(M, 32, device=device)
    val128 = torch.ones(128, device=device)
    idx128 = np.zeros(128, np.int32)
    for why, ptr in {"offsets beyond nnz": (400_000_000 + 8 * np.arange(17)).astype(np.int32),
                     "negative offsets": (-1_000_000 + 8 * np.arange(17)).astype(np.int32)}.items():
        d_ptr, d_idx = to_dev(device, ptr, idx128)
        op = SpMMOpt(CSR(M, 128, d_ptr, d_idx, val128), 32)
        op.set_option("gpu_preprocess", gpu_pre)
        with pytest.raises(MiSpmmError) as e:
            op.preprocess(B32, C32)
        assert e.value.code == -4, why


def test_reference_kernel_agrees_with_oracle_live(device, oracle):
    """Live pinning on the GPU box: the reference's spmm_kernel_ref (hipcc-compiled from the
    reference tree) == CPU restatement == our kernel, bit for bit, on a mid-size random case."""
    import torch

    if not oracle.ref_available():
        pytest.fail("oracle/_ref/libspmm_ref_gfx950.so did not travel to the GPU box")
    M, N = 20000, 64
    ptr, idx, vals, B = _rand_case(M, M, N, 0, 64, seed=77)
    d_ptr, d_idx, d_val, d_B = to_dev(device, ptr, idx, vals, B)
    d_C = torch.full((M, N), float("nan"), dtype=torch.float32, device=device)
    oracle.ref_kernel_run(d_ptr, d_idx, d_val, d_B, d_C, M, N)
    torch.cuda.synchronize()
    ref_gpu = d_C.cpu().numpy()
    cpu = oracle.spmm_omp(ptr, idx, vals, B)
    assert np.array_equal(bits(ref_gpu), bits(cpu))
    ours, _ = run_spmm(device, ptr, idx, vals, B)
    assert np.array_equal(bits(ours), bits(ref_gpu))
    # and the reference's own acceptance test (test_spmm.cu:43) passes with zero bad elements
    from hpc_amd import valid
    d_o = torch.from_numpy(ours).to(device)
    assert valid(d_o, d_C, M * N) == 0


# ---- block (MFMA) path: 16-row groups sharing one column list --------------------------------
def _shared_list_case(n_groups, K, N, seed, lens=None, tail_rows=5):
    """Groups of 16 rows sharing a random (unsorted, possibly repeated) column list of random
    length, interleaved with ordinary ragged rows; M is not a multiple of 16."""
    g = np.random.Generator(np.random.Philox(key=[seed, 0]))
    ptr = [0]
    idx = []
    kinds = []
    for b in range(n_groups):
        kind = int(g.integers(0, 3))            # 0: shared list, 1: ragged rows, 2: same length but one row differs
        L = int(lens[b % len(lens)]) if lens is not None else int(g.integers(1, 200))
        if kind == 0:
            cols = g.integers(0, K, size=L)
            for _ in range(16):
                idx.append(cols)
                ptr.append(ptr[-1] + L)
        elif kind == 1:
            for _ in range(16):
                d = int(g.integers(0, 40))
                idx.append(np.sort(g.choice(K, d, replace=False)))
                ptr.append(ptr[-1] + d)
        else:
            cols = g.integers(0, K, size=L)
            for r in range(16):
                c = cols.copy()
                if r == 11:
                    c[L // 2] = (c[L // 2] + 1) % K
                idx.append(c)
                ptr.append(ptr[-1] + L)
        kinds.append((kind, L))
    for _ in range(tail_rows):
        d = int(g.integers(1, 30))
        idx.append(np.sort(g.choice(K, d, replace=False)))
        ptr.append(ptr[-1] + d)
    ptr = np.asarray(ptr, np.int32)
    idx = np.concatenate(idx).astype(np.int32)
    vals = synth.normal_f32(idx.size, seed + 1)
    B = synth.normal_f32(K * N, seed + 2).reshape(K, N)
    return ptr, idx, vals, B, kinds


@pytest.mark.parametrize("N", [32, 64, 96, 128, 160, 192, 256, 384, 512])
def test_block_path_bitwise(device, oracle, N):
    ptr, idx, vals, B, kinds = _shared_list_case(60, 3000, N, seed=300 + N)
    ref = oracle.spmm_omp(ptr, idx, vals, B)
    C, op = run_spmm(device, ptr, idx, vals, B, options={"long_row_threshold": 2048})
    expect_groups = sum(1 for k, L in kinds if k == 0 and 8 <= L <= op.get_option("long_row_threshold"))
    assert op.get_option("n_block_groups") == expect_groups and expect_groups > 5
    assert not np.isnan(C).any()
    assert np.array_equal(bits(C), bits(ref)), f"{(bits(C) != bits(ref)).sum()} of {C.size} differ"
    C0, op0 = run_spmm(device, ptr, idx, vals, B, options={"block_path": 0})
    assert op0.get_option("n_block_groups") == 0
    assert np.array_equal(bits(C0), bits(ref))



@pytest.mark.parametrize("L", [8, 9, 11, 12, 31, 32, 33, 64, 65, 127, 256, 257, 500])
def test_block_path_list_lengths(device, oracle, L):
    """Every tail shape of the k loop (batches of 8/16/32/64 k-rows, MFMA k-steps of 4)."""
    for N in (32, 128, 256):
        ptr, idx, vals, B, kinds = _shared_list_case(9, 700, N, seed=900 + L, lens=[L])
        C, op = run_spmm(device, ptr, idx, vals, B, options={"long_row_threshold": 2048})
        assert op.get_option("n_block_groups") == sum(1 for k, _ in kinds if k == 0)
        assert np.array_equal(bits(C), bits(oracle.spmm_omp(ptr, idx, vals, B))), (L, N)


def test_block_path_on_block_dense_config(device, oracle):
    """BASELINE configs[4] family (down-sized): every row >= 64 contiguous nonzeros, N = 256."""
    M = 1 << 17
    ptr, idx = synth.csr_block_dense_fast(M)
    vals = synth.normal_f32(idx.size, 6)
    B = synth.normal_f32(M * 256, 7).reshape(M, 256)
    C, op = run_spmm(device, ptr, idx, vals, B)
    assert op.get_option("n_block_groups") == M // 16
    assert np.array_equal(bits(C), bits(oracle.spmm_omp(ptr, idx, vals, B)))


def test_block_path_with_row_panels_and_pitches(device, oracle):
    import torch
    from hpc_amd import CSR, SpMMOpt

    ptr, idx, vals, B, kinds = _shared_list_case(40, 2000, 128, seed=77)
    M = ptr.size - 1
    d_ptr, d_idx, d_val, d_B = to_dev(device, ptr, idx, vals, B)
    wide = torch.full((M, 256), float("nan"), dtype=torch.float32, device=device)
    op = SpMMOpt(CSR(M, idx.size, d_ptr, d_idx, d_val), 128, num_cols=2000)
    op.preprocess(d_B, wide)
    assert op.get_option("n_block_groups") > 0
    for r0, r1 in ((0, 200), (200, 456), (456, M)):
        op.run_rows(d_B, 128, wide.view(-1)[128:], 256, r0, r1)     # right half of a wider C
    torch.cuda.synchronize()
    got = wide[:, 128:].cpu().numpy()
    assert np.array_equal(bits(got), bits(oracle.spmm_omp(ptr, idx, vals, B)))
    assert torch.isnan(wide[:, :128]).all()


@pytest.mark.parametrize("N", [32, 128, 256])
@pytest.mark.parametrize("rpb", [0, 16, 40, 1000])
def test_block_groups_with_long_lane_group_runs(device, oracle, N, rpb):
    """A lane group of the rows kernel may own up to LPR - 1 = 63 consecutive rows ("rows_per_block" large), i.e. up
    to five 16-row groups, each of which is or is not owned by the MFMA block path: ownership is looked up per row.
    (Round-1 defect: only the first and the last group's flag were sampled -- middle rows were skipped or computed
    twice.)  Shared-list groups interleaved with ragged groups, every row NaN-poisoned beforehand."""
    ptr, idx, vals, B, kinds = _shared_list_case(120, 1500, N, seed=5150 + N, lens=[12, 40, 9, 64, 33])
    ref = oracle.spmm_omp(ptr, idx, vals, B)
    C, op = run_spmm(device, ptr, idx, vals, B, options={"rows_per_block": rpb, "long_row_threshold": 2048})
    assert op.get_option("n_block_groups") == sum(1 for k, _ in kinds if k == 0) > 20
    assert sum(1 for k, _ in kinds if k != 0) > 20          # ... next to groups the rows kernel owns
    assert not np.isnan(C).any(), f"{int(np.isnan(C).any(axis=1).sum())} rows left unwritten"
    assert np.array_equal(bits(C), bits(ref)), f"{(bits(C) != bits(ref)).any(axis=1).sum()} rows differ"


def test_product_library_refuses_ablation_options(device):
    """The first-generation rows kernel, the timing-only block-kernel builds and the B-stationary sweep experiment were
    removed from the tree in round 3 (record: profiles/r02_c4_block_path_notes.txt, git 0894343); the ABI has no option
    that can produce a wrong C."""
    import torch
    from hpc_amd import CSR, SpMMOpt
    from hpc_amd.spmm import MiSpmmError

    d_ptr = torch.zeros(2, dtype=torch.int32, device=device)
    e = torch.zeros(0, dtype=torch.float32, device=device)
    op = SpMMOpt(CSR(1, 0, d_ptr, torch.zeros(0, dtype=torch.int32, device=device), e), 8)
    for key, v in (("block_ablate", 2), ("kernel", 1), ("block_sweep", 1), ("block_merge_unsafe", 1)):
        with pytest.raises(MiSpmmError) as ei:
            op.set_option(key, v)
        assert ei.value.code == -5          # MI_SPMM_EUNSUPPORTED
    with pytest.raises(MiSpmmError) as ei:
        op.set_option("long_row_chunk", 1 << 21)
    assert ei.value.code == -1              # MI_SPMM_EINVAL: piece length beyond what the int32 plan arithmetic allows
    op.set_option("kernel", 2)


def _run_groups_case(n_groups, K, N, seed, lens=(32, 36, 40, 64, 96, 128, 33, 200), slots=12, tail_rows=3, align=1, min_gap=0):
    """16-row groups whose shared column list is 1-3 RUNS of consecutive columns drawn from a small pool of start
    columns (so that runs of different groups coincide and get shared), next to groups with a too-short run, groups
    with a plain random list and ragged rows.  Returns (ptr, idx, vals, B, expected number of pieces per group kind)."""
    g = np.random.Generator(np.random.Philox(key=[seed, 0]))
    pitch = K // slots // align * align     # one possible run start per slot; a run never reaches the next slot's start
    assert pitch * max(min_gap, 1) > max(lens) + 2      # min_gap > 1: runs of DIFFERENT groups overlap partially, a group's own runs never
    ptr, idx, n_runs_of = [0], [], []
    for b in range(n_groups):
        kind = int(g.integers(0, 6))        # 0-3: run groups, 4: random list, 5: ragged rows
        if kind <= 3:
            nr = int(g.integers(1, 4))
            sl = np.sort(g.choice(slots, nr, replace=False))
            while min_gap > 1 and nr > 1 and np.diff(sl).min() < min_gap:
                sl = np.sort(g.choice(slots, nr, replace=False))
            cols = []
            for q in sl:
                L = int(g.choice(lens)) if kind != 3 else int(g.choice([5, 64, 12]))   # kind 3: may hold a run < 32
                cols.append(np.arange(q * pitch, q * pitch + L))
            cols = np.concatenate(cols).astype(np.int32)
            n_runs_of.append(nr)
            for _ in range(16):
                idx.append(cols)
                ptr.append(ptr[-1] + cols.size)
        elif kind == 4:
            cols = g.integers(0, K, size=int(g.integers(8, 150))).astype(np.int32)
            for _ in range(16):
                idx.append(cols)
                ptr.append(ptr[-1] + cols.size)
        else:
            for _ in range(16):
                d = int(g.integers(0, 40))
                idx.append(np.sort(g.choice(K, d, replace=False)).astype(np.int32))
                ptr.append(ptr[-1] + d)
    for _ in range(tail_rows):
        d = int(g.integers(1, 30))
        idx.append(np.sort(g.choice(K, d, replace=False)).astype(np.int32))
        ptr.append(ptr[-1] + d)
    ptr = np.asarray(ptr, np.int32)
    idx = np.concatenate(idx).astype(np.int32)
    return ptr, idx, synth.normal_f32(idx.size, seed + 1), synth.normal_f32(K * N, seed + 2).reshape(K, N)


@pytest.mark.parametrize("N", [32, 64, 128, 256, 384, 512])
def test_block_path_shared_runs_and_passes(device, oracle, N):
    """The block path's items: runs shared by several groups (one staged B tile feeding two groups' MFMAs), lists cut
    into up to three runs = up to three passes whose fma chains continue through C, runs whose length is not a
    multiple of the MFMA k-step (never shared), lists with a short run (kept whole).  Always the oracle's bits, and
    the same bits with sharing / cutting switched off."""
    ptr, idx, vals, B = _run_groups_case(160, 3000, N, seed=7100 + N)
    ref = oracle.spmm_omp(ptr, idx, vals, B)
    C, op = run_spmm(device, ptr, idx, vals, B, options={"long_row_threshold": 2048})
    assert op.get_option("n_block_groups") > 80
    assert op.get_option("n_block_passes") == 3
    assert op.get_option("n_block_pieces") > op.get_option("n_block_groups")
    if N % 128 == 0:
        assert op.get_option("n_block_shared_items") > 20
    else:
        assert op.get_option("n_block_shared_items") == 0        # the two-piece kernels exist for 128/256-column slabs
    assert not np.isnan(C).any()
    assert np.array_equal(bits(C), bits(ref)), f"{(bits(C) != bits(ref)).any(axis=1).sum()} rows differ"
    for opts in ({"block_share": 1}, {"block_max_pieces": 1}, {"block_max_pieces": 2, "block_run_min": 64}, {"block_path": 0}):
        o = {"long_row_threshold": 2048}
        o.update(opts)
        C2, op2 = run_spmm(device, ptr, idx, vals, B, options=o)
        assert np.array_equal(bits(C2), bits(ref)), opts
        if "block_share" in opts:
            assert op2.get_option("n_block_shared_items") == 0
        if opts.get("block_max_pieces") == 1:
            assert op2.get_option("n_block_passes") == 1 and op2.get_option("n_block_pieces") == op2.get_option("n_block_groups")


def test_block_items_row_panels_pitches_and_special_values(device, oracle):
    """Shared items whose pieces fall on different sides of a row-range boundary (the multi-GPU driver's row panels),
    C with a row pitch wider than N, and inf / NaN / -0 / subnormals in A and B: a shorter piece of a shared item
    must not multiply the longer piece's extra B rows (0 * inf would poison it), carried tiles must round-trip
    exactly."""
    import torch
    from hpc_amd import CSR, SpMMOpt

    g = np.random.Generator(np.random.Philox(key=[77, 9]))
    ptr, idx, vals, B = _run_groups_case(120, 2600, 256, seed=4321)
    special = np.array([np.inf, -np.inf, np.nan, -0.0, 0.0, 1e-40, -3e-42, 1.1754942e-38, 3.4e38, -3.4e38], np.float32)
    vals[g.integers(0, vals.size, 600)] = special[g.integers(0, special.size, 600)]
    B.reshape(-1)[g.integers(0, B.size, 6000)] = special[g.integers(0, special.size, 6000)]
    ref = oracle.spmm_omp(ptr, idx, vals, B)
    assert np.isnan(ref).any() and np.isinf(ref).any()
    M = ptr.size - 1
    d_ptr, d_idx, d_val, d_B = to_dev(device, ptr, idx, vals, B)
    op = SpMMOpt(CSR(M, idx.size, d_ptr, d_idx, d_val), 256, num_cols=2600)
    op.set_option("long_row_threshold", 2048)
    wide = torch.full((M, 512), float("nan"), dtype=torch.float32, device=device)
    op.preprocess(d_B, wide)
    assert op.get_option("n_block_shared_items") > 10 and op.get_option("n_block_passes") == 3
    cuts = [0, 16, 200, 203, 640, 1111, M]
    for r0, r1 in zip(cuts, cuts[1:]):
        op.run_rows(d_B, 256, wide.view(-1)[256:], 512, r0, r1)       # right half of a wider C, ragged row ranges
    torch.cuda.synchronize()
    got = wide[:, 256:].cpu().numpy()
    same = (bits(got) == bits(ref)) | (np.isnan(got) & np.isnan(ref))
    assert same.all(), int((~same).sum())
    assert torch.isnan(wide[:, :256]).all()
    # exactly the requested rows are written, also when the range cuts a group and an item
    C = torch.full((M, 256), float("nan"), dtype=torch.float32, device=device)
    op.run_rows(d_B, 256, C, 256, 37, 999)
    torch.cuda.synchronize()
    got = C.cpu().numpy()
    same = (bits(got[37:999]) == bits(ref[37:999])) | (np.isnan(got[37:999]) & np.isnan(ref[37:999]))
    assert same.all()
    assert np.isnan(got[:37]).all() and np.isnan(got[999:]).all()


@pytest.mark.parametrize("split", [0, 1])
@pytest.mark.parametrize("N", [32, 128, 36])
def test_run_rows_writes_exactly_the_range_for_every_row_class(device, oracle, N, split):
    """mi_spmm_run_rows on an arbitrary range: exactly those rows of C are written -- short rows, medium rows
    (segment kernel), hub rows (hub kernel; or pieces + ordered reduce) and rows of block groups cut by the range boundary."""
    import torch
    from hpc_amd import CSR, SpMMOpt

    ptr, idx, vals, B, kinds = _shared_list_case(30, 2500, N, seed=4242)
    # append hub and medium rows so every class is present
    g = np.random.Generator(np.random.Philox(key=[4243, 0]))
    extra = [int(x) for x in (700, 90, 1500, 70, 3, 0, 300)]
    cols = [np.sort(g.choice(2500, d, replace=False)).astype(np.int32) for d in extra]
    ptr = np.concatenate([ptr, ptr[-1] + np.cumsum(extra)]).astype(np.int32)
    idx = np.concatenate([idx] + cols).astype(np.int32)
    vals = synth.normal_f32(idx.size, 11)
    M = ptr.size - 1
    d_ptr, d_idx, d_val, d_B = to_dev(device, ptr, idx, vals, B)
    op = SpMMOpt(CSR(M, idx.size, d_ptr, d_idx, d_val), N, num_cols=2500)
    op.set_option("long_row_threshold", 256)
    op.set_option("medium_row_threshold", 32)      # (the row classes are the subject here, not the auto rule: 2 500 columns all count as "local", auto = 256)
    op.set_option("split_long_rows", split)
    op.set_option("hub_overlap", 2)
    full = torch.empty(M, N, dtype=torch.float32, device=device)
    op.preprocess(d_B, full)
    assert op.get_option("n_long_rows") >= 3 and op.get_option("n_medium_rows") >= 2
    if N % 32 == 0:
        assert op.get_option("n_block_groups") > 0
    op.run(d_B, full)
    ref_full = full.cpu().numpy()
    # split rows carry the documented tolerance against the oracle; everything else is bit-exact
    exp = oracle.spmm_omp(ptr, idx, vals, B)
    unsplit = (np.diff(ptr) <= 256) if split else np.ones(M, bool)
    assert np.array_equal(bits(ref_full)[unsplit], bits(exp)[unsplit])
    for r0, r1 in ((0, M), (5, 6), (7, 41), (100, 333), (M - 9, M), (M - 7, M - 2), (17, 17)):
        C = torch.full((M, N), float("nan"), dtype=torch.float32, device=device)
        op.run_rows(d_B, N, C, N, r0, r1)
        torch.cuda.synchronize()
        got = C.cpu().numpy()
        assert np.array_equal(bits(got[r0:r1]), bits(ref_full[r0:r1])), (r0, r1)
        outside = np.ones(M, bool)
        outside[r0:r1] = False
        assert np.isnan(got[outside]).all(), f"rows outside [{r0},{r1}) were written"
    # a ragged cover of [0, M): same C as one call
    C = torch.full((M, N), float("nan"), dtype=torch.float32, device=device)
    cuts = [0, 3, 50, 51, 190, 402, M]
    for r0, r1 in zip(cuts, cuts[1:]):
        op.run_rows(d_B, N, C, N, r0, r1)
    torch.cuda.synchronize()
    assert np.array_equal(bits(C.cpu().numpy()), bits(ref_full))


def test_gather_pipeline_on_gpu_streams(device, oracle):
    """The multi-GPU step's device side (compute stream / comm stream / staging / unpack kernel),
    rehearsed on one GPU with a world_size-1 RCCL group forced through the collective path."""
    import os
    import torch
    import torch.distributed as dist
    from hpc_amd import CSR, SpMMOpt
    from hpc_amd.dist import ColumnShardedSpMM, ShardLayout
    from hpc_amd.spmm import unpack_gathered

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    created = not dist.is_initialized()
    if created:
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=device)
    try:
        M, n_loc = 5000, 128
        ptr, idx = synth.csr_powerlaw(M, 24.0, 2000, seed=8)       # includes hub rows
        vals = synth.normal_f32(idx.size, 9)
        B = synth.normal_f32(M * n_loc, 10).reshape(M, n_loc)
        d_ptr, d_idx, d_val, d_B = to_dev(device, ptr, idx, vals, B)
        op = SpMMOpt(CSR(M, idx.size, d_ptr, d_idx, d_val), n_loc)
        op.set_option("long_row_threshold", 300)
        C_loc = torch.empty(M, n_loc, device=device)
        C_full = torch.full((M, n_loc), float("nan"), device=device)
        op.preprocess(d_B, C_loc)
        sh = ColumnShardedSpMM(op, ShardLayout(M, n_loc, 1, 0), unpack_gathered, n_panels=5, force_collective=True)
        for _ in range(3):
            sh.run(d_B, C_loc, C_full)
        torch.cuda.synchronize()
        assert op.get_option("n_hub_rows") > 0
        exp = oracle.spmm_omp(ptr, idx, vals, B)
        assert np.array_equal(bits(C_full.cpu().numpy()), bits(exp))
    finally:
        if created:
            dist.destroy_process_group()


@pytest.mark.parametrize("exchange", ["allgather", "direct", "peer2d", "peer_store"])
def test_native_dist_step_world_one_rehearsal(device, oracle, exchange):
    """include/mi_spmm_dist.h on one GPU: a world of one with "rehearse" on runs the whole N > 1 machinery -- our own RCCL
    communicator (ncclCommInitRank from a unique id), the panel pipeline over three streams, the in-place all-gather /
    grouped send-recv into staging + the re-layout kernel, or the IPC / strided-copy path with its all-reduce
    barriers -- and must give the plain operator's C; repeated steps reuse the staging buffers."""
    import torch
    from hpc_amd import CSR, SpMMOpt
    from hpc_amd.dist import NativeColumnShardedSpMM, ShardLayout

    ptr, idx, vals, B, _ = _shared_list_case(70, 2500, 128, seed=99)      # block groups + ragged rows
    g = np.random.Generator(np.random.Philox(key=[9, 9]))
    extra = [np.sort(g.choice(2500, d, replace=False)).astype(np.int32) for d in (900, 120)]
    idx = np.concatenate([idx] + extra)
    ptr = np.concatenate([ptr, ptr[-1] + np.cumsum([e.size for e in extra])]).astype(np.int32)
    vals = synth.normal_f32(idx.size, 3)
    M = ptr.size - 1
    d_ptr, d_idx, d_val, d_B = to_dev(device, ptr, idx, vals, B)
    op = SpMMOpt(CSR(M, idx.size, d_ptr, d_idx, d_val), 128, num_cols=2500)
    op.set_option("long_row_threshold", 256)
    d_C = torch.full((M, 128), float("nan"), dtype=torch.float32, device=device)
    op.preprocess(d_B, d_C)
    from hpc_amd.dist import MiSpmmDistError
    for bad in (ShardLayout(M, 64, 1, 0), ShardLayout(M - 1, 128, 1, 0)):      # not this operator's columns / rows
        with pytest.raises(MiSpmmDistError) as ei:
            NativeColumnShardedSpMM(op, bad, n_panels=3, exchange=exchange)
        assert ei.value.code == -1
    sh = NativeColumnShardedSpMM(op, ShardLayout(M, 128, 1, 0), n_panels=3, exchange=exchange, rehearse=True)
    sh.init_comm()
    if exchange in ("peer2d", "peer_store"):
        sh.set_peers(d_C)
    assert sh.get_option("has_comm") == 1 and sh.get_option("n_panels") == 3
    for _ in range(3):
        sh.run(d_B, d_C)
    torch.cuda.synchronize()
    exp = oracle.spmm_omp(ptr, idx, vals, B)
    assert np.array_equal(bits(d_C.cpu().numpy()), bits(exp))
    assert (sh.get_option("staging_bytes") > 0) == (exchange not in ("peer2d", "peer_store"))
    assert sh.get_option("comm_stream_overlaps") == 1 and sh.get_option("post_stream_overlaps") == 1      # the exchange and re-layout streams run beside the compute stream (tested candidates)
    # the two legs on their own (bench.py's breakdown: timing legs; with staging the panels share two buffers, so only
    # the staging-free exchange leaves a complete C behind)
    d_C.fill_(float("nan"))
    sh.run_compute_only(d_B, d_C)
    sh.run_exchange_only(d_C)
    torch.cuda.synchronize()
    if exchange in ("peer2d", "peer_store"):
        assert np.array_equal(bits(d_C.cpu().numpy()), bits(exp))
    sh.set_option("n_panels", 1)
    d_C.fill_(float("nan"))
    sh.run(d_B, d_C)
    torch.cuda.synchronize()
    assert np.array_equal(bits(d_C.cpu().numpy()), bits(exp))


def test_native_dist_cpp_rehearsal(device):
    """tests/native/dist_rehearsal.cpp: the C ABI driven from C++ (no Python, no torch) -- world of one over RCCL."""
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "tests", "native", "dist_rehearsal")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-s", "-C", os.path.join(root, "tests", "native")])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    for ex in ("allgather", "direct", "peer2d", "peer_store"):
        assert f"exchange {ex}: bit-identical" in r.stdout, r.stdout


def test_device_fill_matches_restatement(device, oracle):
    """allocate<T>()'s fill (data.h:24-37): device Philox stream bit-exact, normals within libm/ocml rounding."""
    import torch
    from hpc_amd.spmm import allocate, fill_normal, fill_philox_u32

    for n in (1, 3, 4, 5, 1023, 1 << 16):
        u = torch.empty(n, dtype=torch.int32, device=device)
        fill_philox_u32(u, seed=123, subsequence=9)
        assert np.array_equal(u.cpu().numpy().view(np.uint32), oracle.fill_philox_u32(n, 123, 9))
        z = torch.full((n,), float("nan"), device=device)
        fill_normal(z, seed=77, subsequence=2, mean=0.5, stddev=2.0)
        assert np.abs(z.cpu().numpy() - oracle.fill_normal(n, 77, 2, 0.5, 2.0)).max() <= 2e-5   # stddev 2.0: 1e-6 * scale
    t = allocate(1000)                       # rounds up to 1024 floats, N(0, 0.1), seed 123
    assert t.numel() == 1024
    ref = oracle.fill_normal(1024, 123, 0, 0.0, 0.1)
    assert np.abs(t.cpu().numpy() - ref).max() <= 1e-6
    z = allocate(1 << 22).double()
    assert abs(z.mean().item()) < 3e-4 and abs(z.std().item() - 0.1) < 3e-4
    # unaligned start (a view one float in): scalar tail path
    buf = torch.zeros(4099, device=device)
    fill_normal(buf[1:4098], seed=1)
    torch.cuda.synchronize()
    assert buf[0].item() == 0.0 and buf[4098].item() == 0.0
    assert np.abs(buf[1:4098].cpu().numpy() - oracle.fill_normal(4097, 1)).max() <= 1e-6


def test_rocsparse_comparator_agrees(device, oracle):
    """SpMMCuSparse's counterpart (spmm_cusparse.cu:3-34): an independent GPU-side value check, held to the
    reference's own acceptance rule (valid.cu:6, test_spmm.cu:43)."""
    import torch
    from hpc_amd import CSR, SpMMOpt, valid
    from hpc_amd.comparator import SpMMRocSparse

    M, N = 30000, 64
    ptr, idx, vals, B = _rand_case(M, M, N, 0, 64, seed=55)
    d_ptr, d_idx, d_val, d_B = to_dev(device, ptr, idx, vals, B)
    g = CSR(M, idx.size, d_ptr, d_idx, d_val)
    d_C = torch.full((M, N), float("nan"), device=device)
    d_R = torch.full((M, N), float("nan"), device=device)
    ours = SpMMOpt(g, N)
    ours.preprocess(d_B, d_C)
    ours.run(d_B, d_C)
    vend = SpMMRocSparse(g, N)
    vend.preprocess(d_B, d_R)
    vend.run(d_B, d_R)
    vend.run(d_B, d_R)          # beta = 0: idempotent
    torch.cuda.synchronize()
    bad = valid(d_C, d_R, M * N)
    assert oracle.validation_passes(bad, M, N), bad
    _, sabs = oracle.spmm_f64(ptr, idx, vals, B)
    assert (np.abs(d_R.cpu().numpy().astype(np.float64) - oracle.spmm_omp(ptr, idx, vals, B)) <= 1e-5 * sabs + 1e-30).all()


def test_native_harness_end_to_end(device, tmp_path):
    """The reference's `unit_tests` flow (test/main.cpp + test/test_spmm.cu) re-stated natively over the C++
    adapter: course-format graph files in, gtest-shaped output and plot.py-parsable log lines out."""
    import re
    import subprocess
    from hpc_amd import graph_io

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "tests", "native", "unit_tests")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-s", "-C", os.path.join(root, "tests", "native")])
    ptr, idx = synth.csr_powerlaw(30000, 20.0, 1500, seed=12)
    graph_io.write_graph(str(tmp_path), "syn", ptr, idx, text=True, dumps=False)
    for n_len, has_cache in ((32, False), (256, True)):
        r = subprocess.run([exe, "--dataset", "syn", "--datadir", str(tmp_path), "--len", str(n_len)],
                           capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stdout + r.stderr
        assert "[  PASSED  ] 3 tests." in r.stdout
        for t in ("validation", "cusparse_performance", "opt_performance"):
            assert f"[       OK ] SpMMTest.{t}" in r.stdout
        # the reference's own log scrapers (plot.py:13-14)
        assert re.search(r"dset = \"([\w\.]*)\"", r.stderr).group(1) == "syn"
        times = [float(m) for m in re.findall(r"time = ([\d\.]*) \(double\)", r.stderr)]
        assert len(times) == 2 and all(0 < t < 1 for t in times)
        m = re.search(r"bad = (\d+) \(int\)\s+bitdiff = (\d+)", r.stderr)
        assert int(m.group(1)) < 30000 * n_len // 10000 + 1
        # SpMMRef (include/spmm_adapter.hpp: the library's exact-order configuration) == the CPU oracle, bit for bit
        assert "ref_vs_oracle_bit_identical = 1 (int)" in r.stderr
        assert os.path.exists(tmp_path / "syn.graph.ptrdump") and os.path.exists(tmp_path / "syn.graph.edgedump")


@pytest.mark.parametrize("overlap", [0, 2])
@pytest.mark.parametrize("split", [0, 1])
def test_run_is_graph_capturable_and_stream_ordered(device, oracle, split, overlap):
    """run() allocates nothing and never synchronises: it can be captured into a HIP
    graph and replayed, and it runs on the caller's stream (the reference: null stream, util.h:133-136)."""
    import torch
    from hpc_amd import CSR, SpMMOpt

    ptr, idx = synth.csr_powerlaw(6000, 24.0, 2000, seed=3)      # hub + segments + rows (split: segments + rows + reduce): three launches
    vals = synth.normal_f32(idx.size, 4)
    B = synth.normal_f32(6000 * 64, 5).reshape(6000, 64)
    d_ptr, d_idx, d_val, d_B = to_dev(device, ptr, idx, vals, B)
    d_C = torch.full((6000, 64), float("nan"), device=device)
    op = SpMMOpt(CSR(6000, idx.size, d_ptr, d_idx, d_val), 64)
    op.set_option("long_row_threshold", 512)
    op.set_option("split_long_rows", split)
    op.set_option("hub_overlap", overlap)      # 2: hub (and, with "segment_overlap", segment) kernels on the handle's side streams, forked and joined inside run()
    op.set_option("segment_overlap", 1 if overlap == 2 else 0)
    op.preprocess(d_B, d_C)
    exp = expected(oracle, ptr, idx, vals, B, split, 512, 256)
    side = torch.cuda.Stream(device=device)
    with torch.cuda.stream(side):
        op.run(d_B, d_C)
    side.synchronize()
    assert np.array_equal(bits(d_C.cpu().numpy()), bits(exp))
    assert op.get_option("n_launches") == 3
    d_C.fill_(float("nan"))
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        op.run(d_B, d_C)
    d_C.fill_(float("nan"))
    g.replay()
    torch.cuda.synchronize()
    assert np.array_equal(bits(d_C.cpu().numpy()), bits(exp))
    # new B contents, same graph
    d_B.mul_(2.0)
    g.replay()
    torch.cuda.synchronize()
    assert np.array_equal(bits(d_C.cpu().numpy()), bits((exp * 2).astype(np.float32)))


def test_every_handle_gets_a_hub_stream_that_overlaps(device, oracle):
    """The hub kernel's side stream is worth its fork only if its kernels run BESIDE the caller's.  Which hardware queue a new stream gets depends
    on how many streams the process made before, and a queue on the caller's pipe is served first, not alongside (round 4: every other handle of
    a process lost its overlap -- youtube-shaped N = 32: 0.32 ms instead of 0.17).  preprocess therefore tests its candidates with a pair of spin
    kernels (mi_spmm_stream_create_concurrent) and keeps one that passes: six coexisting handles, every one of them with an overlapping stream and
    the same step time; and the result is the oracle's whichever stream was picked."""
    import torch
    from hpc_amd import CSR, SpMMOpt

    ptr, idx = synth.csr_powerlaw(200_000, 6.0, 40_000, seed=11, force_max=True)      # one 40 000-nonzero hub beside many short rows: the chain is the step
    M, N = ptr.size - 1, 32
    vals = synth.normal_f32(idx.size, 12)
    B = synth.normal_f32(M * N, 13).reshape(M, N)
    d_ptr, d_idx, d_val, d_B = to_dev(device, ptr, idx, vals, B)
    d_C = torch.empty((M, N), device=device)
    ops, times = [], []
    for i in range(6):
        op = SpMMOpt(CSR(M, idx.size, d_ptr, d_idx, d_val), N)
        op.preprocess(d_B, d_C)
        assert op.get_option("n_hub_rows") >= 1 and op.get_option("side_stream_overlaps") == 1, (i, op.get_option("side_stream_overlaps"))
        for _ in range(3):
            op.run(d_B, d_C)
        batches = []
        for _ in range(3):      # the best of three batches: one host hiccup inside a batch of asynchronous enqueues (a shared box) is not a lost overlap
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            a.record()
            for _ in range(10):
                op.run(d_B, d_C)
            b.record()
            torch.cuda.synchronize()
            batches.append(a.elapsed_time(b) / 10)
        times.append(min(batches))
        ops.append(op)
    assert max(times) < 1.25 * min(times), times
    assert np.array_equal(bits(d_C.cpu().numpy()), bits(oracle.spmm_omp(ptr, idx, vals, B)))
    # the same maker through the C ABI (the multi-GPU step's exchange and re-layout streams come from it)
    import ctypes as C
    from hpc_amd import _lib
    s, ov = C.c_void_p(None), C.c_int(-1)
    assert _lib.load().mi_spmm_stream_create_concurrent(C.byref(s), 1, C.byref(ov)) == 0 and s.value and ov.value == 1


def test_use_graph_option_is_gone(device):
    """Round 4's "use_graph" (the handle replaying its own captured launch set) only ever lost -- on the caller's stream and, round 5, on a tested stream of
    its own (profiles/r05_use_graph_experiment.md) -- and was removed: the key is refused.  A caller's own capture of run() is covered by
    test_run_is_graph_capturable_and_stream_ordered and test_column_strips_inside_a_captured_graph."""
    import torch
    from hpc_amd import CSR, SpMMOpt, MiSpmmError

    ptr = torch.zeros(2, dtype=torch.int32, device=device)
    op = SpMMOpt(CSR(1, 0, ptr, torch.zeros(0, dtype=torch.int32, device=device), torch.zeros(0, dtype=torch.float32, device=device)), 4)
    with pytest.raises(MiSpmmError) as e:
        op.set_option("use_graph", 1)
    assert e.value.code == -5
    for key in ("use_graph", "graph_ready", "graph_replays"):
        with pytest.raises(MiSpmmError):
            op.get_option(key)


def test_negative_zero_accumulators_survive_padded_batches(device, oracle):
    """A chain of negative products that all underflow leaves the accumulator at -0 (fma(-1e-30, 1e-30, +0) is -1e-60 exactly and rounds to -0;
    -0 + -0 stays -0), and the reference kernel stores that -0.  The kernels pad their last batch of a row: the padding must be an identity for
    -0 too -- (+0) x (-0) + acc, not (+0) x (+0) + acc, which turns -0 into +0 (round 4; found by the flush-to-zero test).  Every row length around
    the batch sizes (8 gathers, 32- and 64-pair fetches), every kernel class, N = 32 / 128 / 256 (one, two lane groups per wave, whole-wave rows),
    block-dense groups through the list kernel's masked tail."""
    import torch

    lens = list(range(0, 20)) + [31, 32, 33, 63, 64, 65, 100, 257, 600, 2000, 7, 9] + [11] * 16 + [5] * 16     # rows 32-47 and 48-63: two 16-row groups
    M = len(lens)
    assert M == 64
    K = 1000
    g = np.random.Generator(np.random.Philox(key=[31, 7]))
    ptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    idx = g.integers(0, K, size=int(ptr[-1])).astype(np.int32)
    for r0, L in ((32, 11), (48, 5)):                 # one shared column list per group (lengths 11 and 5: the list kernel's masked last batch)
        cols = g.integers(0, K, size=L).astype(np.int32)
        for r in range(r0, r0 + 16):
            idx[ptr[r]:ptr[r + 1]] = cols
    vals = (-np.abs(g.normal(size=idx.size)) * 1e-30).astype(np.float32)          # every product: negative, ~1e-60: underflows to -0
    for N in (32, 128, 256):
        B = (np.abs(g.normal(size=(K, N))) * 1e-30 + 1e-32).astype(np.float32)
        exp = oracle.spmm_omp(ptr, idx, vals, B)
        nz = np.array(lens) > 0
        assert (bits(exp)[nz] == 0x80000000).all() and (bits(exp)[~nz] == 0).all()      # -0 in every non-empty row, +0 in the empty ones
        if oracle.ref_available():
            d_ptr, d_idx, d_val, d_B = to_dev(device, ptr, idx, vals, B)
            R = torch.full((M, N), float("nan"), device=device)
            oracle.ref_kernel_run(d_ptr, d_idx, d_val, d_B, R, M, N)
            torch.cuda.synchronize()
            assert np.array_equal(bits(R.cpu().numpy()), bits(exp))
        for opts in ({}, {"long_row_threshold": 512, "medium_row_threshold": 16}, {"long_row_threshold": 64, "medium_row_threshold": 8, "hub_slice": 16},
                     {"block_min_len": 4}, {"block_min_len": 4, "long_row_threshold": 256}, {"long_row_threshold": 256, "split_long_rows": 1}):
            C, op = run_spmm(device, ptr, idx, vals, B, options=opts, num_cols=K)
            if "block_min_len" in opts:
                assert op.get_option("n_block_groups") == 2
            bad = np.nonzero((bits(C) != bits(exp)).any(axis=1))[0]
            assert bad.size == 0, (N, opts, [(int(r), lens[r]) for r in bad[:10]])


def test_flush_denormals_matches_the_reference_build(device, oracle):
    """ "flush_denormals" = 1 is the arithmetic of the reference's actual BUILD: nvcc --use_fast_math (W/CMakeLists.txt:46) implies -ftz=true, so
    spmm_kernel_ref's multiply-adds are fma.rn.ftz.f32.  Data full of fp32 subnormals (values, B, and products that land in the subnormal
    range), through every kernel class (short rows, exact segments, hub rows at every slice width, rows of block-dense groups): the product
    equals BOTH the reference kernel compiled with the matching switch (hipcc -fgpu-flush-denormals-to-zero, oracle/_ref) AND the CPU
    restatement of fma.ftz, bit for bit -- and differs from the IEEE result, which the default option still reproduces."""
    import torch
    from hpc_amd import CSR, SpMMOpt

    if not (oracle.ref_available() and oracle.ref_ftz_available()):
        pytest.fail("oracle/_ref missing: run __graft_entry__.build() in the CPU container first")
    g = np.random.Generator(np.random.Philox(key=[909, 1]))
    K = 4000
    lens = [0, 3, 17, 40, 70, 200, 700, 3000] * 6 + [5] * 32         # 80 rows; the last 32 form two 16-row groups below
    ptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    idx = g.integers(0, K, size=int(ptr[-1])).astype(np.int32)
    shared = g.integers(0, K, size=5).astype(np.int32)               # two block-dense groups: 16 rows x one column list of 5 (block_min_len lowered)
    for r in range(48, 80):
        idx[ptr[r]:ptr[r + 1]] = shared
    # every row of A has its own scale (1 ... 1e-36) and every row of B one of 1, 1e-4, 1e-8 (a tenth of them 1e-40: subnormal inputs): a row's products sit together somewhere between
    # 1 and 1e-44, so whole output rows live in or next to the subnormal range instead of being dominated by their largest term
    row_exp = g.choice([0.0, -20.0, -30.0, -36.0], size=len(lens))
    vals = (g.normal(size=idx.size) * np.power(10.0, np.repeat(row_exp, lens))).astype(np.float32)
    for N in (32, 128, 260):
        B = (g.normal(size=(K, N)) * np.power(10.0, g.choice([0.0, -4.0, -8.0, -40.0], size=(K, 1), p=[0.3, 0.3, 0.3, 0.1]))).astype(np.float32)
        assert (np.abs(vals[vals != 0]) < np.finfo(np.float32).tiny).any() and (np.abs(B[B != 0]) < np.finfo(np.float32).tiny).any()
        ieee = oracle.spmm_omp(ptr, idx, vals, B)
        ftz = oracle.spmm_ftz(ptr, idx, vals, B)
        n_differ = int((bits(ieee) != bits(ftz)).sum())
        assert n_differ > ieee.size // 8, "the data must make the two arithmetics differ, or the test shows nothing"
        d_ptr, d_idx, d_val, d_B = to_dev(device, ptr, idx, vals, B)
        M = len(lens)
        R = torch.full((M, N), float("nan"), device=device)
        oracle.ref_kernel_run_ftz(d_ptr, d_idx, d_val, d_B, R, M, N)
        torch.cuda.synchronize()
        assert np.array_equal(bits(R.cpu().numpy()), bits(ftz)), "the CPU restatement of fma.ftz must be what the reference kernel's ftz build computes"
        oracle.ref_kernel_run(d_ptr, d_idx, d_val, d_B, R, M, N)
        torch.cuda.synchronize()
        assert np.array_equal(bits(R.cpu().numpy()), bits(ieee))
        for opts in ({}, {"hub_slice": 16}, {"hub_slice": 64}, {"long_row_threshold": 1 << 30}, {"hub_overlap": 2, "segment_overlap": 1}):
            o = {"long_row_threshold": 512, "medium_row_threshold": 32, "block_min_len": 4, "flush_denormals": 1}
            o.update(opts)
            C, op = run_spmm(device, ptr, idx, vals, B, options=o, num_cols=K)
            assert op.get_option("flush_denormals") == 1 and op.get_option("n_block_groups") == 0      # the f32 MFMA path is not used with it
            assert np.array_equal(bits(C), bits(ftz)), (N, opts, int((bits(C) != bits(ftz)).sum()))
        # the default keeps IEEE subnormals, block path included
        C, op = run_spmm(device, ptr, idx, vals, B, options={"long_row_threshold": 512, "medium_row_threshold": 32, "block_min_len": 4}, num_cols=K)
        assert op.get_option("n_block_groups") == (2 if N % 32 == 0 else 0)
        assert np.array_equal(bits(C), bits(ieee))
        # split mode under ftz: pieces and their left-to-right sum flush as well; checked against the oracle's fma.ftz pieces added with flushing
        # only where nothing subnormal arises is not the point here: just that it runs and keeps the documented tolerance against the ftz chain
        C, op = run_spmm(device, ptr, idx, vals, B, options={"long_row_threshold": 512, "split_long_rows": 1, "flush_denormals": 1}, num_cols=K)
        _, sabs = oracle.spmm_f64(ptr, idx, vals, B)
        assert (np.abs(C.astype(np.float64) - ftz) <= TOL_SPLIT * sabs + 1e-33).all()      # (flushed terms: up to ~1.2e-38 each)


def test_rmat_and_banded_structures(device, oracle):
    """Hub-dominated (R-MAT) and locality-rich (banded) graphs: every row bit-exact by default (hubs through the hub
    kernel); with "split_long_rows" the hubs follow the documented piece order."""
    for name, (ptr, idx) in {"rmat": synth.csr_rmat(14, 16, seed=2), "banded": synth.csr_banded(20000, 4, 40, width=300, seed=2)}.items():
        M = ptr.size - 1
        vals = synth.normal_f32(idx.size, 3)
        B = synth.normal_f32(M * 64, 4).reshape(M, 64)
        plain = oracle.spmm_omp(ptr, idx, vals, B)
        C0, op0 = run_spmm(device, ptr, idx, vals, B, options={"long_row_threshold": 512})
        assert np.array_equal(bits(C0), bits(plain)), name
        C, op = run_spmm(device, ptr, idx, vals, B, options={"long_row_threshold": 512, "split_long_rows": 1})
        exp = oracle.spmm_chunked(ptr, idx, vals, B, 512, 256)
        assert np.array_equal(bits(C), bits(exp)), name
        if name == "rmat":
            assert op.get_option("n_long_rows") > 0 and op0.get_option("n_hub_rows") > 0 and (np.diff(ptr) == 0).any()
            _, sabs = oracle.spmm_f64(ptr, idx, vals, B)
            assert (np.abs(C.astype(np.float64) - plain) <= TOL_SPLIT * sabs + 1e-30).all()


def test_fuzz_shapes_pitches_thresholds(device, oracle):
    """80 seeded random cases: ragged shapes, K != M, odd widths, row pitches wider than N, both rows
    kernels, random medium / hub thresholds, hubs in stored order (default) or split, column strips of the segments off / auto / forced -- always
    bit-equal to the oracle in the documented order."""
    import torch
    from hpc_amd import CSR, SpMMOpt

    n_cases = int(os.environ.get("MI_SPMM_FUZZ_CASES", "80"))          # soak: MI_SPMM_FUZZ_CASES=3000 MI_SPMM_FUZZ_SEED=7
    fuzz_seed = int(os.environ.get("MI_SPMM_FUZZ_SEED", "1"))
    g = np.random.Generator(np.random.Philox(key=[2024, fuzz_seed]))
    for case in range(n_cases):
        M = int(g.integers(1, 400))
        K = int(g.integers(1, 500))
        N = int(g.choice([1, 2, 3, 4, 7, 8, 16, 20, 32, 33, 64, 96, 128, 130, 256, 257, 300, 514, 602]))
        hi = int(g.choice([0, 3, 20, 90, min(K, 400)]))
        ptr, idx = synth.csr_uniform(M, 0, min(hi, K), K=K, seed=1000 * fuzz_seed + case)
        if idx.size and g.random() < 0.5:       # unsorted / duplicated columns inside rows
            idx = g.integers(0, K, size=idx.size).astype(np.int32)
        vals = synth.normal_f32(idx.size, 5000 + case)
        ldb = N + int(g.choice([0, 0, 4, 5, 64]))
        ldc = N + int(g.choice([0, 0, 4, 3, 128]))
        Bp = synth.normal_f32(K * ldb, 9000 + case).reshape(K, ldb)
        opts = {"medium_row_threshold": int(g.choice([0, 1, 5, 64, 1000])),
                "long_row_threshold": int(g.choice([6, 40, 2048])), "long_row_chunk": int(g.choice([3, 16, 256])),
                "block_path": int(g.choice([0, 1])), "segment_unroll": int(g.choice([8, 16, 32])), "split_long_rows": int(case % 3 == 2),
                "hub_slice": (0, 16, 32, 64)[case % 4], "hub_overlap": (1, 2, 0)[case % 3], "segment_overlap": (case // 3) % 2,
                "col_strips": (0, 2, 1, 3, 7)[case % 5]}      # forced strip counts take effect where the segments' columns ascend and no row is split
        d_ptr, d_idx, d_val, d_B = to_dev(device, ptr, idx, vals, Bp)
        d_C = torch.full((M, ldc), float("nan"), dtype=torch.float32, device=device)
        op = SpMMOpt(CSR(M, idx.size, d_ptr, d_idx, d_val), N, num_cols=K)
        for k, v in opts.items():
            op.set_option(k, v)
        op.preprocess(d_B, d_C)
        if g.random() < 0.3 and M > 2:          # the same C from a ragged sequence of row-range calls
            cuts = sorted(set([0, M] + [int(x) for x in g.integers(0, M, 3)]))
            for r0, r1 in zip(cuts, cuts[1:]):
                op.run_rows(d_B, ldb, d_C, ldc, r0, r1)
        else:
            op.run_ld(d_B, ldb, d_C, ldc)
        torch.cuda.synchronize()
        got = d_C.cpu().numpy()
        exp = expected(oracle, ptr, idx, vals, np.ascontiguousarray(Bp[:, :N]), opts["split_long_rows"], opts["long_row_threshold"], opts["long_row_chunk"])
        assert np.array_equal(bits(got[:, :N]), bits(exp)), (case, M, K, N, ldb, ldc, opts)
        assert np.isnan(got[:, N:]).all(), "wrote outside its N columns"


def test_fuzz_block_items(device, oracle):
    """Seeded random block structures through the item machinery: 1-4 runs per group of arbitrary start / length (aligned or
    not to the k batches), plain lists, groups that miss qualifying by one column, every slab width, row pitches, random
    option combinations, ragged row-range calls -- always the oracle's bits."""
    import torch
    from hpc_amd import CSR, SpMMOpt

    n_cases = int(os.environ.get("MI_SPMM_BLOCK_FUZZ_CASES", "40"))       # soak: MI_SPMM_BLOCK_FUZZ_CASES=400
    seen = {"groups": 0, "shared": 0, "multi_pass": 0, "panels": 0}
    g = np.random.Generator(np.random.Philox(key=[4040, int(os.environ.get("MI_SPMM_FUZZ_SEED", "1"))]))
    for case in range(n_cases):
        K = int(g.integers(1200, 2600))
        N = int(g.choice([32, 64, 96, 128, 160, 256, 288, 384, 512]))
        n_groups = int(g.integers(3, 40))
        starts_pool = np.sort(g.choice((K - 140) // 160, size=int(g.integers(2, 6)), replace=False)) * 160   # far enough apart never to collide
        ptr, idx = [0], []
        for b in range(n_groups):
            kind = int(g.integers(0, 5))
            if kind <= 2:                                     # runs from a small pool of starts (sharing), lengths of every residue
                nr = int(g.integers(1, 5))
                st = np.sort(g.choice(starts_pool, size=min(nr, starts_pool.size), replace=False))
                cols, last = [], -1
                for q in st:
                    L = int(g.choice([32, 64, 96, 128, 32, 64, 16, 33, 40, 48, 7, 100]))
                    q = max(int(q), last + 2)                 # keep the runs apart (a gap of at least one column)
                    if q + L > K:
                        continue
                    cols.append(np.arange(q, q + L))
                    last = q + L
                cols = np.concatenate(cols).astype(np.int32) if cols else g.integers(0, K, size=9).astype(np.int32)
            elif kind == 3:
                cols = g.integers(0, K, size=int(g.integers(8, 120))).astype(np.int32)
            else:
                cols = None
            for r in range(16):
                if cols is None:
                    c = np.sort(g.choice(K, int(g.integers(0, 30)), replace=False)).astype(np.int32)
                else:
                    c = cols.copy()
                    if kind == 2 and b % 5 == 0 and r == 9:
                        c[c.size // 2] = (c[c.size // 2] + 1) % K      # one row differs: the group must NOT qualify
                idx.append(c)
                ptr.append(ptr[-1] + c.size)
        for _ in range(int(g.integers(0, 9))):                # M not a multiple of 16
            c = np.sort(g.choice(K, int(g.integers(1, 20)), replace=False)).astype(np.int32)
            idx.append(c)
            ptr.append(ptr[-1] + c.size)
        ptr = np.asarray(ptr, np.int32)
        idx = np.concatenate(idx).astype(np.int32)
        vals = synth.normal_f32(idx.size, 7000 + case)
        M = ptr.size - 1
        ldb = N + int(g.choice([0, 0, 4, 64]))
        ldc = N + int(g.choice([0, 0, 4, 128]))
        Bp = synth.normal_f32(K * ldb, 8000 + case).reshape(K, ldb)
        opts = {"split_long_rows": case % 2, "long_row_threshold": int(g.choice([256, 2048])), "block_share": int(g.choice([1, 2])), "block_max_pieces": int(g.choice([1, 2, 4])),
                "block_run_min": int(g.choice([8, 32, 64])), "block_min_len": int(g.choice([8, 30]))}
        d_ptr, d_idx, d_val, d_B = to_dev(device, ptr, idx, vals, Bp)
        d_C = torch.full((M, ldc), float("nan"), dtype=torch.float32, device=device)
        op = SpMMOpt(CSR(M, idx.size, d_ptr, d_idx, d_val), N, num_cols=K)
        for k, v in opts.items():
            op.set_option(k, v)
        op.preprocess(d_B, d_C)
        if g.random() < 0.4 and M > 2:
            cuts = sorted(set([0, M] + [int(x) for x in g.integers(0, M, 3)]))
            for r0, r1 in zip(cuts, cuts[1:]):
                op.run_rows(d_B, ldb, d_C, ldc, r0, r1)
        else:
            op.run_ld(d_B, ldb, d_C, ldc)
            op.run_ld(d_B, ldb, d_C, ldc)              # a second step over the carried tiles of the first
        torch.cuda.synchronize()
        got = d_C.cpu().numpy()
        exp = expected(oracle, ptr, idx, vals, np.ascontiguousarray(Bp[:, :N]), opts["split_long_rows"], opts["long_row_threshold"], 256)
        assert np.array_equal(bits(got[:, :N]), bits(exp)), (case, M, K, N, ldb, ldc, opts, op.get_option("n_block_groups"), op.get_option("n_block_passes"))
        assert np.isnan(got[:, N:]).all(), "wrote outside its N columns"
        seen["groups"] += op.get_option("n_block_groups")
        seen["shared"] += op.get_option("n_block_shared_items")
        seen["multi_pass"] += 1 if op.get_option("n_block_passes") > 1 else 0
    assert seen["groups"] > 5 * n_cases and seen["shared"] > n_cases // 8 and seen["multi_pass"] > n_cases // 4, seen


def test_special_values_all_paths(device, oracle):
    """inf, NaN, -0.0 and subnormals in A and B through the rows, segment (medium), hub (and split) and MFMA block
    paths: same bits as the oracle (NaNs compared as NaN: payloads are not part of the contract).
    Guards the kernels' padding trick (empty slots are fma(+0, +0, acc)) and the f32 MFMA's subnormal handling."""
    g = np.random.Generator(np.random.Philox(key=[77, 7]))
    ptr, idx, vals, B, kinds = _shared_list_case(24, 1500, 128, seed=4242)      # block groups + ragged rows
    # add some long rows at the end
    extra = [np.sort(g.choice(1500, d, replace=False)).astype(np.int32) for d in (700, 90, 300)]
    idx = np.concatenate([idx] + extra)
    ptr = np.concatenate([ptr, ptr[-1] + np.cumsum([e.size for e in extra])]).astype(np.int32)
    vals = synth.normal_f32(idx.size, 1)
    B = synth.normal_f32(1500 * 128, 2).reshape(1500, 128)
    special = np.array([np.inf, -np.inf, np.nan, -0.0, 0.0, 1e-40, -3e-42, 1.1754942e-38, 3.4e38, -3.4e38], np.float32)
    vals[g.integers(0, vals.size, 400)] = special[g.integers(0, special.size, 400)]
    B.reshape(-1)[g.integers(0, B.size, 4000)] = special[g.integers(0, special.size, 4000)]
    for opts in ({}, {"block_path": 0}, {"rows_per_block": 1000}, {"split_long_rows": 1}, {"hub_slice": 16}, {"hub_slice": 64}):
        ref = expected(oracle, ptr, idx, vals, B, opts.get("split_long_rows", 0), 256, 64)
        o = {"long_row_threshold": 256, "long_row_chunk": 64, "medium_row_threshold": 32}
        o.update(opts)
        C, op = run_spmm(device, ptr, idx, vals, B, options=o)
        both_nan = np.isnan(C) & np.isnan(ref)
        same = (bits(C) == bits(ref)) | both_nan
        assert same.all(), (opts, int((~same).sum()))
        assert np.isnan(ref).any() and np.isinf(ref).any()
        if "block_path" not in opts:
            assert op.get_option("n_block_groups") > 0 and op.get_option("n_long_rows") > 0 and op.get_option("n_medium_rows") > 0


def test_gpu_and_host_plan_builders_agree(device, oracle):
    """the segment table and the block items built on the device (classify + scans + emit + radix sorts) against
    the reference-style host loops: same counts, same launches, same results, on hub-heavy, block and mixed structures."""
    cases = {
        "rmat": synth.csr_rmat(15, 24, seed=4),
        "powerlaw": synth.csr_powerlaw(30000, 40.0, 3000, seed=4),
        "uniform": synth.csr_uniform(5000, 0, 50, seed=4),
        "blocks": _shared_list_case(50, 2500, 128, seed=44)[:2],
        "block_runs": _run_groups_case(300, 5000, 128, seed=45, lens=(32, 48, 64, 96, 128, 33, 208, 16), slots=20)[:2],   # 1-3 passes, shared items, list pieces
        "single_row": (np.array([0, 5000], np.int32), np.arange(5000, dtype=np.int32)),
    }
    for name, (ptr, idx) in cases.items():
        M = ptr.size - 1
        K = max(M, int(idx.max()) + 1 if idx.size else 1)
        vals = synth.normal_f32(idx.size, 8)
        B = synth.normal_f32(K * 128, 9).reshape(K, 128)
        for thr, split in ((0, 0), (100, 0), (0, 1), (100, 1)):    # auto and a low explicit threshold; hubs whole or in pieces
            res = {}
            for gpu_pre in (1, 0):
                # ("col_strips" off: its auto rule looks at the column-locality sample, which only the device builder takes;
                #  forced strip counts through both builders: test_column_strips_keep_every_bit)
                C, op = run_spmm(device, ptr, idx, vals, B, options={"gpu_preprocess": gpu_pre, "long_row_threshold": thr,
                                                                    "long_row_chunk": 64, "split_long_rows": split, "col_strips": 1})
                res[gpu_pre] = (C, {k: op.get_option(k) for k in ("n_chunks", "n_long_rows", "n_hub_rows", "n_medium_rows", "n_partial_slots",
                                                                  "n_block_groups", "max_row_nnz", "long_row_threshold", "n_block_items",
                                                                  "n_block_pieces", "n_block_passes", "n_block_shared_items", "n_launches")})
            if name == "block_runs":
                assert res[1][1]["n_block_passes"] >= 2 and res[1][1]["n_block_shared_items"] > 0 and res[1][1]["n_block_items"] > 100
            assert res[1][1] == res[0][1], (name, thr, split, res[1][1], res[0][1])
            assert np.array_equal(bits(res[1][0]), bits(res[0][0])), (name, thr, split)
            t = res[1][1]["long_row_threshold"]
            assert np.array_equal(bits(res[1][0]), bits(expected(oracle, ptr, idx, vals, B, split, t, 64))), (name, thr, split)


def test_unaligned_pointers_fall_back_cleanly(device, oracle):
    """B/C that are only 4-byte aligned cannot use the MFMA block path (it keeps the strict 16-byte rule): the rows
    kernel -- still 16 bytes per lane, dword-aligned -- takes over the rows of detected block groups, identical bits."""
    import torch
    from hpc_amd import CSR, SpMMOpt

    ptr, idx, vals, B, kinds = _shared_list_case(30, 1200, 128, seed=321)
    extra = np.sort(np.random.Generator(np.random.Philox(key=[5, 5])).choice(1200, 300, replace=False)).astype(np.int32)
    idx = np.concatenate([idx, extra])
    ptr = np.concatenate([ptr, [ptr[-1] + extra.size]]).astype(np.int32)       # one medium row as well
    vals = synth.normal_f32(idx.size, 1)
    M = ptr.size - 1
    d_ptr, d_idx, d_val = to_dev(device, ptr, idx, vals)
    Bbuf = torch.zeros(1200 * 128 + 1, device=device)
    Bbuf[1:] = torch.from_numpy(B.reshape(-1)).to(device)
    Cbuf = torch.full((M * 128 + 1,), float("nan"), device=device)
    op = SpMMOpt(CSR(M, idx.size, d_ptr, d_idx, d_val), 128, num_cols=1200)
    op.set_option("long_row_threshold", 2048)
    op.preprocess(Bbuf[1:], Cbuf[1:])
    assert op.get_option("n_block_groups") > 0 and op.get_option("n_medium_rows") > 0
    op.run(Bbuf[1:], Cbuf[1:])
    torch.cuda.synchronize()
    assert op.get_option("vector_width") == 4
    got = Cbuf[1:].view(M, 128).cpu().numpy()
    assert np.array_equal(bits(got), bits(oracle.spmm_omp(ptr, idx, vals, B)))
    assert torch.isnan(Cbuf[0])
    # auto hub threshold (256 here) BELOW a block group's list length (1100): the group still belongs to the block path
    # (detection goes up to the auto rule's largest candidate), and on an unaligned call the rows kernel must take its rows
    # although they are longer than every threshold.  20 000 short rows keep the group's share of the nonzeros small.
    g = np.random.Generator(np.random.Philox(key=[6, 6]))
    cols = np.sort(g.choice(1200, 1100, replace=False)).astype(np.int32)
    bp, bi = synth.csr_uniform(20000, 10, 30, K=1200, seed=77)
    pad = (-(ptr.size - 1 + 20000)) % 16             # empty rows so that the appended group starts on a 16-row boundary
    ptr2 = np.concatenate([ptr, ptr[-1] + bp[1:], np.full(pad, ptr[-1] + bp[-1]), ptr[-1] + bp[-1] + 1100 * np.arange(1, 17)]).astype(np.int32)
    idx2 = np.concatenate([idx, bi] + [cols] * 16).astype(np.int32)
    assert (ptr2.size - 1) % 16 == 0 and ptr2[-1] == idx2.size
    vals2 = synth.normal_f32(idx2.size, 3)
    M2 = ptr2.size - 1
    d_ptr2, d_idx2, d_val2 = to_dev(device, ptr2, idx2, vals2)
    Cbuf2 = torch.full((M2 * 128 + 1,), float("nan"), device=device)
    op2 = SpMMOpt(CSR(M2, idx2.size, d_ptr2, d_idx2, d_val2), 128, num_cols=1200)
    op2.preprocess(Bbuf[1:], Cbuf2[1:])
    assert op2.get_option("long_row_threshold") == 256 and op2.get_option("n_block_groups") > 0
    exp2 = oracle.spmm_omp(ptr2, idx2, vals2, B)
    op2.run(Bbuf[1:], Cbuf2[1:])                     # unaligned: fallback
    torch.cuda.synchronize()
    assert np.array_equal(bits(Cbuf2[1:].view(M2, 128).cpu().numpy()), bits(exp2))
    d_B2 = torch.from_numpy(B).to(device)
    d_C2 = torch.full((M2, 128), float("nan"), device=device)
    op2.run(d_B2, d_C2)                              # aligned: the MFMA path takes the 500-long group
    torch.cuda.synchronize()
    assert np.array_equal(bits(d_C2.cpu().numpy()), bits(exp2))


def test_wide_addressing_variants(device, oracle):
    """The narrow kernels use a 32-bit byte offset per gathered B row (host-checked: K <= 2^24, pitch < 16 MiB,
    B <= 4 GiB).  A row pitch of 4 Mi floats forces the 64-bit ("wide") variants of every kernel."""
    import torch
    from hpc_amd import CSR, SpMMOpt

    K, N, ldb = 96, 128, (1 << 22) + 8
    ptr, idx, vals, Bs, kinds = _shared_list_case(20, K, N, seed=808)              # block groups + ragged rows
    g = np.random.Generator(np.random.Philox(key=[8, 8]))
    extra = [g.integers(0, K, size=d).astype(np.int32) for d in (700, 150)]          # a hub row and a medium row
    idx = np.concatenate([idx] + extra)
    ptr = np.concatenate([ptr, ptr[-1] + np.cumsum([e.size for e in extra])]).astype(np.int32)
    vals = synth.normal_f32(idx.size, 2)
    M = ptr.size - 1
    d_ptr, d_idx, d_val = to_dev(device, ptr, idx, vals)
    d_B = torch.zeros(K * ldb, device=device)                                          # 1.6 GB, only N columns per row used
    d_B.view(K, ldb)[:, :N] = torch.from_numpy(Bs).to(device)
    d_C = torch.full((M, N), float("nan"), device=device)
    op = SpMMOpt(CSR(M, idx.size, d_ptr, d_idx, d_val), N, num_cols=K)
    op.set_option("long_row_threshold", 256)
    op.set_option("medium_row_threshold", 64)
    op.preprocess(d_B, d_C)
    assert op.get_option("n_block_groups") > 0 and op.get_option("n_long_rows") == 1 and op.get_option("n_medium_rows") >= 1
    op.run_ld(d_B, ldb, d_C, N)
    torch.cuda.synchronize()
    assert op.get_option("wide_addressing") == 1 and op.get_option("vector_width") == 4
    exp = oracle.spmm_omp(ptr, idx, vals, Bs)
    assert op.get_option("n_hub_rows") == 1
    assert np.array_equal(bits(d_C.cpu().numpy()), bits(exp))
    for rpb in (0, 1000):                     # long lane-group runs through the wide variants too
        op.set_option("rows_per_block", rpb)
        d_C.fill_(float("nan"))
        op.run_ld(d_B, ldb, d_C, N)
        torch.cuda.synchronize()
        assert np.array_equal(bits(d_C.cpu().numpy()), bits(exp)), rpb
    del d_B, op
    # the run kernels of the block path (shared items, passes) through the 64-bit-address variants as well
    K2 = 210
    ptr, idx, vals, Bs = _run_groups_case(60, K2, N, seed=909, lens=(32, 64), slots=3)
    M = ptr.size - 1
    d_ptr, d_idx, d_val = to_dev(device, ptr, idx, vals)
    d_B = torch.zeros(K2 * ldb, device=device)                                         # 3.5 GB
    d_B.view(K2, ldb)[:, :N] = torch.from_numpy(Bs).to(device)
    d_C = torch.full((M, N), float("nan"), device=device)
    op = SpMMOpt(CSR(M, idx.size, d_ptr, d_idx, d_val), N, num_cols=K2)
    op.preprocess(d_B, d_C)
    assert op.get_option("n_block_shared_items") > 0 and op.get_option("n_block_passes") >= 2
    op.run_ld(d_B, ldb, d_C, N)
    torch.cuda.synchronize()
    assert op.get_option("wide_addressing") == 1
    assert np.array_equal(bits(d_C.cpu().numpy()), bits(oracle.spmm_omp(ptr, idx, vals, Bs)))
    del d_B


# ---- column strips of the exact segments (plan.hpp resolve_col_strips) ----
def _strip_case(M, K, lo, hi, seed, hubs=()):
    ptr, idx = synth.csr_uniform(M, lo, hi, K=K, seed=seed)          # columns ascending and distinct inside a row
    if hubs:
        g = np.random.Generator(np.random.Philox(key=[77, seed]))
        deg = np.diff(ptr).astype(np.int64)
        rows = g.choice(M, size=len(hubs), replace=False)
        for r, L in zip(rows, hubs):
            deg[r] = L
        new_ptr = np.zeros(M + 1, np.int64)
        np.cumsum(deg, out=new_ptr[1:])
        new_idx = np.empty(int(new_ptr[-1]), np.int32)
        for r in range(M):
            if deg[r] == ptr[r + 1] - ptr[r]:
                new_idx[new_ptr[r]:new_ptr[r + 1]] = idx[ptr[r]:ptr[r + 1]]
            else:
                new_idx[new_ptr[r]:new_ptr[r + 1]] = np.sort(g.choice(K, int(deg[r]), replace=False)).astype(np.int32)
        ptr, idx = new_ptr.astype(np.int32), new_idx
    return ptr, idx


@pytest.mark.parametrize("N", [3, 4, 32, 100, 128, 256, 300])
def test_column_strips_keep_every_bit(device, oracle, N):
    """"col_strips" = S cuts every exact segment at S - 1 column boundaries and runs strip after strip, each continuing the rows' fma
    chains through C: scheduling only.  Forced strip counts (also more strips than a short row has nonzeros: empty sub-segments), short,
    medium and hub rows side by side, pitched B / C, ragged row-range calls, both plan builders: always the oracle's bits."""
    import torch
    from hpc_amd import CSR, SpMMOpt

    M, K = 1500, 2300
    ptr, idx = _strip_case(M, K, 0, 260, seed=31 + N, hubs=(900, 1500, 2299))
    vals = synth.normal_f32(idx.size, 32)
    ldb, ldc = N + (0 if N % 8 else 4), N + (3 if N == 100 else 0)
    Bp = synth.normal_f32(K * ldb, 33).reshape(K, ldb)
    exp = oracle.spmm_omp(ptr, idx, vals, np.ascontiguousarray(Bp[:, :N]))
    d_ptr, d_idx, d_val, d_B = to_dev(device, ptr, idx, vals, Bp)
    for S, gpu_pre, ranges in ((2, 1, False), (3, 1, True), (5, 0, False), (8, 1, True), (33, 1, False), (1, 1, False), (0, 1, False)):
        d_C = torch.full((M, ldc), float("nan"), dtype=torch.float32, device=device)
        op = SpMMOpt(CSR(M, idx.size, d_ptr, d_idx, d_val), N, num_cols=K)
        for k, v in {"col_strips": S, "medium_row_threshold": 24, "long_row_threshold": 512, "gpu_preprocess": gpu_pre}.items():
            op.set_option(k, v)
        op.preprocess(d_B, d_C)
        assert op.get_option("n_medium_rows") > 1000
        assert op.get_option("n_col_strips") == (S if S >= 2 else 1), (S, op.get_option("n_col_strips"))   # auto (0): far too small a B to strip
        if S != 1 and S != 0:
            assert op.get_option("segments_unsorted") == 0
        for rep in range(2):                           # idempotent: strip 0 starts every chain from +0 again
            if ranges:
                for r0, r1 in ((0, 1), (1, 700), (700, 701), (701, M)):
                    op.run_rows(d_B, ldb, d_C, ldc, r0, r1)
            else:
                op.run_ld(d_B, ldb, d_C, ldc)
        torch.cuda.synchronize()
        got = d_C.cpu().numpy()
        assert np.array_equal(bits(got[:, :N]), bits(exp)), (S, gpu_pre, int((bits(got[:, :N]) != bits(exp)).sum()))
        assert np.isnan(got[:, N:]).all()


def test_column_strips_need_ascending_columns_and_survive_special_values(device, oracle):
    """A row whose columns do not ascend cannot be cut by column without changing its order: one such segment switches the strips off
    for the plan ("segments_unsorted" says how many).  Equal neighbours (duplicate columns) are fine.  inf, NaN, -0 and subnormals
    cross strip boundaries unchanged: a strip hands the chain on as the f32 it is."""
    M, K, N = 600, 900, 64
    ptr, idx = _strip_case(M, K, 40, 300, seed=5)
    vals = synth.normal_f32(idx.size, 6)
    B = synth.normal_f32(K * N, 7).reshape(K, N)
    opts = {"col_strips": 4, "medium_row_threshold": 16, "long_row_threshold": 4096}
    # (1) sorted with duplicates
    idx_dup = idx.copy()
    for r in range(0, M, 7):
        b, e = ptr[r], ptr[r + 1]
        if e - b > 3:
            idx_dup[b + 1] = idx_dup[b]
            idx_dup[e - 1] = idx_dup[e - 2]
    C, op = run_spmm(device, ptr, idx_dup, vals, B, options=opts)
    assert op.get_option("n_col_strips") == 4 and op.get_option("segments_unsorted") == 0
    assert np.array_equal(bits(C), bits(oracle.spmm_omp(ptr, idx_dup, vals, B)))
    # (2) one unsorted segment
    idx_bad = idx.copy()
    b = ptr[300]
    idx_bad[b], idx_bad[b + 5] = idx_bad[b + 5], idx_bad[b]
    C, op = run_spmm(device, ptr, idx_bad, vals, B, options=opts)
    assert op.get_option("n_col_strips") == 1 and op.get_option("segments_unsorted") == 1
    assert np.array_equal(bits(C), bits(oracle.spmm_omp(ptr, idx_bad, vals, B)))
    # (3) special values on both sides of the strip boundaries
    v = vals.copy()
    Bs = B.copy()
    g = np.random.Generator(np.random.Philox(key=[9, 9]))
    v[g.integers(0, v.size, 40)] = np.float32(np.inf)
    v[g.integers(0, v.size, 40)] = np.float32(-0.0)
    v[g.integers(0, v.size, 200)] = np.float32(1e-30)
    Bs[g.integers(0, K, 30), g.integers(0, N, 30)] = np.float32(np.nan)
    Bs[g.integers(0, K, 200), :] *= np.float32(-1e-12)
    Bs[g.integers(0, K, 50), :] = np.float32(-1e-30)
    for ftz in (0, 1):
        C, op = run_spmm(device, ptr, idx, v, Bs, options=dict(opts, flush_denormals=ftz))
        assert op.get_option("n_col_strips") == 4
        ref = oracle.spmm_ftz(ptr, idx, v, Bs) if ftz else oracle.spmm_omp(ptr, idx, v, Bs)
        assert np.array_equal(bits(C), bits(ref)), ftz


def test_column_strips_auto_rule_and_extra_destinations(device, oracle):
    """The rule picks strips on a graph of long rows over few columns (B = 16 MiB at N = 64: three strips of 5.3 MiB) and leaves C1-like
    graphs alone; with extra destinations (the multi-GPU peer_store exchange) only the LAST strip stores into them, and it does so for
    every row -- also rows with no nonzero in the last strip."""
    import ctypes

    import torch
    from hpc_amd import CSR, SpMMOpt, _lib

    M = K = 65536
    N = 64
    ptr, idx = synth.csr_uniform(M, 150, 350, K=K, seed=12)
    # a quarter of the rows only use the first half of the columns: nothing of them in the last strip
    for r in range(0, M, 4):
        seg = idx[ptr[r]:ptr[r + 1]]
        idx[ptr[r]:ptr[r + 1]] = np.sort(seg // 2)
    vals = synth.normal_f32(idx.size, 13)
    B = synth.normal_f32(K * N, 14).reshape(K, N)
    exp = oracle.spmm_omp(ptr, idx, vals, B)
    d_ptr, d_idx, d_val, d_B = to_dev(device, ptr, idx, vals, B)
    d_C = torch.full((M, N), float("nan"), dtype=torch.float32, device=device)
    op = SpMMOpt(CSR(M, idx.size, d_ptr, d_idx, d_val), N)
    op.preprocess(d_B, d_C)
    assert op.get_option("n_col_strips") == 3, op.get_option("n_col_strips")
    op.run(d_B, d_C)
    torch.cuda.synchronize()
    assert np.array_equal(bits(d_C.cpu().numpy()), bits(exp))
    # extra destinations
    extra = [torch.full((M, N), float("nan"), dtype=torch.float32, device=device) for _ in range(2)]
    arr = (ctypes.c_void_p * 2)(*[ctypes.c_void_p(t.data_ptr()) for t in extra])
    d_C.fill_(float("nan"))
    rc = _lib.load().mi_spmm_run_rows_multi(op._h, ctypes.c_void_p(d_B.data_ptr()), N, ctypes.c_void_p(d_C.data_ptr()), N, 0, M, 2, arr, None)
    assert rc == 0
    torch.cuda.synchronize()
    for t in [d_C] + extra:
        assert np.array_equal(bits(t.cpu().numpy()), bits(exp))
    # C1-like: short rows, B far beyond the caches -> no strips, no survey
    p1, i1 = synth.csr_uniform(20000, 16, 48, seed=3)
    C1, op1 = run_spmm(device, p1, i1, synth.normal_f32(i1.size, 1), synth.normal_f32(20000 * 32, 2).reshape(20000, 32))
    assert op1.get_option("n_col_strips") == 1 and op1.get_option("segments_unsorted") == -1


def test_column_strips_fold_hubs_whose_chains_hide_inside_the_strips(device, oracle):
    """Where strips are in force and the hub threshold is the library's to choose, rows above it become stripped segments too when the longest row's
    sub-chains hide inside the strips' launches (mi_spmm.hip preprocess_on_gpu: 100 ns per nonzero of the longest row against half of the stripped
    step's estimate); a row too long for that keeps the hubs, and so does an explicit threshold.  Scheduling only: always the oracle's bits."""
    M = K = 65536
    N = 128
    vals_seed = 21
    for hubs, thr_opt, folded in (((3000, 3000, 2500), 0, True), ((3000, 3000, 2500), 2048, False), ((3000, 60000), 0, False)):
        ptr, idx = _strip_case(M, K, 150, 350, seed=17, hubs=hubs)
        vals = synth.normal_f32(idx.size, vals_seed)
        B = synth.normal_f32(K * N, 22).reshape(K, N)
        C, op = run_spmm(device, ptr, idx, vals, B, options={"long_row_threshold": thr_opt})
        assert op.get_option("n_col_strips") >= 2 and op.get_option("segments_unsorted") == 0
        if folded:
            assert op.get_option("n_hub_rows") == 0 and op.get_option("long_row_threshold") == 1 << 30
        else:
            assert op.get_option("n_hub_rows") == len(hubs) if thr_opt else op.get_option("n_hub_rows") >= 1
            assert op.get_option("long_row_threshold") < 1 << 30
        assert np.array_equal(bits(C), bits(oracle.spmm_omp(ptr, idx, vals, B))), (hubs, thr_opt)


@pytest.mark.parametrize("overlap", [0, 2])
def test_column_strips_inside_a_captured_graph(device, oracle, overlap):
    """The strip launches are ordinary stream-ordered launches: captured into a caller's own HIP graph they replay to the same bits, with the segment
    kernel on the caller's stream or on its side stream."""
    import torch
    from hpc_amd import CSR, SpMMOpt

    M, K, N = 3000, 4000, 64
    ptr, idx = _strip_case(M, K, 60, 300, seed=41, hubs=(1500, 3999))
    vals = synth.normal_f32(idx.size, 42)
    B = synth.normal_f32(K * N, 43).reshape(K, N)
    exp = oracle.spmm_omp(ptr, idx, vals, B)
    d_ptr, d_idx, d_val, d_B = to_dev(device, ptr, idx, vals, B)
    d_C = torch.full((M, N), float("nan"), device=device)
    op = SpMMOpt(CSR(M, idx.size, d_ptr, d_idx, d_val), N, num_cols=K)
    for k, v in {"col_strips": 4, "long_row_threshold": 1024, "medium_row_threshold": 32, "hub_overlap": overlap, "segment_overlap": 1 if overlap == 2 else 0}.items():
        op.set_option(k, v)
    op.preprocess(d_B, d_C)
    assert op.get_option("n_col_strips") == 4 and op.get_option("n_hub_rows") == 2
    op.run(d_B, d_C)
    torch.cuda.synchronize()
    assert np.array_equal(bits(d_C.cpu().numpy()), bits(exp)) and op.get_option("n_launches") == 5        # hub kernel + 4 strips (no row is short enough for the rows kernel)
    s = torch.cuda.Stream(device=device)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(s):
        op.run(d_B, d_C)                      # warm-up outside the capture
        s.synchronize()
        with torch.cuda.graph(g, stream=s):
            op.run(d_B, d_C)
    for rep in range(2):
        d_C.fill_(float("nan"))
        torch.cuda.synchronize()
        g.replay()
        torch.cuda.synchronize()
        assert np.array_equal(bits(d_C.cpu().numpy()), bits(exp))


# ---- round 5 -----------------------------------------------------------------------------------------------------------------------
def test_run_on_a_non_blocking_stream_straight_after_preprocess(device, oracle):
    """mi_spmm.h: preprocess synchronises.  Its last launches (the strip tables, round 4: build_col_strips on the null stream, nothing behind it when the
    hubs were folded and no side stream was made) used to be still in flight when it returned; a run() on a hipStreamNonBlocking stream -- every
    torch.cuda.Stream(), the multi-GPU step's streams -- is not ordered behind the null stream and could read a half-written table.
    No synchronize between preprocess and run here, many handles in a row so that a race would have its chances."""
    import torch
    from hpc_amd import CSR, SpMMOpt

    M = K = 30000
    N = 64
    ptr, idx = _strip_case(M, K, 150, 350, seed=51)
    vals = synth.normal_f32(idx.size, 52)
    B = synth.normal_f32(K * N, 53).reshape(K, N)
    exp = oracle.spmm_omp(ptr, idx, vals, B)
    d_ptr, d_idx, d_val, d_B = to_dev(device, ptr, idx, vals, B)
    s = torch.cuda.Stream(device=device)          # hipStreamNonBlocking
    torch.cuda.synchronize()
    for builder in (0, 1, 0, 1, 0, 0):
        d_C = torch.full((M, N), float("nan"), dtype=torch.float32, device=device)
        torch.cuda.synchronize()
        op = SpMMOpt(CSR(M, idx.size, d_ptr, d_idx, d_val), N)
        for k, v in {"col_strips": 16, "hub_overlap": 0, "segment_overlap": 0, "col_strips_builder": builder}.items():
            op.set_option(k, v)
        op.preprocess(d_B, d_C)
        with torch.cuda.stream(s):
            op.run(d_B, d_C)
        s.synchronize()
        assert op.get_option("n_col_strips") == 16
        assert np.array_equal(bits(d_C.cpu().numpy()), bits(exp)), builder


def test_strip_builders_agree(device, oracle):
    """Round 5's one-pass strip builder (strip_segments: a wave strides a segment's columns once; a lane whose column and its left neighbour's lie in
    different strips has found where the strips in between begin) writes the tables of round 4's survey + S binary searches per segment, record for
    record: same FNV hash over the table, for several strip counts, duplicate columns, K not a multiple of S, segments shorter than S, both plan builders."""
    M, K, N = 5000, 7001, 32
    ptr, idx = _strip_case(M, K, 0, 200, seed=61, hubs=(900, 3000, 7001))
    for r in range(0, M, 5):                       # equal neighbours
        b, e = ptr[r], ptr[r + 1]
        if e - b > 4:
            idx[b + 2] = idx[b + 1]
            idx[e - 1] = idx[e - 2]
    vals = synth.normal_f32(idx.size, 62)
    B = synth.normal_f32(K * N, 63).reshape(K, N)
    exp = oracle.spmm_omp(ptr, idx, vals, B)
    for S in (2, 3, 7, 13, 32, 64):
        for gpu_pre in (1, 0):
            h = []
            for builder in (0, 1):
                C, op = run_spmm(device, ptr, idx, vals, B, options={"col_strips": S, "medium_row_threshold": 8, "long_row_threshold": 1 << 30,
                                                                     "gpu_preprocess": gpu_pre, "col_strips_builder": builder})
                assert op.get_option("n_col_strips") == S and op.get_option("segments_unsorted") == 0
                assert op.get_option("segment_nnz") == int(np.diff(ptr)[np.diff(ptr) > 8].sum())
                assert np.array_equal(bits(C), bits(exp)), (S, gpu_pre, builder)
                h.append(op.get_option("col_strips_table_hash"))
            assert h[0] == h[1] and h[0] != 0, (S, gpu_pre, h)
    # an unsorted segment: both builders say so and neither table is used
    idx_bad = idx.copy()
    r = int(np.argmax(np.diff(ptr)))
    idx_bad[ptr[r] + 10], idx_bad[ptr[r] + 400] = idx_bad[ptr[r] + 400], idx_bad[ptr[r] + 10]
    for builder in (0, 1):
        C, op = run_spmm(device, ptr, idx_bad, vals, B, options={"col_strips": 7, "medium_row_threshold": 8, "long_row_threshold": 1 << 30, "col_strips_builder": builder})
        assert op.get_option("n_col_strips") == 1 and op.get_option("segments_unsorted") == 1 and op.get_option("col_strips_table_hash") == 0
        assert np.array_equal(bits(C), bits(oracle.spmm_omp(ptr, idx_bad, vals, B)))


def test_fold_falls_back_to_the_hub_plan_when_a_former_hub_is_unsorted(device, oracle):
    """the folded plan (no hubs: every row a stripped segment) surveys the former hub rows for the first time.  One of them with columns
    out of order switches the strips off -- and used to leave a 3 000-nonzero row as ONE unstripped segment chain with no hub kernel.  Now the plan with
    hubs is built again: strips for the (sorted) segments, the hub kernel for the hubs."""
    M = K = 65536
    N = 128
    ptr, idx = _strip_case(M, K, 150, 350, seed=17, hubs=(3000, 3000, 2500))
    deg = np.diff(ptr)
    r = int(np.argmax(deg))
    idx[ptr[r] + 3], idx[ptr[r] + 2000] = idx[ptr[r] + 2000], idx[ptr[r] + 3]
    vals = synth.normal_f32(idx.size, 21)
    B = synth.normal_f32(K * N, 22).reshape(K, N)
    C, op = run_spmm(device, ptr, idx, vals, B)
    assert op.get_option("n_hub_rows") >= 1 and op.get_option("long_row_threshold") < 1 << 30
    assert op.get_option("n_col_strips") >= 2 and op.get_option("segments_unsorted") == 0
    assert np.array_equal(bits(C), bits(oracle.spmm_omp(ptr, idx, vals, B)))


def test_autotune_measures_keeps_the_callers_values_and_every_bit(device, oracle):
    """ "autotune" = 1: preprocess times the step under the auto plan and under forced settings of the options the caller left at auto, and keeps the fastest.
    Scheduling only: the bits are the oracle's whatever it picks; an explicit value of the caller's is never changed; a second preprocess starts from auto
    again; switching the option off restores the rules."""
    import torch
    from hpc_amd import CSR, SpMMOpt

    M = K = 40000
    N = 128
    ptr, idx = _strip_case(M, K, 20, 200, seed=71, hubs=(3000, 9000))
    vals = synth.normal_f32(idx.size, 72)
    B = synth.normal_f32(K * N, 73).reshape(K, N)
    exp = oracle.spmm_omp(ptr, idx, vals, B)
    d_ptr, d_idx, d_val, d_B = to_dev(device, ptr, idx, vals, B)
    d_C = torch.full((M, N), float("nan"), dtype=torch.float32, device=device)
    op = SpMMOpt(CSR(M, idx.size, d_ptr, d_idx, d_val), N)
    op.set_option("fused_step", 2)
    op.set_option("autotune", 1)
    op.set_option("tile_cols", 128)                  # the caller's: not the tuner's to change
    op.preprocess(d_B, d_C)
    assert op.get_option("autotune_evals") >= 5 and op.get_option("tile_cols") == 128 and not (op.get_option("autotune_mask") & 1)
    assert 0 < op.get_option("autotune_best_us") <= op.get_option("autotune_auto_us")
    chosen = {k: op.get_option(k) for k in ("col_strips", "medium_row_threshold", "fused_step")}
    mask = op.get_option("autotune_mask")
    assert (chosen["col_strips"] != 0) == bool(mask & 2) and (chosen["fused_step"] != 2) == bool(mask & 8)
    d_C.fill_(float("nan"))
    op.run(d_B, d_C)
    torch.cuda.synchronize()
    assert np.array_equal(bits(d_C.cpu().numpy()), bits(exp))
    op.preprocess(d_B, d_C)                          # again: from auto, not from the last winner
    assert op.get_option("autotune_evals") >= 5
    op.run(d_B, d_C)
    torch.cuda.synchronize()
    assert np.array_equal(bits(d_C.cpu().numpy()), bits(exp))
    op.set_option("autotune", 0)
    op.preprocess(d_B, d_C)
    assert op.get_option("autotune_mask") == 0 and op.get_option("col_strips") == 0 and op.get_option("fused_step") == 2 and op.get_option("tile_cols") == 128
    assert op.get_option("rows_unroll") == 0
    op.run(d_B, d_C)
    torch.cuda.synchronize()
    assert np.array_equal(bits(d_C.cpu().numpy()), bits(exp))
    # the hub threshold is the tuner's to move one notch (mask bit 6) only while the caller left it at auto
    op2 = SpMMOpt(CSR(M, idx.size, d_ptr, d_idx, d_val), N)
    op2.set_option("autotune", 1)
    op2.set_option("long_row_threshold", 2048)
    op2.preprocess(d_B, d_C)
    assert not (op2.get_option("autotune_mask") & 64) and op2.get_option("long_row_threshold") == 2048 and op2.get_option("n_hub_rows") == 2
    d_C.fill_(float("nan"))
    op2.run(d_B, d_C)
    torch.cuda.synchronize()
    assert np.array_equal(bits(d_C.cpu().numpy()), bits(exp))


@pytest.mark.gpu
@pytest.mark.parametrize("N", [4, 32, 64, 100, 128, 256, 384])
def test_rows_unroll_16_same_bits(device, oracle, N):
    """ "rows_unroll" = 16: the rows kernel keeps sixteen B-row gathers in flight per lane group instead of eight -- the same fma chain per row in stored
    order (spmm_ref.cu:10-14), so the same bits: rows of 0 .. 70 nonzeros (empty rows, rows shorter than one batch, rows of several batches plus a tail),
    every lane-group width, two column tiles (384), a width with a partial tile (100).  Values other than 0 / 8 / 16 are refused."""
    from hpc_amd.spmm import MiSpmmError

    M = K = 6000
    rng = np.random.default_rng(5)
    deg = rng.integers(0, 71, M).astype(np.int64)
    deg[:64] = np.arange(64) % 35
    ptr = np.concatenate([[0], np.cumsum(deg)]).astype(np.int32)
    idx = rng.integers(0, K, int(ptr[-1])).astype(np.int32)
    vals = synth.normal_f32(idx.size, 81)
    B = synth.normal_f32(K * N, 82).reshape(K, N)
    exp = oracle.spmm_omp(ptr, idx, vals, B)
    C, op = run_spmm(device, ptr, idx, vals, B, options={"rows_unroll": 16, "medium_row_threshold": 1024})
    assert op.get_option("rows_unroll") == 16 and op.get_option("n_chunks") == 0
    assert np.array_equal(bits(C), bits(exp))
    with pytest.raises(MiSpmmError):
        op.set_option("rows_unroll", 12)
