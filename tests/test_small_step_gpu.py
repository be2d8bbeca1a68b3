"""The small-step kernel (spmm_kernels.hpp spmm_small_step, "fused_step"): hub rows, segments and short rows as the three roles of ONE launch.
Same device functions, same arguments, same arithmetic as the separate kernels -> the oracle's bits, through the C ABI."""
import ctypes

import numpy as np
import pytest

from conftest import bits, to_dev
from hpc_amd import synth

pytestmark = pytest.mark.gpu


def _graph(M, K, seed, hubs=(700, 1500), lo=0, hi=120):
    """short rows (incl. empty ones), medium rows and a few hubs; columns ascending"""
    g = np.random.Generator(np.random.Philox(key=[seed, 5]))
    deg = g.integers(lo, hi + 1, size=M).astype(np.int64)
    deg[g.choice(M, size=M // 3, replace=False)] = g.integers(0, 9, size=M // 3)
    for r, L in zip(g.choice(M, size=len(hubs), replace=False), hubs):
        deg[r] = min(L, K)
    ptr = np.zeros(M + 1, np.int64)
    np.cumsum(deg, out=ptr[1:])
    idx = np.empty(int(ptr[-1]), np.int32)
    for r in range(M):
        idx[ptr[r]:ptr[r + 1]] = np.sort(g.choice(K, int(deg[r]), replace=False))
    return ptr.astype(np.int32), idx


def _op(device, ptr, idx, vals, N, K, opts):
    from hpc_amd import CSR, SpMMOpt

    d_ptr, d_idx, d_val = to_dev(device, ptr, idx, vals)
    op = SpMMOpt(CSR(ptr.size - 1, idx.size, d_ptr, d_idx, d_val), N, num_cols=K)
    for k, v in opts.items():
        op.set_option(k, v)
    op._keep = (d_ptr, d_idx, d_val)
    return op


@pytest.mark.parametrize("N", [4, 7, 32, 64, 100, 128, 256])
def test_small_step_gives_the_separate_kernels_bits(device, oracle, N):
    import torch

    M, K = 3000, 4100
    ptr, idx = _graph(M, K, seed=100 + N)
    vals = synth.normal_f32(idx.size, 3)
    ldb, ldc = N + (0 if N % 8 else 4), N + (5 if N == 100 else 0)
    Bp = synth.normal_f32(K * ldb, 4).reshape(K, ldb)
    exp = oracle.spmm_omp(ptr, idx, vals, np.ascontiguousarray(Bp[:, :N]))
    (d_B,) = to_dev(device, Bp)
    for fused, ranges in ((1, False), (0, False), (1, True), (2, False)):
        op = _op(device, ptr, idx, vals, N, K, {"fused_step": fused, "medium_row_threshold": 24, "long_row_threshold": 512})
        d_C = torch.full((M, ldc), float("nan"), dtype=torch.float32, device=device)
        op.preprocess(d_B, d_C)
        assert op.get_option("n_hub_rows") == 2 and op.get_option("n_medium_rows") > 100
        for rep in range(2):                          # idempotent
            if ranges:
                for r0, r1 in ((0, 1), (1, 900), (900, 901), (901, M)):
                    op.run_rows(d_B, ldb, d_C, ldc, r0, r1)
            else:
                op.run_ld(d_B, ldb, d_C, ldc)
        torch.cuda.synchronize()
        assert op.get_option("fused_step_in_force") == (1 if fused else 0), (N, fused)
        assert op.get_option("n_launches") == (1 if fused else 3), (N, fused, op.get_option("n_launches"))
        got = d_C.cpu().numpy()
        assert np.array_equal(bits(got[:, :N]), bits(exp)), (N, fused, ranges, int((bits(got[:, :N]) != bits(exp)).sum()))
        assert np.isnan(got[:, N:]).all()


def test_small_step_roles_may_be_absent_and_ineligible_steps_keep_their_kernels(device, oracle):
    import torch

    M, K, N = 2000, 2500, 32
    B = synth.normal_f32(K * N, 8).reshape(K, N)
    (d_B,) = to_dev(device, B)
    cases = {
        "no hubs": (_graph(M, K, 1, hubs=()), {"medium_row_threshold": 24}, 1),
        "no short rows": (synth.csr_uniform(M, 300, 600, K=K, seed=2), {"medium_row_threshold": 24, "long_row_threshold": 512}, 1),
        "only short rows": (synth.csr_uniform(M, 0, 20, K=K, seed=3), {}, 0),                # one launch anyway: nothing to fuse
        "column strips in force": (synth.csr_uniform(M, 300, 600, K=K, seed=2), {"col_strips": 3, "long_row_threshold": 1 << 30}, 0),
        "split rows": (_graph(M, K, 4), {"split_long_rows": 1, "long_row_threshold": 512}, 0),
        "plain stores asked for": (_graph(M, K, 5), {"nt_store": 0, "long_row_threshold": 512}, 0),
    }
    for name, ((ptr, idx), opts, fused_expected) in cases.items():
        vals = synth.normal_f32(idx.size, 9)
        op = _op(device, ptr, idx, vals, N, K, dict(opts, fused_step=1))
        d_C = torch.full((M, N), float("nan"), dtype=torch.float32, device=device)
        op.preprocess(d_B, d_C)
        op.run(d_B, d_C)
        torch.cuda.synchronize()
        assert op.get_option("fused_step_in_force") == fused_expected, name
        if opts.get("split_long_rows"):
            continue          # (its own tolerance tests: test_parity_gpu.py)
        assert np.array_equal(bits(d_C.cpu().numpy()), bits(oracle.spmm_omp(ptr, idx, vals, B))), name


def test_small_step_special_values_flush_to_zero_and_extra_destinations(device, oracle):
    import torch
    from hpc_amd import _lib

    M, K, N = 1500, 1800, 64
    ptr, idx = _graph(M, K, 21, hubs=(900,))
    vals = synth.normal_f32(idx.size, 22)
    B = synth.normal_f32(K * N, 23).reshape(K, N)
    g = np.random.Generator(np.random.Philox(key=[9, 1]))
    vals[g.integers(0, vals.size, 30)] = np.float32(np.inf)
    vals[g.integers(0, vals.size, 300)] = np.float32(1e-30)
    vals[g.integers(0, vals.size, 30)] = np.float32(-0.0)
    B[g.integers(0, K, 20), g.integers(0, N, 20)] = np.float32(np.nan)
    B[g.integers(0, K, 200), :] *= np.float32(-1e-12)
    (d_B,) = to_dev(device, B)
    for ftz in (0, 1):
        ref = oracle.spmm_ftz(ptr, idx, vals, B) if ftz else oracle.spmm_omp(ptr, idx, vals, B)
        op = _op(device, ptr, idx, vals, N, K, {"fused_step": 1, "flush_denormals": ftz, "long_row_threshold": 512})
        d_C = torch.full((M, N), float("nan"), dtype=torch.float32, device=device)
        op.preprocess(d_B, d_C)
        extra = [torch.full((M, N), float("nan"), dtype=torch.float32, device=device) for _ in range(3)]
        arr = (ctypes.c_void_p * 3)(*[ctypes.c_void_p(t.data_ptr()) for t in extra])
        rc = _lib.load().mi_spmm_run_rows_multi(op._h, ctypes.c_void_p(d_B.data_ptr()), N, ctypes.c_void_p(d_C.data_ptr()), N, 0, M, 3, arr, None)
        assert rc == 0
        torch.cuda.synchronize()
        assert op.get_option("fused_step_in_force") == 1
        for t in [d_C] + extra:
            assert np.array_equal(bits(t.cpu().numpy()), bits(ref)), ftz


@pytest.mark.parametrize("name,N", [("arxiv", 32), ("collab", 32), ("ddi", 32), ("ddi", 128)])
def test_small_step_is_the_default_on_short_steps_and_matches_the_reference_kernel(device, oracle, name, N):
    """auto ("fused_step" = 2): a step whose bytes take under 0.1 ms (an L2-resident B priced at the L2's rate: ddi at N = 128) goes through ONE launch -- the
    dataset-shaped graphs the reference's report times at tens of microseconds -- and the whole C equals spmm_kernel_ref's; a C1-sized step keeps its
    kernels (tests/test_fullsize_gpu.py runs those)."""
    import torch
    from hpc_amd import CSR, SpMMOpt
    from hpc_amd.spmm import count_bitdiff, fill_normal

    d_ptr, d_idx = synth.csr_dataset_structured_device(name, device)
    M, nnz = d_ptr.numel() - 1, int(d_idx.numel())
    d_val = torch.empty(nnz, dtype=torch.float32, device=device)
    fill_normal(d_val, 124)
    d_B = torch.empty(M * N, dtype=torch.float32, device=device)
    fill_normal(d_B, 125)
    d_B = d_B.view(M, N)
    d_C = torch.full((M, N), float("nan"), dtype=torch.float32, device=device)
    op = SpMMOpt(CSR(M, nnz, d_ptr, d_idx, d_val), N)
    op.set_option("fused_step", 2)          # the library's default (the test session's Python-side default is 0: conftest.py)
    op.preprocess(d_B, d_C)
    op.run(d_B, d_C)
    assert op.get_option("fused_step_in_force") == 1 and op.get_option("n_launches") == 1
    # "fused_order" auto: arxiv's 12 681-nonzero hub row (40 us of chain) leads its grid; ddi's and collab's 512- / 256-nonzero segments outlast their hub rows
    assert op.get_option("fused_order_in_force") == (1 if name == "arxiv" else 2), (name, N, op.get_option("fused_order_in_force"))
    if not oracle.ref_available():
        pytest.fail("oracle/_ref missing")
    d_R = torch.full((M, N), float("nan"), dtype=torch.float32, device=device)
    oracle.ref_kernel_run(d_ptr, d_idx, d_val, d_B, d_R, M, N)
    assert count_bitdiff(d_C, d_R) == (0, 0.0)


@pytest.mark.parametrize("N", [8, 32, 100, 128])
def test_small_step_role_order_is_scheduling_only(device, oracle, N):
    """ "fused_order" 1 (hub workgroups lead the grid) / 2 (segment workgroups lead): the two roles trade blockIdx ranges, nothing else -- the oracle's bits
    either way, with every role present, with row ranges, and with one of the two roles absent (then there is nothing to trade)."""
    import torch

    M, K = 2500, 3000
    ptr, idx = _graph(M, K, seed=300 + N, hubs=(600, 900, 2100))
    vals = synth.normal_f32(idx.size, 13)
    Bm = synth.normal_f32(K * N, 14).reshape(K, N)
    exp = oracle.spmm_omp(ptr, idx, vals, Bm)
    (d_B,) = to_dev(device, Bm)
    for order in (1, 2, 0):
        for opts, hubs in (({"long_row_threshold": 512}, 3), ({"long_row_threshold": 8192}, 0)):
            op = _op(device, ptr, idx, vals, N, K, dict(opts, fused_step=1, fused_order=order, medium_row_threshold=24))
            d_C = torch.full((M, N), float("nan"), dtype=torch.float32, device=device)
            op.preprocess(d_B, d_C)
            assert op.get_option("n_hub_rows") == hubs
            op.run(d_B, d_C)
            in_force = op.get_option("fused_order_in_force")
            for r0, r1 in ((0, 700), (700, M)):
                op.run_rows(d_B, N, d_C, N, r0, r1)
            torch.cuda.synchronize()
            assert op.get_option("fused_step_in_force") == 1
            if order and hubs:
                assert in_force == order, (N, order, in_force)
            if not hubs:
                assert in_force == 1            # no hub role: nothing to put segments in front of
            got = d_C.cpu().numpy()
            assert np.array_equal(bits(got), bits(exp)), (N, order, hubs, int((bits(got) != bits(exp)).sum()))
    with pytest.raises(Exception):
        op.set_option("fused_order", 3)
