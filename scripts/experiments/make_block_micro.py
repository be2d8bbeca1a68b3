#!/usr/bin/env python3
"""EXPERIMENT (not part of the library): writes scripts/experiments/block_micro.hip = spmm_block_items' text, taken from
hpc_amd/csrc/spmm_kernels.hpp as it stands, with s_memtime stamps (record / prologue / k loop / epilogue per item) + a host
driver that INCLUDES the product translation unit (so it can read the handle's item table) and runs C4.
    python scripts/experiments/make_block_micro.py && hipcc -O3 --offload-arch=gfx950 -std=c++17 -ffp-contract=off \
        -I hpc_amd/csrc scripts/experiments/block_micro.hip hpc_amd/csrc/preprocess_gpu.hip -o scripts/experiments/build/block_micro
"""
import os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
src = open(os.path.join(ROOT, "hpc_amd/csrc/spmm_kernels.hpp")).read()
i = src.index("template <int XC, int V, int G, bool WIDE, bool RUN>\n__global__ __launch_bounds__(kBlockThreads, 2) void spmm_block_items(BlockArgs a)")
j = src.index("// Cuts a qualifying group's column list (the list of its first row) into runs of consecutive columns.")
k = src[i:j]
def rep(old, new):
    global k
    assert k.count(old) == 1, old
    k = k.replace(old, new)
rep("void spmm_block_items(BlockArgs a)", "void spmm_block_items_stamped(BlockArgs a, unsigned long long *dbg)")
rep("    const int lane = threadIdx.x & 63;\n", "    const unsigned long long t_start = __builtin_amdgcn_s_memtime(), r_start = __builtin_amdgcn_s_memrealtime();\n    const int lane = threadIdx.x & 63;\n")
rep("    if (n_in == 0) return;\n", "    if (n_in == 0) return;\n    const unsigned long long t_rec = __builtin_amdgcn_s_memtime();\n    unsigned long long t_loop0 = 0, t_loop1 = 0;\n")
rep("        int kb = 0;\n", "        asm volatile(\"s_waitcnt vmcnt(16)\" ::: \"memory\");\n        t_loop0 = __builtin_amdgcn_s_memtime();\n        int kb = 0;\n")
rep("        // Epilogue: the tiles that have not left yet\n", "        t_loop1 = __builtin_amdgcn_s_memtime();\n        // Epilogue: the tiles that have not left yet\n")
idx = k.rindex("}\n")
k = k[:idx] + ("    asm volatile(\"s_waitcnt vmcnt(0)\" ::: \"memory\");\n    if (lane == 0) { const unsigned long long t_end = __builtin_amdgcn_s_memtime(); unsigned long long *o = dbg + (size_t)ii * 8;\n"
               "        o[0] = t_rec - t_start; o[1] = t_loop0 - t_rec; o[2] = t_loop1 - t_loop0; o[3] = t_end - t_loop1; o[4] = (unsigned long long)plen[0]; o[5] = (unsigned long long)m; o[6] = r_start; o[7] = __builtin_amdgcn_s_memrealtime(); }\n}\n")
host = r'''
using namespace mi;
#define CK(x) do { int e_ = (int)(x); if (e_ != 0) { printf("%s: %d\n", #x, e_); return 1; } } while (0)
__global__ void gen_c4(int32_t *ptr, int32_t *idx, const int64_t *len1, const int64_t *len2, const int64_t *s1, const int64_t *s2, const int64_t *bptr, int M)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= M) return;
    const int b = r >> 4;
    const int64_t L = len1[b] + len2[b], p0 = bptr[b] + (int64_t)(r & 15) * L;
    ptr[r] = (int32_t)p0;
    if (r == M - 1) ptr[M] = (int32_t)(p0 + L);
    for (int64_t k = 0; k < L; ++k) idx[p0 + k] = (int32_t)(k < len1[b] ? s1[b] + k : s2[b] + (k - len1[b]));
}
int main()
{
    // a C4-like structure made here (two runs of 64 or 128 aligned columns per 16-row group, LCG-drawn): same statistics
    const int M = 1 << 20, K = 1 << 20, N = 256, nb = M / 16;
    std::vector<int64_t> l1(nb), l2(nb), s1(nb), s2(nb), bp(nb + 1);
    unsigned long long s = 88172645463325252ull;
    auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; };
    bp[0] = 0;
    for (int b = 0; b < nb; ++b) {
        l1[b] = 64 * (1 + rnd() % 2); l2[b] = (rnd() % 2) ? 64 * (1 + rnd() % 2) : 0;
        s1[b] = 64 * (rnd() % (K / 64 - 4)); int64_t gap = 64 * (rnd() % (K / 128));
        s2[b] = std::min<int64_t>(s1[b] + l1[b] + gap, (K / 64 - 2) * 64); s2[b] = std::max<int64_t>(s2[b], s1[b] + l1[b]);
        bp[b + 1] = bp[b] + 16 * (l1[b] + l2[b]);
    }
    const int64_t nnz = bp[nb];
    int32_t *d_ptr, *d_idx; float *d_val, *d_B, *d_C; int64_t *d_l1, *d_l2, *d_s1, *d_s2, *d_bp;
    CK(hipMalloc(&d_ptr, (M + 1) * 4)); CK(hipMalloc(&d_idx, nnz * 4)); CK(hipMalloc(&d_val, nnz * 4));
    CK(hipMalloc(&d_B, (size_t)K * N * 4)); CK(hipMalloc(&d_C, (size_t)M * N * 4));
    CK(hipMalloc(&d_l1, nb * 8)); CK(hipMalloc(&d_l2, nb * 8)); CK(hipMalloc(&d_s1, nb * 8)); CK(hipMalloc(&d_s2, nb * 8)); CK(hipMalloc(&d_bp, (nb + 1) * 8));
    CK(hipMemcpy(d_l1, l1.data(), nb * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(d_l2, l2.data(), nb * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_s1, s1.data(), nb * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(d_s2, s2.data(), nb * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_bp, bp.data(), (nb + 1) * 8, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(gen_c4, dim3(M / 256), dim3(256), 0, 0, d_ptr, d_idx, d_l1, d_l2, d_s1, d_s2, d_bp, M);
    CK(mi_spmm_fill_normal(d_val, nnz, 124, 0, 0.f, 0.1f, nullptr)); CK(mi_spmm_fill_normal(d_B, (int64_t)K * N, 125, 0, 0.f, 0.1f, nullptr));
    mi_spmm_handle *h = nullptr;
    CK(mi_spmm_create(&h, d_ptr, d_idx, d_val, M, K, nnz, N));
    CK(mi_spmm_preprocess(h, d_B, d_C));
    printf("nnz %lld, items %d (shared %d), passes %d\n", (long long)nnz, h->n_blk_items, h->n_blk_shared_items, h->n_blk_passes);
    for (int it = 0; it < 3; ++it) CK(mi_spmm_run(h, d_B, d_C, nullptr));
    CK(hipDeviceSynchronize());
    unsigned long long *d_dbg; CK(hipMalloc(&d_dbg, (size_t)h->n_blk_items * 64)); CK(hipMemset(d_dbg, 0, (size_t)h->n_blk_items * 64));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int wpw : {4, 1}) {
        std::vector<unsigned long long> all;
        CK(hipEventRecord(e0, 0));
        for (int pass = 0; pass < h->n_blk_passes; ++pass) {
            const int n = h->blk_launch[pass][2].n, off = h->blk_launch[pass][2].off;
            if (n == 0) continue;
            BlockArgs ba{}; ba.items = h->d_blk_items + off; ba.col_idx = d_idx; ba.vals = d_val; ba.B = d_B; ba.C = d_C; ba.ldb = N; ba.ldc = N; ba.n_items = n; ba.N = N;
            ba.remap = 1; ba.row_lo = 0; ba.row_hi = M; std::memset(&ba.po, 0, sizeof(ba.po));
            hipLaunchKernelGGL((spmm_block_items_stamped<4, 4, 2, false, true>), dim3((n + wpw - 1) / wpw), dim3(64 * wpw), 0, 0, ba, d_dbg + (size_t)off * 8);
        }
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<unsigned long long> hd((size_t)h->n_blk_items * 8);
        CK(hipMemcpy(hd.data(), d_dbg, hd.size() * 8, hipMemcpyDeviceToHost));
        double sum[4] = {0, 0, 0, 0}, tot = 0; long cnt = 0; double byclass[2][3][5] = {};
        for (int i = 0; i < h->n_blk_items; ++i) {
            const unsigned long long *o = &hd[(size_t)i * 8];
            if (o[7] == 0) continue;
            for (int q = 0; q < 4; ++q) sum[q] += (double)o[q];
            tot += (double)(o[7] - o[6]); ++cnt;
            const int lc = o[4] >= 128 ? 1 : 0, mc = (int)std::min<unsigned long long>(o[5], 2ull);
            for (int q = 0; q < 4; ++q) byclass[lc][mc][q] += (double)o[q];
            byclass[lc][mc][4] += 1;
        }
        {   // residency over time from the stamps (s_memtime is the same counter on every XCD as far as this shows)
            unsigned long long t0 = ~0ull, t1 = 0;
            for (int i = 0; i < h->n_blk_items; ++i) { const unsigned long long *o = &hd[(size_t)i * 8]; if (o[7]) { t0 = std::min(t0, o[6]); t1 = std::max(t1, o[7]); } }
            const int NBIN = 40; std::vector<double> act(NBIN, 0.0);
            const double w = (double)(t1 - t0) / NBIN;
            for (int i = 0; i < h->n_blk_items; ++i) {
                const unsigned long long *o = &hd[(size_t)i * 8]; if (!o[7]) continue;
                const double a = (double)(o[6] - t0), b = (double)(o[7] - t0);
                for (int q = (int)(a / w); q < NBIN && q * w < b; ++q) act[q] += (std::min(b, (q + 1) * w) - std::max(a, q * w)) / w;
            }
            printf("  span %.1f us (100 MHz real-time counter); resident waves per SIMD in %d time bins:", (double)(t1 - t0) / 100.0, NBIN);
            for (int q = 0; q < NBIN; ++q) printf(" %.2f", act[q] / 1024.0);
            printf("\n");
        }
        printf("workgroup of %d waves: %.3f ms (both passes, stamped kernel); per item, mean ticks: record %.0f | prologue (A, first B, carried tile) %.0f | k loop %.0f | epilogue %.0f | total %.0f  (%ld items)\n",
               wpw, ms, sum[0] / cnt, sum[1] / cnt, sum[2] / cnt, sum[3] / cnt, tot / cnt, cnt);
        for (int lc = 0; lc < 2; ++lc) for (int mc = 1; mc <= 2; ++mc) if (byclass[lc][mc][4] > 0)
            printf("    L %s, %d piece(s): %.0f items: record %.0f prologue %.0f loop %.0f (= %.0f per MFMA) epilogue %.0f\n", lc ? ">=128" : "64", mc, byclass[lc][mc][4],
                   byclass[lc][mc][0] / byclass[lc][mc][4], byclass[lc][mc][1] / byclass[lc][mc][4], byclass[lc][mc][2] / byclass[lc][mc][4],
                   byclass[lc][mc][2] / byclass[lc][mc][4] / ((lc ? 128 : 64) / 16.0 * 64 * mc), byclass[lc][mc][3] / byclass[lc][mc][4]);
    }
    return 0;
}
'''
head = '''// block_micro.hip -- GENERATED by scripts/experiments/make_block_micro.py; EXPERIMENT, not part of the library.
// Includes the product translation unit to reach the handle's item table.
#include "mi_spmm.hip"
#include <cstdio>
namespace mi {
'''
open(os.path.join(ROOT, "scripts/experiments/block_micro.hip"), "w").write(head + k + "\n}  // namespace mi\n" + host)
print("wrote scripts/experiments/block_micro.hip")
