// preprocess_gpu.hip -- the segment table built on the device.
//
// The reference builds its Task list with a single-threaded host loop over row_ptr after a D2H copy
// (PA4/workspace/src/spmm_opt.cu:38-62).  Here: one classify kernel over the rows, three exclusive
// scans (hipCUB), one emit kernel, one stable radix sort by segment length, one flagged select for the
// block-path groups -- and a single 48-byte copy back to size the allocations.
#include "plan.hpp"

#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <cstdio>
#include <cstdlib>

#include "../../include/mi_spmm.h"

namespace mi {

struct PlanStats {
    int32_t max_len;
    uint32_t bad;      // bit0: negative row length, bit1: ptr[0] < 0
    int32_t ptr0, ptrM;
    int32_t n_groups_selected;
    int32_t mthr;      // resolved medium threshold (auto rule below, or the caller's value), already capped by thr
    int32_t near, sampled;   // column locality sample: nonzeros of sampled rows whose column lies near the row's own position
    int32_t thr;       // resolved hub threshold (the caller's value or resolve_hub_threshold)
    int32_t pad;
    LenHist hist;      // rows above 256 .. 8192 nonzeros and what they hold (the auto hub threshold reads it)
};

__global__ void init_plan_stats(PlanStats *stats)
{
    if (threadIdx.x == 0) *stats = PlanStats{};
}

// Column locality of a row sample: how many nonzeros sit within `window` columns of their row's own position
// (row r of M <-> column r*K/M).  Mesh / community / banded structures score ~1, random columns ~2*window/K.
// Only the column-tile width of the rows kernel depends on it (scheduling, never arithmetic).
__global__ __launch_bounds__(kBlockThreads) void sample_locality(const int32_t *__restrict__ row_ptr, const int32_t *__restrict__ col_idx,
                                                                int32_t M, int32_t K, int64_t nnz, int32_t n_samples, int32_t window, PlanStats *stats)
{
    const int i = (int)(blockIdx.x * (unsigned)kBlockThreads + threadIdx.x);
    int near = 0, tot = 0;
    if (i < n_samples) {
        const int r = (int)((int64_t)i * M / n_samples);
        const int beg = row_ptr[r], end = row_ptr[r + 1];
        const int64_t home = (int64_t)r * K / (M > 0 ? M : 1);
        if (beg >= 0 && end <= nnz && end - beg <= 4096) {     // runs before row_ptr has been validated; hubs say nothing about locality
            for (int k = beg; k < end; ++k) {
                const int64_t d = (int64_t)col_idx[k] - home;
                near += (d <= window && d >= -(int64_t)window) ? 1 : 0;
            }
            tot = end - beg;
        }
    }
    for (int off = 32; off > 0; off >>= 1) {
        near += __shfl_down(near, off, 64);
        tot += __shfl_down(tot, off, 64);
    }
    if ((threadIdx.x & 63) == 0 && tot > 0) {
        atomicAdd(&stats->near, near);
        atomicAdd(&stats->sampled, tot);
    }
}

// Longest row, ahead of the classification: the auto medium threshold depends on it.
__global__ __launch_bounds__(kBlockThreads) void row_len_max(const int32_t *__restrict__ row_ptr, int32_t M, PlanStats *stats)
{
    const int64_t r = (int64_t)blockIdx.x * kBlockThreads + threadIdx.x;
    const int len = (r < M) ? row_ptr[r + 1] - row_ptr[r] : 0;
    // rows above 256, 512, ... 8192 nonzeros: count and content, one pair of atomics per wave and threshold that has any
#pragma unroll
    for (int i = 0; i < kHistN; ++i) {
        const bool above = len > hist_threshold(i);
        const unsigned long long mask = __ballot(above);
        if (mask == 0) break;                              // wave-uniform; a longer threshold has none either
        unsigned long long part = above ? (unsigned long long)len : 0ull;
        for (int off = 32; off > 0; off >>= 1) part += __shfl_down(part, off, 64);
        if ((threadIdx.x & 63) == 0) {
            atomicAdd(&stats->hist.cnt[i], (uint32_t)__builtin_popcountll(mask));
            atomicAdd(&stats->hist.nnz[i], part);
        }
    }
    int m = len;
    for (int off = 32; off > 0; off >>= 1) m = max(m, __shfl_down(m, off, 64));
    // one same-address atomic per wave was most of this kernel's time (16 K waves): skip it unless it can raise the max
    if ((threadIdx.x & 63) == 0 && m > __hip_atomic_load(&stats->max_len, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
        atomicMax(&stats->max_len, m);
}

// Rows above the medium threshold leave the rows kernel for the length-sorted segment kernel.  auto: 64 when
// the degrees are even (the rows kernel's lane groups finish together anyway), 32 when the longest row is more
// than 8x the mean -- skewed graphs, where neighbours in a wave differ widely (profiles/r01_medium_threshold.txt).
__device__ __forceinline__ int resolve_mthr(int mthr_user, int mean_len, int max_len, int thr)
{
    int m = mthr_user > 0 ? mthr_user : ((int64_t)max_len > 8 * (int64_t)(mean_len > 1 ? mean_len : 1) ? 32 : 64);
    return m < thr ? m : thr;
}

__global__ __launch_bounds__(kBlockThreads) void classify_rows(const int32_t *__restrict__ row_ptr, int32_t M,
                                                              const uint8_t *__restrict__ blk_flag, int32_t mthr_user,
                                                              int32_t mean_len, int32_t thr_user, int32_t clen, int32_t split,
                                                              int64_t nnz, int32_t N,
                                                              int32_t *__restrict__ seg_cnt,
                                                              int32_t *__restrict__ slot_cnt,
                                                              int32_t *__restrict__ long_cnt, PlanStats *stats)
{
    const int64_t r = (int64_t)blockIdx.x * kBlockThreads + threadIdx.x;
    int len = 0;
    unsigned bad = 0;
    // max_len and the histogram are complete (previous kernel); every thread resolves the same two thresholds
    const int thr = thr_user > 0 ? thr_user : resolve_hub_threshold(nnz, M, N, stats->hist.nnz);
    const int mthr = resolve_mthr(mthr_user, mean_len, stats->max_len, thr);
    if (r < M) {
        const int beg = row_ptr[r], end = row_ptr[r + 1];
        len = end - beg;
        if (len < 0) { bad = 1; len = 0; }
        int segs = 0, slots = 0, lng = 0;
        if (len > mthr && !(blk_flag && blk_flag[r >> 4])) {   // (a 16-row group the block path took stays with it, whatever its length)
            if (len <= thr) {
                segs = 1;                                        // medium row: one exact segment
            } else {
                // hub.  split mode: pieces + partial-sum slots (len > thr >= 1; no int32 overflow); default: no segment
                // at all -- the hub kernel walks the row in stored order
                if (split) segs = slots = 1 + (len - 1) / clen;
                lng = 1;
            }
        }
        seg_cnt[r] = segs;
        slot_cnt[r] = slots;
        long_cnt[r] = lng;
        if (r == 0) { stats->ptr0 = beg; stats->mthr = mthr; stats->thr = thr; if (beg < 0) bad |= 2; }
        if (r == M - 1) stats->ptrM = end;
    } else if (r == M) {   // trailing zero so the exclusive scans leave the totals at index M
        seg_cnt[r] = 0;
        slot_cnt[r] = 0;
        long_cnt[r] = 0;
    }
    if (bad) atomicOr(&stats->bad, bad);   // malformed input only
}

__global__ __launch_bounds__(kBlockThreads) void emit_segments(const int32_t *__restrict__ row_ptr, int32_t M,
                                                              int32_t thr, int32_t clen,
                                                              const int32_t *__restrict__ seg_cnt,
                                                              const int32_t *__restrict__ seg_off,
                                                              const int32_t *__restrict__ slot_off,
                                                              const int32_t *__restrict__ long_off,
                                                              Chunk *__restrict__ chunks, uint32_t *__restrict__ keys,
                                                              LongRow *__restrict__ longs, uint32_t *__restrict__ long_keys)
{
    const int64_t r = (int64_t)blockIdx.x * kBlockThreads + threadIdx.x;
    if (r >= M) return;
    const int n = seg_cnt[r];
    const int beg = row_ptr[r], end = row_ptr[r + 1];
    const bool hub = long_off[r + 1] != long_off[r];       // classified as one (a long row of a block group is not)
    if (!hub) {
        if (n == 0) return;
        const int o = seg_off[r];
        Chunk c;
        c.beg = beg;
        c.end = end;
        c.slot = -1;
        c.row = (int32_t)r;
        chunks[o] = c;
        keys[o] = (uint32_t)(end - beg);
        return;
    }
    // a hub: listed in either mode (n == 0: the hub kernel takes the row whole; n > 0: its pieces follow)
    int o = seg_off[r];
    int slot = slot_off[r];
    LongRow L;
    L.row = (int32_t)r;
    L.first_slot = n > 0 ? slot : -1;
    L.n_chunks = n;
    L.len = end - beg;
    longs[long_off[r]] = L;
    long_keys[long_off[r]] = (uint32_t)(end - beg);
    if (n == 0) return;
    for (int b = beg; b < end; b += (end - b > clen ? clen : end - b), ++o, ++slot) {
        Chunk c;
        c.beg = b;
        c.end = (end - b > clen) ? b + clen : end;
        c.slot = slot;
        c.row = (int32_t)r;
        chunks[o] = c;
        keys[o] = (uint32_t)(c.end - c.beg);
    }
}

// Temporaries are carved out of two grow-only arenas the handle owns (plan.hpp Scratch).  What is guaranteed: between the
// first launch of a preprocess and the copy-back of its statistics there is no allocation call at all, and no temporary
// ever comes from hipMallocAsync's pool; a repeated preprocess allocates no temporaries.  What still allocates: the
// plan's OUTPUT tables (segments, hub rows, block groups) are synchronous hipMalloc calls placed after that copy-back has
// been waited for (at most the block-group list's device-to-device copy is in flight then; no kernel is).  (Round 2: pool buffers allocated while earlier kernels of the same
// preprocess were in flight were seen overlapping rocPRIM's scan state in a torch-free process -- the plan statistics
// came back zeroed now and then.  Two changes went in together -- pool -> arenas, hipMemsetAsync(stats) -> init kernel --
// and which of them removed the failure was not isolated; test_native_harness_end_to_end is the regression.)
namespace {
struct Carver {
    char *base;
    size_t off = 0;
    explicit Carver(void *b) : base(reinterpret_cast<char *>(b)) {}
    template <class T> T *take(size_t n)
    {
        off = (off + 255) & ~(size_t)255;
        T *r = reinterpret_cast<T *>(base + off);
        off += (n ? n : 1) * sizeof(T);
        return r;
    }
};
}  // namespace

int scratch_reserve(Scratch *s, size_t bytes)
{
    if (s->cap >= bytes && s->p) return 0;
    if (s->p) (void)hipFree(s->p);
    s->p = nullptr;
    s->cap = 0;
    if (hipMalloc(&s->p, bytes ? bytes : 256) != hipSuccess) return MI_SPMM_ENOMEM;
    s->cap = bytes ? bytes : 256;
    return 0;
}

void scratch_release(Scratch *s)
{
    if (s->p) (void)hipFree(s->p);
    s->p = nullptr;
    s->cap = 0;
}

#define PLAN_TRY(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return (e_ == hipErrorOutOfMemory) ? MI_SPMM_ENOMEM : (int)e_; } while (0)

int build_plan_gpu(const int32_t *d_row_ptr, const int32_t *d_col_idx, int32_t M, int32_t K, int32_t N, int64_t nnz,
                   const uint8_t *d_blk_flag, const unsigned int *d_col_bad, int32_t mthr, int32_t thr_user, int32_t clen, int32_t split,
                   Scratch *sa, Scratch *sb, PlanOut *out)
{
    *out = PlanOut();
    out->thr = thr_user > 0 ? thr_user : 256;                                   // both replaced by the device's values below
    out->mthr = (mthr > 0 ? mthr : 64) < out->thr ? (mthr > 0 ? mthr : 64) : out->thr;
    if (M <= 0) return (nnz == 0) ? MI_SPMM_OK : MI_SPMM_ECSR;
    const size_t n1 = (size_t)M + 1;
    const int n_groups_all = (M + 15) / 16;
    // sizes first (the queries touch no memory), then ONE reservation, then the launches
    size_t tb = 0, tb2 = 0;
    PLAN_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, tb, (int32_t *)nullptr, (int32_t *)nullptr, (int)n1));
    if (d_blk_flag)
        PLAN_TRY(hipcub::DeviceSelect::Flagged(nullptr, tb2, hipcub::CountingInputIterator<int32_t>(0), d_blk_flag,
                                               (int32_t *)nullptr, (int32_t *)nullptr, n_groups_all));
    int32_t *cnt = nullptr, *off = nullptr, *groups = nullptr;
    PlanStats *stats = nullptr;
    char *tmp = nullptr;
    auto layout_a = [&](Carver &c) {
        cnt = c.take<int32_t>(3 * n1);
        off = c.take<int32_t>(3 * n1);
        stats = c.take<PlanStats>(1);
        tmp = c.take<char>(tb > tb2 ? tb : tb2);
        groups = c.take<int32_t>(d_blk_flag ? (size_t)n_groups_all : 1);
    };
    {
        Carver dry(nullptr);
        layout_a(dry);
        const int rc = scratch_reserve(sa, dry.off + 256);
        if (rc != 0) return rc;
        Carver real(sa->p);
        layout_a(real);
    }
    hipLaunchKernelGGL(init_plan_stats, dim3(1), dim3(64), 0, 0, stats);   // same queue as the kernels that fill it
    int32_t *seg_cnt = cnt, *slot_cnt = seg_cnt + n1, *long_cnt = slot_cnt + n1;
    int32_t *seg_off = off, *slot_off = seg_off + n1, *long_off = slot_off + n1;
    const unsigned grid = (unsigned)((n1 + kBlockThreads - 1) / kBlockThreads);
    hipLaunchKernelGGL(row_len_max, dim3(grid), dim3(kBlockThreads), 0, 0, d_row_ptr, M, stats);
    if (nnz > 0 && K > 0) {
        const int n_samples = M < 8192 ? M : 8192;
        const int window = K / 64 > 4096 ? K / 64 : 4096;
        hipLaunchKernelGGL(sample_locality, dim3((unsigned)((n_samples + kBlockThreads - 1) / kBlockThreads)), dim3(kBlockThreads), 0, 0,
                           d_row_ptr, d_col_idx, M, K, nnz, n_samples, window, stats);
    }
    hipLaunchKernelGGL(classify_rows, dim3(grid), dim3(kBlockThreads), 0, 0, d_row_ptr, M, d_blk_flag, mthr,
                       (int32_t)(nnz / M), thr_user, clen, split, nnz, N, seg_cnt, slot_cnt, long_cnt, stats);
    PLAN_TRY(hipGetLastError());
    size_t t = tb;
    PLAN_TRY(hipcub::DeviceScan::ExclusiveSum(tmp, t, seg_cnt, seg_off, (int)n1));
    t = tb;
    PLAN_TRY(hipcub::DeviceScan::ExclusiveSum(tmp, t, slot_cnt, slot_off, (int)n1));
    t = tb;
    PLAN_TRY(hipcub::DeviceScan::ExclusiveSum(tmp, t, long_cnt, long_off, (int)n1));
    if (d_blk_flag) {
        t = tb2;
        PLAN_TRY(hipcub::DeviceSelect::Flagged(tmp, t, hipcub::CountingInputIterator<int32_t>(0), d_blk_flag, groups,
                                               &stats->n_groups_selected, n_groups_all));
    }
    // the one copy back: totals (element M of each scan), stats, and the column-check flag
    struct { int32_t n_chunks, n_slots, n_long; PlanStats st; unsigned int col_bad; } host;
    host.col_bad = 0;
    PLAN_TRY(hipMemcpyAsync(&host.n_chunks, seg_off + M, sizeof(int32_t), hipMemcpyDeviceToHost, 0));
    PLAN_TRY(hipMemcpyAsync(&host.n_slots, slot_off + M, sizeof(int32_t), hipMemcpyDeviceToHost, 0));
    PLAN_TRY(hipMemcpyAsync(&host.n_long, long_off + M, sizeof(int32_t), hipMemcpyDeviceToHost, 0));
    PLAN_TRY(hipMemcpyAsync(&host.st, stats, sizeof(PlanStats), hipMemcpyDeviceToHost, 0));
    if (d_col_bad) PLAN_TRY(hipMemcpyAsync(&host.col_bad, d_col_bad, sizeof(unsigned int), hipMemcpyDeviceToHost, 0));
    PLAN_TRY(hipStreamSynchronize(0));
    // data.cu:40-45 asserts ptr[num_v] == num_e; monotone rows and in-range columns keep the kernels in bounds
    if (host.st.bad || host.col_bad || (int64_t)host.st.ptrM != nnz || host.st.ptr0 < 0) {
        if (getenv("MI_SPMM_DEBUG"))
            fprintf(stderr, "mi_spmm: CSR rejected: negative-length/ptr0 bits %u, column out of range %u, row_ptr[0] = %d, row_ptr[M] = %d, nnz = %lld "
                            "(M = %d, longest row %d, medium threshold %d, segments %d)\n",
                    host.st.bad, host.col_bad, host.st.ptr0, host.st.ptrM, (long long)nnz, M, host.st.max_len, host.st.mthr, host.n_chunks);
        return MI_SPMM_ECSR;
    }
    out->max_len = host.st.max_len;
    out->mthr = host.st.mthr;
    out->thr = host.st.thr;
    const int32_t thr = host.st.thr;
    out->local_pct = host.st.sampled > 0 ? (int32_t)(100.0 * host.st.near / host.st.sampled) : 0;
    out->n_chunks = host.n_chunks;
    out->n_slots = host.n_slots;
    out->n_long = host.n_long;
    out->n_medium = host.n_chunks - host.n_slots;
    if (d_blk_flag && host.st.n_groups_selected > 0) {
        out->n_blk_groups = host.st.n_groups_selected;
        // the qualifying groups in row order; build_block_items (mi_spmm.hip) cuts them into pieces and orders those
        const int ng = out->n_blk_groups;
        PLAN_TRY(hipMalloc((void **)&out->d_blk_groups, (size_t)ng * sizeof(int32_t)));
        PLAN_TRY(hipMemcpyAsync(out->d_blk_groups, groups, (size_t)ng * sizeof(int32_t), hipMemcpyDeviceToDevice, 0));
    }
    if (host.n_chunks > 0 || host.n_long > 0) {
        const size_t n = (size_t)host.n_chunks, nl = (size_t)host.n_long;
        int end_bit = 1;
        while (end_bit < 32 && (host.st.max_len >> end_bit)) ++end_bit;
        // segments longest first; hub rows longest first (the hub kernel's tail is its longest row: start it first).
        // Both sorts are stable: equal lengths stay in row order.
        size_t sbytes = 0, lbytes = 0;
        if (n > 0)
            PLAN_TRY(hipcub::DeviceRadixSort::SortPairsDescending(nullptr, sbytes, (uint32_t *)nullptr, (uint32_t *)nullptr, (Chunk *)nullptr,
                                                                  (Chunk *)nullptr, (int)n, 0, end_bit));
        if (nl > 0)
            PLAN_TRY(hipcub::DeviceRadixSort::SortPairsDescending(nullptr, lbytes, (uint32_t *)nullptr, (uint32_t *)nullptr, (LongRow *)nullptr,
                                                                  (LongRow *)nullptr, (int)nl, 0, end_bit));
        Chunk *unsorted = nullptr;
        LongRow *longs_unsorted = nullptr;
        uint32_t *keys_in = nullptr, *keys_out = nullptr, *lkeys_in = nullptr, *lkeys_out = nullptr;
        char *tmp2 = nullptr;
        auto layout_b = [&](Carver &c) {
            unsorted = c.take<Chunk>(n);
            keys_in = c.take<uint32_t>(n);
            keys_out = c.take<uint32_t>(n);
            longs_unsorted = c.take<LongRow>(nl);
            lkeys_in = c.take<uint32_t>(nl);
            lkeys_out = c.take<uint32_t>(nl);
            tmp2 = c.take<char>(sbytes > lbytes ? sbytes : lbytes);
        };
        {
            Carver dry(nullptr);
            layout_b(dry);
            const int rc = scratch_reserve(sb, dry.off + 256);
            if (rc != 0) return rc;
            Carver real(sb->p);
            layout_b(real);
        }
        PLAN_TRY(hipMalloc((void **)&out->d_chunks, (n > 0 ? n : 1) * sizeof(Chunk)));
        PLAN_TRY(hipMalloc((void **)&out->d_long, (nl > 0 ? nl : 1) * sizeof(LongRow)));
        hipLaunchKernelGGL(emit_segments, dim3((unsigned)(((size_t)M + kBlockThreads - 1) / kBlockThreads)),
                           dim3(kBlockThreads), 0, 0, d_row_ptr, M, thr, clen, seg_cnt, seg_off, slot_off, long_off,
                           unsorted, keys_in, longs_unsorted, lkeys_in);
        PLAN_TRY(hipGetLastError());
        if (n > 0) PLAN_TRY(hipcub::DeviceRadixSort::SortPairsDescending(tmp2, sbytes, keys_in, keys_out, unsorted, out->d_chunks, (int)n, 0, end_bit));
        if (nl > 0) PLAN_TRY(hipcub::DeviceRadixSort::SortPairsDescending(tmp2, lbytes, lkeys_in, lkeys_out, longs_unsorted, out->d_long, (int)nl, 0, end_bit));
    }
    PLAN_TRY(hipStreamSynchronize(0));
    return MI_SPMM_OK;
}

}  // namespace mi
