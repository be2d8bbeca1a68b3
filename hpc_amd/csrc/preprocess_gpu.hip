// preprocess_gpu.hip -- the segment table built on the device.
//
// The reference builds its Task list with a single-threaded host loop over row_ptr after a D2H copy
// (PA4/workspace/src/spmm_opt.cu:38-62).  Here: one classify kernel over the rows, three exclusive
// scans (hipCUB), one emit kernel, one stable radix sort by segment length, one flagged select for the
// block-path groups -- and a single 48-byte copy back to size the allocations.
#include "plan.hpp"

#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <climits>
#include <cstdint>
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
    int32_t front, pad2;     // ... and those whose column lies in the first quarter of the columns (hubs-first vertex orders)
    int32_t thr;       // resolved hub threshold (the caller's value or resolve_hub_threshold)
    int32_t pad;
    LenHist hist;      // rows above 256 .. 8192 nonzeros and what they hold (the auto hub threshold reads it)
    unsigned long long seg_nnz;   // nonzeros of the rows that became whole (exact) segments: the column-strip rule reads it
};

__global__ void init_plan_stats(PlanStats *stats)
{
    if (threadIdx.x == 0) *stats = PlanStats{};
}

// Column locality of a row sample: how many nonzeros sit within `window` columns of their row's own position
// (row r of M <-> column r*K/M).  Mesh / community / banded structures score ~1, random columns ~2*window/K.
// Only the column-tile width of the rows kernel depends on it (scheduling, never arithmetic).
// "near the row's own position": within max(4096, K / 64) columns -- but never more than K / 16 (round 5: a 4 267-column matrix had EVERY column near,
// 99 % local, and was taken for a banded one)
static inline int locality_window(int K)
{
    int w = K / 64 > 4096 ? K / 64 : 4096;
    if (w > K / 16) w = K / 16;
    return w < 1 ? 1 : w;
}

// (round 5: one WAVE per sampled row, lanes striding its columns -- one thread per row walked a 500-nonzero row serially: 142 us of the LONG_ROWS preprocess)
__global__ __launch_bounds__(kBlockThreads) void sample_locality(const int32_t *__restrict__ row_ptr, const int32_t *__restrict__ col_idx,
                                                                int32_t M, int32_t K, int64_t nnz, int32_t n_samples, int32_t window, PlanStats *stats)
{
    const int i = (int)((blockIdx.x * (unsigned)kBlockThreads + threadIdx.x) >> 6), lane = (int)(threadIdx.x & 63);
    int near = 0, tot = 0, front = 0;
    if (i < n_samples) {
        const int r = (int)((int64_t)i * M / n_samples);
        const int beg = row_ptr[r], end = row_ptr[r + 1];
        const int64_t home = (int64_t)r * K / (M > 0 ? M : 1);
        if (beg >= 0 && end <= nnz && end - beg <= 4096) {     // runs before row_ptr has been validated; hubs say nothing about locality
            for (int k = beg + lane; k < end; k += 64) {
                const int32_t c = col_idx[k];
                const int64_t d = (int64_t)c - home;
                near += (d <= window && d >= -(int64_t)window) ? 1 : 0;
                front += c < K / 4 ? 1 : 0;
            }
            tot = lane == 0 ? end - beg : 0;
        }
    }
    for (int off = 32; off > 0; off >>= 1) {
        near += __shfl_down(near, off, 64);
        tot += __shfl_down(tot, off, 64);
        front += __shfl_down(front, off, 64);
    }
    // one atomic triple per WORKGROUP (four sampled rows): through LDS
    __shared__ int acc[3];
    if (threadIdx.x < 3) acc[threadIdx.x] = 0;
    __syncthreads();
    if (lane == 0 && tot > 0) {
        atomicAdd(&acc[0], near);
        atomicAdd(&acc[1], tot);
        atomicAdd(&acc[2], front);
    }
    __syncthreads();
    if (threadIdx.x == 0 && acc[1] > 0) {
        atomicAdd(&stats->near, acc[0]);
        atomicAdd(&stats->sampled, acc[1]);
        atomicAdd(&stats->front, acc[2]);
    }
}

// Longest row and the row-length histogram, ahead of the classification (the auto thresholds depend on them).  Grid-stride: a thread accumulates over its
// rows, a wave reduces once at the end -- one same-address atomic pair per wave and histogram level in all (round 5; one per wave of 64 ROWS was 73 us of
// the LONG_ROWS preprocess: 2 048 waves x 2 levels on one address each).
__global__ __launch_bounds__(kBlockThreads) void row_len_max(const int32_t *__restrict__ row_ptr, int32_t M, PlanStats *stats)
{
    unsigned cnt[kHistN];
    unsigned long long part[kHistN];
#pragma unroll
    for (int i = 0; i < kHistN; ++i) { cnt[i] = 0; part[i] = 0ull; }
    int m = 0;
    for (int64_t r = (int64_t)blockIdx.x * kBlockThreads + threadIdx.x; r < M; r += (int64_t)gridDim.x * kBlockThreads) {
        const int len = row_ptr[r + 1] - row_ptr[r];
        m = max(m, len);
#pragma unroll
        for (int i = 0; i < kHistN; ++i)
            if (len > hist_threshold(i)) { ++cnt[i]; part[i] += (unsigned long long)len; }
    }
#pragma unroll
    for (int i = 0; i < kHistN; ++i) {
        if (__ballot(cnt[i] != 0) == 0) break;                 // wave-uniform; a longer threshold has none either
        for (int off = 32; off > 0; off >>= 1) {
            cnt[i] += __shfl_down(cnt[i], off, 64);
            part[i] += __shfl_down(part[i], off, 64);
        }
        if ((threadIdx.x & 63) == 0) {
            atomicAdd(&stats->hist.cnt[i], cnt[i]);
            atomicAdd(&stats->hist.nnz[i], part[i]);
        }
    }
    for (int off = 32; off > 0; off >>= 1) m = max(m, __shfl_down(m, off, 64));
    if ((threadIdx.x & 63) == 0 && m > __hip_atomic_load(&stats->max_len, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
        atomicMax(&stats->max_len, m);
}

__global__ __launch_bounds__(kBlockThreads) void classify_rows(const int32_t *__restrict__ row_ptr, int32_t M,
                                                              const uint8_t *__restrict__ blk_flag, int32_t mthr_user,
                                                              int32_t mean_len, int32_t thr_user, int32_t clen, int32_t split,
                                                              int64_t nnz, int32_t K, int32_t N,
                                                              int32_t *__restrict__ seg_cnt,
                                                              int32_t *__restrict__ slot_cnt,
                                                              int32_t *__restrict__ long_cnt, PlanStats *stats)
{
    const int64_t r = (int64_t)blockIdx.x * kBlockThreads + threadIdx.x;
    int len = 0;
    unsigned bad = 0;
    // max_len and the histogram are complete (previous kernel); every thread resolves the same two thresholds
    const int thr = thr_user > 0 ? thr_user : resolve_hub_threshold(nnz, M, K, N, stats->hist.nnz, stats->max_len);
    const int local_pct = stats->sampled > 0 ? (int)(100.0 * stats->near / stats->sampled) : 0;      // (sample_locality ran before this kernel, same stream)
    const int mthr = resolve_medium_threshold(mthr_user, nnz, M, stats->max_len, thr, local_pct, N);
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
        if (segs == 0 || slots != 0) len = 0;                  // (below: only whole segments count)
        if (r == 0) { stats->ptr0 = beg; stats->mthr = mthr; stats->thr = thr; if (beg < 0) bad |= 2; }
        if (r == M - 1) stats->ptrM = end;
    } else if (r == M) {   // trailing zero so the exclusive scans leave the totals at index M
        seg_cnt[r] = 0;
        slot_cnt[r] = 0;
        long_cnt[r] = 0;
    }
    if (bad) atomicOr(&stats->bad, bad);   // malformed input only
    // nonzeros in whole segments, one atomic per wave that has any (r >= M: len = 0)
    unsigned long long part = (r < M) ? (unsigned long long)len : 0ull;
    for (int o = 32; o > 0; o >>= 1) part += __shfl_down(part, o, 64);
    if ((threadIdx.x & 63) == 0 && part) atomicAdd(&stats->seg_nnz, part);
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

int sample_columns_gpu(const int32_t *d_row_ptr, const int32_t *d_col_idx, int32_t M, int32_t K, int64_t nnz, void *d_scratch256, int32_t *local_pct, int32_t *front_pct)
{
    static_assert(sizeof(PlanStats) <= 192, "the caller's 256-byte scratch holds the statistics behind its first 64 bytes");
    *local_pct = 0;
    *front_pct = 0;
    if (M <= 0 || K <= 0 || nnz <= 0) return 0;
    PlanStats *stats = (PlanStats *)d_scratch256;
    hipLaunchKernelGGL(init_plan_stats, dim3(1), dim3(64), 0, 0, stats);
    const int n_samples = M < 8192 ? M : 8192;
    const int window = locality_window(K);
    hipLaunchKernelGGL(sample_locality, dim3((unsigned)(((size_t)n_samples * 64 + kBlockThreads - 1) / kBlockThreads)), dim3(kBlockThreads), 0, 0,
                       d_row_ptr, d_col_idx, M, K, nnz, n_samples, window, stats);
    PLAN_TRY(hipGetLastError());
    PlanStats host;
    PLAN_TRY(hipMemcpy(&host, stats, sizeof(PlanStats), hipMemcpyDeviceToHost));
    if (host.sampled > 0) {
        *local_pct = (int32_t)(100.0 * host.near / host.sampled);
        *front_pct = (int32_t)(100.0 * host.front / host.sampled);
    }
    return 0;
}

int build_plan_gpu(const int32_t *d_row_ptr, const int32_t *d_col_idx, int32_t M, int32_t K, int32_t N, int64_t nnz,
                   const uint8_t *d_blk_flag, const unsigned int *d_col_bad, int32_t mthr, int32_t thr_user, int32_t clen, int32_t split, int32_t seg_order,
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
    hipLaunchKernelGGL(row_len_max, dim3(grid < 1024u ? grid : 1024u), dim3(kBlockThreads), 0, 0, d_row_ptr, M, stats);
    if (nnz > 0 && K > 0) {
        const int n_samples = M < 8192 ? M : 8192;
        const int window = locality_window(K);
        hipLaunchKernelGGL(sample_locality, dim3((unsigned)(((size_t)n_samples * 64 + kBlockThreads - 1) / kBlockThreads)), dim3(kBlockThreads), 0, 0,
                           d_row_ptr, d_col_idx, M, K, nnz, n_samples, window, stats);
    }
    hipLaunchKernelGGL(classify_rows, dim3(grid), dim3(kBlockThreads), 0, 0, d_row_ptr, M, d_blk_flag, mthr,
                       (int32_t)(nnz / M), thr_user, clen, split, nnz, K, N, seg_cnt, slot_cnt, long_cnt, stats);
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
    out->front_pct = host.st.sampled > 0 ? (int32_t)(100.0 * host.st.front / host.st.sampled) : 0;
    out->n_chunks = host.n_chunks;
    out->n_slots = host.n_slots;
    out->n_long = host.n_long;
    out->n_medium = host.n_chunks - host.n_slots;
    out->seg_nnz = (int64_t)host.st.seg_nnz;
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
        // segment order: by length (the lane groups of a wave carry similar lengths, the tail is short; auto) or, "segment_order" = 2, as the rows come.  Row
        // order keeps neighbouring rows -- which gather the same B rows where the columns are local -- together like the rows kernel does; measured on the
        // structured graphs it is a mixed bag (banded long rows without strips 0.55 of the time, protein-unsorted 0.89; but 1.1 - 1.3x at kLen 32 on every
        // community order and with strips in force: profiles/r05_regret.md), and what it gains the medium rule gets by keeping the rows in the rows kernel.
        const bool by_rows = seg_order == 2;
        if (n > 0 && by_rows) PLAN_TRY(hipMemcpyAsync(out->d_chunks, unsorted, n * sizeof(Chunk), hipMemcpyDeviceToDevice, 0));      // row order: as emitted
        else if (n > 0) PLAN_TRY(hipcub::DeviceRadixSort::SortPairsDescending(tmp2, sbytes, keys_in, keys_out, unsorted, out->d_chunks, (int)n, 0, end_bit));
        if (nl > 0) PLAN_TRY(hipcub::DeviceRadixSort::SortPairsDescending(tmp2, lbytes, lkeys_in, lkeys_out, longs_unsorted, out->d_long, (int)nl, 0, end_bit));
    }
    PLAN_TRY(hipStreamSynchronize(0));
    return MI_SPMM_OK;
}


// ---- block path: item assembly on the device -------------------------------------------------------------------------
// Sort key of a piece, 64 bits, ascending: pass (3) | run piece (1: list pieces first) | first column (31) | not shareable (1) |
// 2^28 - 1 - min(len, 2^28 - 1) (28: longest first).  Ties keep the emission order (group, then ordinal): the sort is stable.
// Slots past a group's last piece carry ~0 and sort to the end.
namespace {
constexpr unsigned long long kNoPiece = ~0ull;
constexpr int kLenBits = 28;

__global__ __launch_bounds__(kBlockThreads) void emit_piece_keys(const GroupPieces *__restrict__ gp, int32_t ng, int32_t share,
                                                                int32_t run_unit, unsigned long long *__restrict__ keys,
                                                                uint32_t *__restrict__ vals)
{
    const int64_t t = (int64_t)blockIdx.x * kBlockThreads + threadIdx.x;
    if (t >= (int64_t)ng * kMaxPieces) return;
    const int gi = (int)(t / kMaxPieces), ord = (int)(t % kMaxPieces);
    const GroupPieces g = gp[gi];
    unsigned long long key = kNoPiece;
    if (ord < g.n) {
        const int32_t c = g.c0[ord], len = g.len[ord];
        const bool run = c >= 0 && len % run_unit == 0;          // any other run the run kernels cannot take: a list piece
        const uint32_t col = (uint32_t)(c >= 0 ? c : -1 - c);
        const bool shareable = run && (len % kShareLenUnit) == 0 && share > 1;
        const uint32_t inv = (uint32_t)((1 << kLenBits) - 1 - (len < (1 << kLenBits) - 1 ? len : (1 << kLenBits) - 1));
        key = ((unsigned long long)ord << 61) | ((unsigned long long)(run ? 1 : 0) << 60) | ((unsigned long long)col << 29) |
              ((unsigned long long)(shareable ? 0 : 1) << kLenBits) | inv;
    }
    keys[t] = key;
    vals[t] = (uint32_t)t;        // = gi * kMaxPieces + ord
}

// head[i] = i where piece i opens a new (pass, class, column, shareable) run of the sorted list, else 0 (for a max-scan)
__global__ __launch_bounds__(kBlockThreads) void mark_run_heads(const unsigned long long *__restrict__ keys, int32_t n, int32_t *__restrict__ head)
{
    const int i = (int)(blockIdx.x * (unsigned)kBlockThreads + threadIdx.x);
    if (i >= n) return;
    const unsigned long long k = keys[i];
    head[i] = (i == 0 || (k >> kLenBits) != (keys[i - 1] >> kLenBits)) ? i : 0;
}

// item_head[i] = 1 where sorted piece i starts an item: every unshareable piece, and every share-th piece of a shareable run
__global__ __launch_bounds__(kBlockThreads) void mark_item_heads(const unsigned long long *__restrict__ keys, const int32_t *__restrict__ run_start,
                                                                int32_t n, int32_t share, int32_t *__restrict__ item_head)
{
    const int i = (int)(blockIdx.x * (unsigned)kBlockThreads + threadIdx.x);
    if (i > n) return;
    if (i == n) { item_head[i] = 0; return; }           // trailing zero: the exclusive sum leaves the total at index n
    const unsigned long long k = keys[i];
    const bool valid = k != kNoPiece;
    const bool shareable = ((k >> kLenBits) & 1ull) == 0 && ((k >> 60) & 1ull) == 1;
    item_head[i] = valid && (!shareable || (i - run_start[i]) % share == 0) ? 1 : 0;
}

// Written without same-address atomics (57 K items adding into four words took 1.1 of the assembly's 1.5 ms): the sorted
// list is bucketed by (pass, class), so a bucket's first item is the one whose predecessor piece belongs to another bucket --
// one writer per word; a bucket's size is the distance to the next bucket's first item (host).  Only the shared-item count is
// a sum: one atomic per wave.
struct BlockCounters {
    int32_t first[kMaxPieces][2];     // item index of the bucket's first item; -1: empty bucket
    int32_t n_shared, n_pieces;
};

__global__ void init_block_counters(BlockCounters *c)
{
    if (threadIdx.x == 0) {
        for (int p = 0; p < kMaxPieces; ++p)
            for (int k = 0; k < 2; ++k) c->first[p][k] = -1;
        c->n_shared = 0;
        c->n_pieces = 0;
    }
}

__global__ __launch_bounds__(kBlockThreads) void write_block_items(const unsigned long long *__restrict__ keys, const uint32_t *__restrict__ vals,
                                                                  const int32_t *__restrict__ item_head, const int32_t *__restrict__ item_idx,
                                                                  int32_t n, int32_t share, const GroupPieces *__restrict__ gp,
                                                                  const int32_t *__restrict__ groups, BlockItem *__restrict__ items,
                                                                  BlockCounters *__restrict__ counters)
{
    const int i = (int)(blockIdx.x * (unsigned)kBlockThreads + threadIdx.x);
    const bool is_head = i < n && item_head[i];
    int m_shared = 0;
    if (i < n && keys[i] != kNoPiece && (i + 1 == n || keys[i + 1] == kNoPiece)) counters->n_pieces = i + 1;   // the last piece: one writer
    if (is_head) {
    const unsigned long long k = keys[i];
    int m = 1;
    while (m < share && i + m < n && !item_head[i + m] && keys[i + m] != kNoPiece && (keys[i + m] >> kLenBits) == (k >> kLenBits)) ++m;
    BlockItem it;
    it.m = m;
    const bool run = ((k >> 60) & 1ull) != 0;
    const int32_t col = (int32_t)((k >> 29) & 0x7fffffffull);
    it.c0 = run ? col : -1 - col;
    for (int q = 0; q < kMaxShare; ++q) {
        BlockPiece p = {0, 0, 0, 0, 0, 0};
        if (q < m) {
            const uint32_t v = vals[i + q];
            const int gi = (int)(v / kMaxPieces), ord = (int)(v % kMaxPieces);
            const GroupPieces g = gp[gi];
            p.group = groups[gi];
            p.k0 = g.k0[ord];
            p.len = g.len[ord];
            p.flags = (ord > 0 ? kPieceCarryIn : 0) | (ord + 1 < g.n ? kPieceCarryOut : 0);
            p.p0 = g.p0;
            p.row_len = g.row_len;
        }
        it.p[q] = p;
    }
    it.pad[0] = it.pad[1] = 0;
    const int ii = item_idx[i];
    items[ii] = it;
    const int pass = (int)(k >> 61), cls = run ? 1 : 0;
    if (i == 0 || (keys[i - 1] >> 60) != (k >> 60)) counters->first[pass][cls] = ii;     // (pass, class) = the key's top four bits
    m_shared = m > 1 ? 1 : 0;
    }
    const int ws = __builtin_popcountll(__ballot(m_shared != 0));
    if (ws > 0 && (threadIdx.x & 63) == 0) atomicAdd(&counters->n_shared, ws);
}

struct MaxOp { __device__ __forceinline__ int32_t operator()(int32_t a, int32_t b) const { return a > b ? a : b; } };
}  // namespace

int build_block_items_gpu(const GroupPieces *d_gp, const int32_t *d_groups, int32_t ng, int32_t share, int32_t run_unit, Scratch *sc,
                          BlockPlanOut *out)
{
    BlockItem *const keep = out->d_items;
    const size_t keep_cap = out->items_cap;
    *out = BlockPlanOut();
    out->d_items = keep;
    out->items_cap = keep_cap;
    for (int p = 0; p < kMaxPieces; ++p) out->launch[p][0].off = out->launch[p][0].n = out->launch[p][1].off = out->launch[p][1].n = 0;
    if (ng <= 0) return MI_SPMM_OK;
    const int32_t n = ng * kMaxPieces;              // slots; a group's unused ordinals sort to the end
    if ((int64_t)ng * kMaxPieces > INT32_MAX) return MI_SPMM_EUNSUPPORTED;
    size_t sort_b = 0, scan_b = 0, sum_b = 0;
    PLAN_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, sort_b, (unsigned long long *)nullptr, (unsigned long long *)nullptr, (uint32_t *)nullptr,
                                                (uint32_t *)nullptr, n));
    PLAN_TRY(hipcub::DeviceScan::InclusiveScan(nullptr, scan_b, (int32_t *)nullptr, (int32_t *)nullptr, MaxOp(), n));
    PLAN_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, sum_b, (int32_t *)nullptr, (int32_t *)nullptr, n + 1));
    unsigned long long *k_in = nullptr, *k_out = nullptr;
    uint32_t *v_in = nullptr, *v_out = nullptr;
    int32_t *head = nullptr, *run_start = nullptr, *item_head = nullptr, *item_idx = nullptr;
    BlockItem *items = nullptr;
    BlockCounters *counters = nullptr;
    char *tmp = nullptr;
    size_t tmp_b = sort_b > scan_b ? sort_b : scan_b;
    if (sum_b > tmp_b) tmp_b = sum_b;
    auto layout = [&](Carver &c) {
        k_in = c.take<unsigned long long>((size_t)n);
        k_out = c.take<unsigned long long>((size_t)n);
        v_in = c.take<uint32_t>((size_t)n);
        v_out = c.take<uint32_t>((size_t)n);
        head = c.take<int32_t>((size_t)n);
        run_start = c.take<int32_t>((size_t)n);
        item_head = c.take<int32_t>((size_t)n + 1);
        item_idx = c.take<int32_t>((size_t)n + 1);
        items = c.take<BlockItem>((size_t)n);
        counters = c.take<BlockCounters>(1);
        tmp = c.take<char>(tmp_b);
    };
    {
        Carver dry(nullptr);
        layout(dry);
        const int rc = scratch_reserve(sc, dry.off + 256);
        if (rc != 0) return rc;
        Carver real(sc->p);
        layout(real);
    }
    const unsigned grid = (unsigned)(((size_t)n + 1 + kBlockThreads - 1) / kBlockThreads);
    hipLaunchKernelGGL(init_block_counters, dim3(1), dim3(64), 0, 0, counters);
    hipLaunchKernelGGL(emit_piece_keys, dim3(grid), dim3(kBlockThreads), 0, 0, d_gp, ng, share, run_unit, k_in, v_in);
    PLAN_TRY(hipGetLastError());
    size_t t = sort_b;
    PLAN_TRY(hipcub::DeviceRadixSort::SortPairs(tmp, t, k_in, k_out, v_in, v_out, n));
    hipLaunchKernelGGL(mark_run_heads, dim3(grid), dim3(kBlockThreads), 0, 0, k_out, n, head);
    t = scan_b;
    PLAN_TRY(hipcub::DeviceScan::InclusiveScan(tmp, t, head, run_start, MaxOp(), n));
    hipLaunchKernelGGL(mark_item_heads, dim3(grid), dim3(kBlockThreads), 0, 0, k_out, run_start, n, share, item_head);
    t = sum_b;
    PLAN_TRY(hipcub::DeviceScan::ExclusiveSum(tmp, t, item_head, item_idx, n + 1));
    hipLaunchKernelGGL(write_block_items, dim3(grid), dim3(kBlockThreads), 0, 0, k_out, v_out, item_head, item_idx, n, share, d_gp, d_groups,
                       items, counters);
    PLAN_TRY(hipGetLastError());
    struct { BlockCounters c; int32_t n_items; } host;
    PLAN_TRY(hipMemcpyAsync(&host.c, counters, sizeof(BlockCounters), hipMemcpyDeviceToHost, 0));
    PLAN_TRY(hipMemcpyAsync(&host.n_items, item_idx + n, sizeof(int32_t), hipMemcpyDeviceToHost, 0));
    PLAN_TRY(hipStreamSynchronize(0));
    out->n_items = host.n_items;
    out->n_pieces = host.c.n_pieces;
    out->n_shared = host.c.n_shared;
    int32_t next_first = host.n_items;           // buckets are laid out in (pass, class) order: walk them backwards
    for (int p = kMaxPieces - 1; p >= 0; --p)
        for (int k = 1; k >= 0; --k) {
            const int32_t f = host.c.first[p][k];
            if (f < 0) continue;
            out->launch[p][k].off = f;
            out->launch[p][k].n = next_first - f;
            next_first = f;
            if (p + 1 > out->n_passes) out->n_passes = p + 1;
        }
    if (host.n_items > 0) {
        if (out->items_cap < (size_t)host.n_items) {       // grow-only, like the arenas: a repeated preprocess allocates nothing
            if (out->d_items) (void)hipFree(out->d_items);
            out->d_items = nullptr;
            out->items_cap = 0;
            PLAN_TRY(hipMalloc((void **)&out->d_items, (size_t)host.n_items * sizeof(BlockItem)));
            out->items_cap = (size_t)host.n_items;
        }
        PLAN_TRY(hipMemcpy(out->d_items, items, (size_t)host.n_items * sizeof(BlockItem), hipMemcpyDeviceToDevice));
    }
    return MI_SPMM_OK;
}

// ---- column strips of the exact segments (plan.hpp) ---------------------------------------------------------------------------------
namespace {

// one wave per segment: are its columns ascending (equal neighbours allowed), and how long is it
__global__ __launch_bounds__(kBlockThreads) void survey_segments_kernel(const Chunk *__restrict__ chunks, int32_t n, const int32_t *__restrict__ col_idx,
                                                                       SegmentSurvey *out)
{
    const int wave = (int)((blockIdx.x * (unsigned)kBlockThreads + threadIdx.x) >> 6), lane = (int)(threadIdx.x & 63);
    if (wave >= n) return;
    const Chunk c = chunks[wave];
    int bad = 0;
    for (int i = c.beg + lane; i + 1 < c.end; i += 64) bad |= col_idx[i] > col_idx[i + 1];
    bad = __any(bad);
    if (lane == 0) {
        if (bad) atomicAdd(&out->unsorted, 1u);
        atomicAdd(&out->nnz, (unsigned long long)(c.end - c.beg));
    }
}

// one thread per (segment, strip): the first nonzero of the segment whose column is >= the strip's first column, and the same for the next strip
__global__ __launch_bounds__(kBlockThreads) void build_col_strips_kernel(const Chunk *__restrict__ chunks, int32_t n, const int32_t *__restrict__ col_idx,
                                                                        int32_t K, int32_t S, Chunk *__restrict__ strips)
{
    const int64_t t = (int64_t)blockIdx.x * kBlockThreads + threadIdx.x;
    if (t >= (int64_t)n * S) return;
    const int s = (int)(t / n), ch = (int)(t % n);
    const Chunk c = chunks[ch];
    auto lower_bound = [&](int32_t col) {
        int lo = c.beg, hi = c.end;
        while (lo < hi) {
            const int mid = lo + ((hi - lo) >> 1);
            if (col_idx[mid] < col) lo = mid + 1; else hi = mid;
        }
        return lo;
    };
    Chunk o;
    o.beg = s == 0 ? c.beg : lower_bound((int32_t)((int64_t)K * s / S));
    o.end = s == S - 1 ? c.end : lower_bound((int32_t)((int64_t)K * (s + 1) / S));
    o.slot = s == 0 ? -1 : kSlotContinue;
    o.row = c.row;
    strips[(size_t)s * (size_t)n + (size_t)ch] = o;
}

// ---- round 5: survey and tables in ONE pass over the segments' columns -------------------------------------------------------------
// The two kernels above read every segment's columns twice -- once lane-strided with one same-address atomic pair per segment (132 534 segments:
// 1.6 ms, the atomics), once through S binary searches per segment -- and cost more than a step of the graphs they serve.
// Here a wave strides a segment's columns once, coalesced; a lane compares its column with its left neighbour's (ascending?) and, where the two lie in
// different strips, that position is where the strips in between begin -- exactly lower_bound(first column of the strip).  S comes from the plan's own
// statistics (PlanStats::seg_nnz), so nothing has to come back from the device before the tables are written; one counter does afterwards.
// Grid-stride over the segments (longest first in the table, so neighbours in a workgroup carry similar lengths): one atomic per WAVE, and only if it
// met an unsorted segment.  The tables of a plan with an unsorted segment are incomplete and never used (plan_col_strips drops them).
__global__ __launch_bounds__(kBlockThreads) void strip_segments_kernel(const Chunk *__restrict__ chunks, int32_t n, const int32_t *__restrict__ col_idx,
                                                                      int32_t K, int32_t S, Chunk *__restrict__ strips, SegmentSurvey *out)
{
    __shared__ int32_t bound[kMaxColStrips + 2];      // bound[s]: first column of strip s; bound[S] = K
    for (int t = (int)threadIdx.x; t <= S; t += kBlockThreads) bound[t] = (int32_t)((int64_t)K * t / S);
    __syncthreads();
    const int lane = (int)(threadIdx.x & 63);
    const int wave = (int)(blockIdx.x * (unsigned)(kBlockThreads / 64) + (threadIdx.x >> 6)), n_waves = (int)(gridDim.x * (unsigned)(kBlockThreads / 64));
    const float scale = (float)S / (float)K;
    auto strip_of = [&](int32_t col) {                // the strip that holds column col (0 <= col < K); the float guess is off by one at most
        int s = (int)((float)col * scale);
        s = s < 0 ? 0 : (s > S - 1 ? S - 1 : s);
        while (s + 1 < S && bound[s + 1] <= col) ++s;
        while (s > 0 && bound[s] > col) --s;
        return s;
    };
    unsigned bad_segments = 0;
    for (int ch = wave; ch < n; ch += n_waves) {
        const Chunk c = chunks[ch];
        if (lane < S) {                                // what does not depend on the columns
            Chunk *o = strips + (size_t)lane * (size_t)n + (size_t)ch;
            o->slot = lane == 0 ? -1 : kSlotContinue;
            o->row = c.row;
            if (lane == 0) o->beg = c.beg;
        }
        int bad = 0;
        int32_t carry = -1;                            // column in front of this batch ("-1" lies in strip 0: strip 0 begins at c.beg)
        // positions c.beg .. c.end INCLUSIVE: the virtual element at c.end sits past the last strip and closes every strip still open
        for (int base = c.beg; base <= c.end; base += 64) {
            const int i = base + lane;
            const int32_t col = i < c.end ? col_idx[i] : INT32_MAX;
            int32_t prev = __shfl_up(col, 1, 64);
            if (lane == 0) prev = carry;
            carry = __shfl(col, 63, 64);
            if (i <= c.end) {
                bad |= col < prev;
                const int sp = prev < 0 ? 0 : strip_of(prev);
                const int sc = i < c.end ? strip_of(col) : S;
                for (int s = sp + 1; s <= sc; ++s) {   // strips sp+1 .. sc begin here, strips sp .. sc-1 end here (no trip at all for most lanes)
                    if (s < S) strips[(size_t)s * (size_t)n + (size_t)ch].beg = i;
                    strips[(size_t)(s - 1) * (size_t)n + (size_t)ch].end = i;
                }
            }
        }
        if (__any(bad)) ++bad_segments;
    }
    if (lane == 0 && bad_segments) atomicAdd(&out->unsorted, bad_segments);
}

}  // namespace

int strip_segments(const Chunk *d_chunks, int32_t n_chunks, const int32_t *d_col_idx, int32_t K, int32_t S, Chunk *d_strips, void *d_scratch256,
                   SegmentSurvey *out)
{
    *out = SegmentSurvey{};
    if (n_chunks <= 0 || S < 2 || S > kMaxColStrips) return 0;
    SegmentSurvey *d = (SegmentSurvey *)d_scratch256;
    hipError_t e = hipMemsetAsync(d, 0, sizeof(SegmentSurvey), 0);
    if (e != hipSuccess) return (int)e;
    // enough waves to fill the chip eight deep, never more than there are segments
    int64_t blocks = ((int64_t)n_chunks + kBlockThreads / 64 - 1) / (kBlockThreads / 64);
    if (blocks > 256 * 8) blocks = 256 * 8;
    hipLaunchKernelGGL(strip_segments_kernel, dim3((unsigned)blocks), dim3(kBlockThreads), 0, 0, d_chunks, n_chunks, d_col_idx, K, S, d_strips, d);
    e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
    e = hipMemcpy(out, d, sizeof(SegmentSurvey), hipMemcpyDeviceToHost);
    return e == hipSuccess ? 0 : (int)e;
}

int survey_segments(const Chunk *d_chunks, int32_t n_chunks, const int32_t *d_col_idx, void *d_scratch256, SegmentSurvey *out)
{
    *out = SegmentSurvey{};
    if (n_chunks <= 0) return 0;
    SegmentSurvey *d = (SegmentSurvey *)d_scratch256;     // 256 bytes of device memory the caller owns
    hipError_t e = hipMemsetAsync(d, 0, sizeof(SegmentSurvey), 0);
    if (e != hipSuccess) return (int)e;
    const int64_t blocks = ((int64_t)n_chunks * 64 + kBlockThreads - 1) / kBlockThreads;
    hipLaunchKernelGGL(survey_segments_kernel, dim3((unsigned)blocks), dim3(kBlockThreads), 0, 0, d_chunks, n_chunks, d_col_idx, d);
    e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
    e = hipMemcpy(out, d, sizeof(SegmentSurvey), hipMemcpyDeviceToHost);
    return e == hipSuccess ? 0 : (int)e;
}

int build_col_strips(const Chunk *d_chunks, int32_t n_chunks, const int32_t *d_col_idx, int32_t K, int32_t S, Chunk *d_strips)
{
    if (n_chunks <= 0 || S < 2) return 0;
    const int64_t blocks = ((int64_t)n_chunks * S + kBlockThreads - 1) / kBlockThreads;
    hipLaunchKernelGGL(build_col_strips_kernel, dim3((unsigned)blocks), dim3(kBlockThreads), 0, 0, d_chunks, n_chunks, d_col_idx, K, S, d_strips);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}

}  // namespace mi
