// mi_spmm.hip -- C ABI (include/mi_spmm.h) over the gfx950 kernels.
//
// Host side of the drop-in: what SpMMOpt::preprocess / SpMMOpt::run do in the
// reference (PA4/workspace/src/spmm_opt.cu:37-75), re-designed for MI355X:
//   preprocess: validate the CSR, detect block groups, classify rows and build the
//               segment table on the device (preprocess_gpu.hip; the reference-style
//               host loop stays behind gpu_preprocess=0), allocate the partial-sum
//               workspace (handle-owned);
//   run:        rows kernel (+ segments + reduce + blocks where such rows exist) on
//               the caller's stream; overwrite semantics; no host sync.
#include "../../include/mi_spmm.h"
#include "plan.hpp"
#include "spmm_kernels.hpp"

#include <algorithm>
#include <chrono>
#include <climits>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

using namespace mi;

#define HIP_TRY(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return (int)e_; } while (0)

struct mi_spmm_handle {
    uint32_t magic;
    const int32_t *d_ptr;
    const int32_t *d_idx;
    const float *d_val;
    int32_t num_v, num_cols, feat;
    int64_t nnz;
    // options
    int64_t long_thr_user;  // what the caller asked for (0 = auto); long_thr holds the resolved value
    int64_t medium_thr;  // rows longer than this (and not split) run as one exact segment in the segment kernel; 0 = auto
    int64_t medium_res;  // the value in force after preprocess (auto resolved, capped by long_thr)
    int64_t split_long;   // 0 (default): hubs keep their stored order (spmm_hub); 1: hubs are cut into pieces summed piece by piece
    int64_t hub_slice;    // columns per hub wave: 16, 32, 64; 0 = auto
    int64_t hub_overlap;  // 1 (default): the hub kernel and the segment kernel run on handle-owned side streams, forked from and joined
                          // back into the caller's stream inside every run call (their longest rows then overlap the rows kernel)
    hipStream_t side[2];  // [0]: hub kernel, [1]: segment (+ reduce) kernels; created by the first preprocess that needs them
    int32_t side_overlaps[2]; // concurrent_stream's verdict on each side stream (1: its kernels run beside the null stream's)
    int64_t side_priority; // bit 0: the hub stream is a high-priority stream, bit 1: the segment stream is
    int64_t ftz;          // "flush_denormals": fp32 subnormals flushed like the reference's nvcc --use_fast_math build (spmm_kernels.hpp apply_ftz)
    int64_t segment_overlap; // 1: the segment kernel may go to side stream 1; 0: it stays on the caller's stream, in front of the rows kernel
    hipEvent_t ev_fork, ev_join[2];
    bool fork_recorded, forked[2];   // state of the current run_rows call
    bool overlap_on[2];   // resolved by preprocess: is the fork/join (~20 us of launch latency per call) worth it for this class
    int64_t long_thr, long_chunk, rows_per_block, xcd_remap, nt_store, nt_stream, block_path;
    int64_t block_threads;  // v2 workgroup size: 64, 128 or 256
    int64_t gpu_preprocess;  // 1: segment table built on the device (default); 0: reference-style host loop
    int64_t kernel;  // 2 = spmm_rows_v2 (pipelined items); 1 named the first-generation kernel, retired in round 3
    int64_t split_cols;  // 1 (default): columns past the last full 256-column tile get their own launches
    int64_t segment_unroll;  // B-row gathers in flight per lane group in the segment kernel: 8, 16 or 32 (default)
    int64_t tile_cols;       // widest column tile of the rows/segment kernels: 256 (whole wave on a row), 128, 64, 32; 0 = auto
    int64_t col_strips;      // column strips of the exact segments (plan.hpp): 0 = auto, 1 = off, S >= 2 = that many (if the segments' columns ascend)
    Chunk *d_strips;         // [n_strips][n_chunks] sub-segments, strip-major; grow-only buffer (capacity strips_cap sub-segments), kept across preprocess calls
    size_t strips_cap;
    int32_t n_strips;        // strips in force (1 = none)
    int32_t seg_unsorted;    // segments whose columns do not ascend (-1: not looked at)
    int64_t seg_nnz;         // nonzeros in the whole segments (either plan builder): the strip rule's input, known before any column is read
    int64_t strips_builder;  // 0 (default): strip_segments, one pass; 1: the round-4 pair survey_segments + build_col_strips (cross-check)
    int64_t seg_order;       // "segment_order": 0 = auto, 1 = segments longest first, 2 = in row order
    int64_t rows_unroll;     // "rows_unroll": B-row gathers in flight per lane group in the rows kernel: 0 = auto, 8, 16
    int64_t fused_step;      // "fused_step": 2 (default) = auto, 0 = never, 1 = whenever the step is eligible: hub rows, segments and short rows as the three
                             // roles of ONE launch (spmm_kernels.hpp spmm_small_step) instead of 2-3 launches and a side-stream fork / join
    int32_t last_fused;      // 1: the last run call went through the small-step kernel
    int64_t fused_order;     // "fused_order": 0 (default) = auto, 1 = hub workgroups first in the small-step grid, 2 = segment workgroups first
    int32_t last_seg_first;  // 1: the last small-step launch put its segment workgroups first
    int32_t last_rows_deep;  // 1: the last rows launch kept 16 gathers in flight per lane group
    // "autotune" (round 5): the rules above are guesses from a 8 192-row sample and a histogram; a wrong guess is silent (same bits, slower).  With the option on,
    // preprocess MEASURES: the step is timed on the buffers it is given (the reference's preprocess touches vout too: spmm_opt.cu:67) under the auto plan and
    // under a handful of forced settings of the options the caller left to us -- tile width, strip count, medium threshold, one launch or several -- and the
    // fastest is kept.  Scheduling only: whatever wins gives the same bits.
    int64_t autotune;
    uint32_t tuned_mask;     // options the tuner set (bit 0 tile_cols, 1 col_strips, 2 medium_thr, 3 fused_step, 4 segment_order, 5 rows_unroll, 6 long_row_threshold): back to auto before the next tuning
    int32_t tune_evals;      // candidate settings timed by the last preprocess
    double tune_auto_ms, tune_best_ms;
    // plan
    bool prepared;
    Chunk *d_chunks;
    LongRow *d_long;
    float *d_partials;
    int32_t n_chunks, n_long, n_medium, n_slots;
    int64_t ldp;
    size_t ws_bytes;
    int32_t max_row_nnz;
    int32_t local_pct;   // column locality sample (percent); -1 = not measured
    int32_t front_pct;   // sampled nonzeros in the first quarter of the columns (percent; uniform columns: 25)
    double preprocess_us;
    double phase_us[5];  // d2h row_ptr + validate, column check, block detection, host segment table, upload
    int32_t last_lpr, last_v, last_launches, last_wide;
    // block (MFMA) path
    int64_t block_min_len;
    uint8_t *d_blk_flag;
    int32_t *d_blk_groups;
    int32_t n_blk_groups;
    int64_t n_rows_for_rows_kernel;  // rows neither split nor owned by the block path
    // block path, run time: pieces of the groups' column lists, grouped into items, per pass (spmm_kernels.hpp)
    int64_t block_share;       // most pieces per item (1 = no sharing; default 2)
    int64_t block_max_pieces;  // most pieces a group's list is cut into = most passes (1 = never cut)
    int64_t block_wg_waves;    // waves (= items) per workgroup of the block kernels: 1 (default), 2, 4
    int64_t block_run_min;     // shortest run worth a piece of its own
    BlockItem *d_blk_items;    // grow-only (capacity blk_items_cap items): kept across preprocess calls, released by destroy
    size_t blk_items_cap;
    int32_t n_blk_items, n_blk_pieces, n_blk_passes, n_blk_shared_items;
    // preprocess temporaries (grow-only, kept across preprocess calls, released by destroy)
    Scratch scratch_a, scratch_b;
    unsigned int *d_col_bad;   // 256 bytes: the column-range flag
    struct { int32_t off, n; } blk_launch[kMaxPieces][3];   // [pass][0: list items, 1: run items of one piece, 2: shared run items]
    std::vector<int32_t> *hub_rows_sorted;   // host copy of the hub rows, ascending (null: unknown -> every call launches the hub kernel)
};

static const uint32_t kMagic = 0x4d49534du;  // "MISM"
// Pieces of a split row: the plan builders step through a row in int32; a piece length far beyond any row
// that can be split usefully would overflow `b + clen`.
static const int64_t kMaxLongChunk = 1 << 20;

static bool good(const mi_spmm_handle *h) { return h && h->magic == kMagic; }

static void free_plan(mi_spmm_handle *h)
{
    if (h->d_chunks) (void)hipFree(h->d_chunks);
    h->n_strips = 1;         // (the strip buffer itself is grow-only: released by destroy)
    h->seg_unsorted = -1;
    h->seg_nnz = 0;
    if (h->d_long) (void)hipFree(h->d_long);
    if (h->d_partials) (void)hipFree(h->d_partials);
    if (h->d_blk_flag) (void)hipFree(h->d_blk_flag);
    if (h->d_blk_groups) (void)hipFree(h->d_blk_groups);
    h->n_blk_items = h->n_blk_pieces = h->n_blk_passes = h->n_blk_shared_items = 0;
    std::memset(h->blk_launch, 0, sizeof(h->blk_launch));
    h->d_blk_flag = nullptr;
    h->d_blk_groups = nullptr;
    h->n_blk_groups = 0;
    h->d_chunks = nullptr;
    h->d_long = nullptr;
    h->d_partials = nullptr;
    h->n_chunks = h->n_long = h->n_medium = h->n_slots = 0;
    delete h->hub_rows_sorted;
    h->hub_rows_sorted = nullptr;
    h->local_pct = -1;
    h->front_pct = 25;
    h->ws_bytes = 0;
    h->prepared = false;
}

// The block kernel works on column slabs of 256/128/64/32 columns (blockIdx.y walks them): any N that is
// a multiple of 32 qualifies, with the widest slab that divides it.
static int block_slab_width(int32_t N) { return N <= 0 ? 0 : (N % 256 == 0 ? 256 : N % 128 == 0 ? 128 : N % 64 == 0 ? 64 : N % 32 == 0 ? 32 : 0); }
static bool block_path_shape_ok(int32_t N) { return block_slab_width(N) != 0; }

// Block path, second half of preprocess: the qualifying groups (d_blk_groups, either plan builder) are cut into
// pieces on the device (analyze_group_runs: one wave per group, O(nnz / 16) reads); every pass's pieces are ordered by
// first column and formed into items: run pieces with the same first column share their B rows (longest first, at most
// block_share per item, every shared length a whole number of k batches so that no shared piece ends inside a batch).
// On the device by default (preprocess_gpu.hip build_block_items_gpu: one radix sort, two scans, one 80-byte copy back);
// the host assembler below -- the pieces come back, O(groups log groups) on one core: 8 ms for C4's 65 536 groups --
// stays behind "gpu_preprocess" = 0 as the cross-check (test_gpu_and_host_plan_builders_agree: same items, same launches).
static int build_block_items(mi_spmm_handle *h)
{
    const int32_t ng = h->n_blk_groups;
    if (ng <= 0) return MI_SPMM_OK;
    // the groups' pieces live in the first preprocess arena (build_plan_gpu is done with it): no allocation call here
    if (scratch_reserve(&h->scratch_a, (size_t)ng * sizeof(GroupPieces) + 256) != 0) return MI_SPMM_ENOMEM;
    GroupPieces *d_gp = reinterpret_cast<GroupPieces *>(h->scratch_a.p);
    hipLaunchKernelGGL(analyze_group_runs, dim3((unsigned)((ng + 3) / 4)), dim3(kBlockThreads), 0, 0, h->d_ptr, h->d_idx,
                       h->d_blk_groups, ng, (int32_t)h->block_max_pieces, (int32_t)h->block_run_min, d_gp);
    hipError_t e = hipGetLastError();
    // shared items need the two-piece kernels, which exist for 256- and 128-column slabs (N % 128 == 0)
    const int slab_w = block_slab_width(h->feat);
    const int share = slab_w >= 128 ? (int)h->block_share : 1;
    // the run kernels sweep whole trips (two k batches: 16 / 32 / 64 rows for 256- / 128- / narrower slabs) and fetch their A
    // operands 16 bytes at a time; a run of any other length goes through the list kernel (general lengths, dword A loads)
    const int run_unit = slab_w == 256 ? 16 : slab_w == 128 ? 32 : 64;
    if (e == hipSuccess && h->gpu_preprocess) {
        BlockPlanOut bo;
        bo.d_items = h->d_blk_items;
        bo.items_cap = h->blk_items_cap;
        const int rc = build_block_items_gpu(d_gp, h->d_blk_groups, ng, share, run_unit, &h->scratch_b, &bo);
        h->d_blk_items = bo.d_items;
        h->blk_items_cap = bo.items_cap;
        if (rc != MI_SPMM_OK) return rc;
        h->n_blk_items = bo.n_items;
        h->n_blk_pieces = bo.n_pieces;
        h->n_blk_passes = bo.n_passes;
        h->n_blk_shared_items = bo.n_shared;
        h->ws_bytes += (size_t)bo.n_items * sizeof(BlockItem);
        const int run_cls = share > 1 ? 2 : 1;     // run items of one and of two pieces share a launch where the two-piece kernels exist
        for (int pass = 0; pass < kMaxPieces; ++pass) {
            h->blk_launch[pass][0].off = bo.launch[pass][0].off;
            h->blk_launch[pass][0].n = bo.launch[pass][0].n;
            h->blk_launch[pass][run_cls].off = bo.launch[pass][1].off;
            h->blk_launch[pass][run_cls].n = bo.launch[pass][1].n;
        }
        return MI_SPMM_OK;
    }
    std::vector<GroupPieces> gp((size_t)ng);
    std::vector<int32_t> groups((size_t)ng);
    if (e == hipSuccess) e = hipMemcpy(gp.data(), d_gp, (size_t)ng * sizeof(GroupPieces), hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(groups.data(), h->d_blk_groups, (size_t)ng * sizeof(int32_t), hipMemcpyDeviceToHost);
    if (e != hipSuccess) return (int)e;

    struct Key { int32_t col, len, gi, ord; bool shareable, run; };
    struct Launch { int32_t off, n; };
    std::vector<BlockItem> items;
    int64_t n_pieces = 0;
    int32_t n_pass = 0, n_shared = 0;
    std::vector<Key> keys;
    std::vector<BlockItem> lists, singles, shared;

    auto piece_of = [&](const Key &k) {
        const GroupPieces &g = gp[(size_t)k.gi];
        BlockPiece p;
        p.group = groups[(size_t)k.gi];
        p.k0 = g.k0[k.ord];
        p.len = g.len[k.ord];
        p.flags = (k.ord > 0 ? kPieceCarryIn : 0) | (k.ord + 1 < g.n ? kPieceCarryOut : 0);
        p.p0 = g.p0;
        p.row_len = g.row_len;
        return p;
    };
    // one pass's keys -> its items, appended to `out` class by class (lists, singles, shared)
    auto form_items = [&](std::vector<Key> &ks, std::vector<BlockItem> &out, Launch (&launch)[3], bool count) {
        // by first column; among equals the shareable ones together, longest first; ties in group order (deterministic)
        std::sort(ks.begin(), ks.end(), [](const Key &x, const Key &y) {
            if (x.col != y.col) return x.col < y.col;
            if (x.shareable != y.shareable) return x.shareable > y.shareable;
            if (x.len != y.len) return x.len > y.len;
            return x.gi < y.gi;
        });
        lists.clear();
        singles.clear();
        shared.clear();
        size_t i = 0;
        while (i < ks.size()) {
            size_t j = i + 1;
            if (ks[i].shareable)
                while (j < ks.size() && j - i < (size_t)share && ks[j].shareable && ks[j].col == ks[i].col) ++j;
            BlockItem it;
            std::memset(&it, 0, sizeof(it));
            it.m = (int32_t)(j - i);
            it.c0 = ks[i].run ? ks[i].col : -1 - ks[i].col;
            for (size_t q = i; q < j; ++q) it.p[q - i] = piece_of(ks[q]);
            // run items of one and of two pieces go through the same launch where the two-piece kernels exist: one long
            // list in column order instead of two short ones (a short launch pays its tail on 256 CUs)
            (it.c0 < 0 ? lists : (it.m > 1 || share > 1) ? shared : singles).push_back(it);
            if (count) {
                n_pieces += it.m;
                if (it.m > 1) ++n_shared;
            }
            i = j;
        }
        std::vector<BlockItem> *cls[3] = {&lists, &singles, &shared};
        for (int c = 0; c < 3; ++c) {
            launch[c].off = (int32_t)out.size();
            launch[c].n = (int32_t)cls[c]->size();
            out.insert(out.end(), cls[c]->begin(), cls[c]->end());
        }
    };
    for (int pass = 0; pass < kMaxPieces; ++pass) {
        keys.clear();
        for (int32_t gi = 0; gi < ng; ++gi) {
            const GroupPieces &g = gp[(size_t)gi];
            for (int ord = 0; ord < g.n; ++ord) {
                if (ord != pass) continue;
                Key k;
                const int32_t c = g.c0[ord];
                k.run = c >= 0 && g.len[ord] % run_unit == 0;      // any other run the run kernels cannot take: a list piece
                k.col = c >= 0 ? c : -1 - c;
                k.len = g.len[ord];
                k.gi = gi;
                k.ord = ord;
                k.shareable = k.run && (g.len[ord] % kShareLenUnit) == 0 && share > 1;
                keys.push_back(k);
            }
        }
        if (keys.empty()) break;
        n_pass = pass + 1;
        Launch al[3];
        form_items(keys, items, al, true);
        for (int c = 0; c < 3; ++c) { h->blk_launch[pass][c].off = al[c].off; h->blk_launch[pass][c].n = al[c].n; }
    }
    if (items.empty()) return MI_SPMM_OK;

    auto upload = [&](void **dst, const void *src, size_t bytes) -> int {
        if (bytes == 0) return MI_SPMM_OK;
        if (hipMalloc(dst, bytes) != hipSuccess) return MI_SPMM_ENOMEM;
        const hipError_t ue = hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice);
        if (ue != hipSuccess) return (int)ue;
        h->ws_bytes += bytes;
        return MI_SPMM_OK;
    };
    if (h->blk_items_cap < items.size()) {
        if (h->d_blk_items) (void)hipFree(h->d_blk_items);
        h->d_blk_items = nullptr;
        h->blk_items_cap = 0;
    }
    int rc = MI_SPMM_OK;
    if (h->d_blk_items) {
        const hipError_t ue = hipMemcpy(h->d_blk_items, items.data(), items.size() * sizeof(BlockItem), hipMemcpyHostToDevice);
        if (ue != hipSuccess) rc = (int)ue;
        else h->ws_bytes += items.size() * sizeof(BlockItem);
    } else {
        rc = upload((void **)&h->d_blk_items, items.data(), items.size() * sizeof(BlockItem));
        if (rc == MI_SPMM_OK) h->blk_items_cap = items.size();
    }
    if (rc != MI_SPMM_OK) return rc;
    h->n_blk_items = (int32_t)items.size();
    h->n_blk_pieces = (int32_t)n_pieces;
    h->n_blk_passes = n_pass;
    h->n_blk_shared_items = n_shared;
    return MI_SPMM_OK;
}

// Host copy of the hub rows in row order: a run_rows call (a row panel of the multi-GPU step) whose range holds no hub row
// skips the hub launch and its side-stream fork (~20 us) instead of launching a grid whose workgroups all leave at once.
static int note_hub_rows(mi_spmm_handle *h)
{
    delete h->hub_rows_sorted;
    h->hub_rows_sorted = nullptr;
    if (h->n_long <= 0 || h->split_long || h->n_long > (1 << 16)) return MI_SPMM_OK;
    std::vector<LongRow> lr((size_t)h->n_long);
    HIP_TRY(hipMemcpy(lr.data(), h->d_long, lr.size() * sizeof(LongRow), hipMemcpyDeviceToHost));
    h->hub_rows_sorted = new (std::nothrow) std::vector<int32_t>();
    if (!h->hub_rows_sorted) return MI_SPMM_ENOMEM;
    h->hub_rows_sorted->reserve(lr.size());
    for (const LongRow &r : lr) h->hub_rows_sorted->push_back(r.row);
    std::sort(h->hub_rows_sorted->begin(), h->hub_rows_sorted->end());
    return MI_SPMM_OK;
}

static bool range_has_hub(const mi_spmm_handle *h, int32_t row_begin, int32_t row_end)
{
    if (!h->hub_rows_sorted) return true;
    const auto it = std::lower_bound(h->hub_rows_sorted->begin(), h->hub_rows_sorted->end(), row_begin);
    return it != h->hub_rows_sorted->end() && *it < row_end;
}

// The hub kernel's longest row is the step's longest dependent chain; on a side stream it runs beside the rows kernel
// instead of in front of it.  The stream and its two events belong to the handle (created once, here, never in run()).
// A side stream is only worth its fork if its kernels RUN BESIDE the caller's.  The runtime hands a process's streams a few hardware
// queues per priority, in creation order, and the queues sit on a handful of command-processor pipes: a high-priority queue that shares its
// pipe with the caller's queue is served FIRST, not alongside -- the hub kernel then runs alone and the rows kernel after it (youtube-shaped
// N = 32: 0.32 ms instead of 0.17), and which handle gets such a queue is a matter of how many streams the process has made before
// (every other handle in a loop that makes one stream per handle: profiles/r04_side_streams.txt).  So a candidate stream is TESTED: a 40 us
// spin kernel on it and one on the null stream (the reference's stream, util.h:133-136, and torch's default) -- beside each other they take
// ~40 us, one after the other ~80 -- and a stream that fails is kept aside (so that the next one made gets another queue) until one passes
// or four have been tried.  preprocess only (it synchronises anyway); ~0.2 ms per candidate, once per handle.
static int concurrent_stream(hipStream_t *out, int priority, int *overlaps_out)
{
    constexpr int kTries = 4;
    hipStream_t cand[kTries] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t e0 = nullptr, e1 = nullptr, ef = nullptr, ej = nullptr;
    hipError_t e = hipEventCreate(&e0);
    if (e == hipSuccess) e = hipEventCreate(&e1);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&ef, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&ej, hipEventDisableTiming);
    int chosen = -1, made = 0;
    const unsigned long long ticks = 4000;      // 40 us of the 100 MHz counter
    for (int t = 0; t < kTries && e == hipSuccess && chosen < 0; ++t) {
        e = hipStreamCreateWithPriority(&cand[t], hipStreamNonBlocking, priority);
        if (e != hipSuccess) break;
        ++made;
        float best = 1e30f;
        for (int rep = 0; rep < 2 && e == hipSuccess; ++rep) {      // the first pair also pays the queue's first use
            e = hipEventRecord(e0, nullptr);
            if (e == hipSuccess) e = hipEventRecord(ef, nullptr);
            if (e == hipSuccess) e = hipStreamWaitEvent(cand[t], ef, 0);
            if (e == hipSuccess) { hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, cand[t], ticks, (unsigned int *)nullptr); e = hipGetLastError(); }
            if (e == hipSuccess) { hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, nullptr, ticks, (unsigned int *)nullptr); e = hipGetLastError(); }
            if (e == hipSuccess) e = hipEventRecord(ej, cand[t]);
            if (e == hipSuccess) e = hipStreamWaitEvent(nullptr, ej, 0);
            if (e == hipSuccess) e = hipEventRecord(e1, nullptr);
            if (e == hipSuccess) e = hipEventSynchronize(e1);
            float ms = 0.f;
            if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        if (std::getenv("MI_SPMM_DEBUG")) std::fprintf(stderr, "[mi_spmm] side stream candidate %d: spin pair %.1f us\n", t, best * 1e3f);
        if (e == hipSuccess && best < 0.078f) chosen = t;          // measured: 65-70 us beside each other (40 + launch and event overhead), 88-120 one after the other
    }
    const int keep = chosen >= 0 ? chosen : 0;
    for (int t = 0; t < made; ++t)
        if (t != keep && cand[t]) (void)hipStreamDestroy(cand[t]);
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (ef) (void)hipEventDestroy(ef);
    if (ej) (void)hipEventDestroy(ej);
    if (e != hipSuccess) { if (made > 0 && cand[keep]) (void)hipStreamDestroy(cand[keep]); return (int)e; }
    *out = cand[keep];
    *overlaps_out = chosen >= 0 ? 1 : 0;
    return MI_SPMM_OK;
}

static int ensure_side_streams(mi_spmm_handle *h)
{
    // A fork + join costs ~20 us of launch latency per run call (ddi-shaped N = 32: 0.047 -> 0.069 ms): worth it once the
    // step is long enough to hide something behind (estimate >= 0.2 ms) or the longest hub alone is a 50 us chain.
    // "hub_overlap": 0 = never, 1 = this rule, 2 = always.
    const double bytes = (double)h->nnz * (4.0 * h->feat + 8.0) + 4.0 * (double)h->num_v * h->feat;
    const bool long_step = bytes / 6e12 >= 200e-6;
    const bool exact_hubs = h->n_long > 0 && !h->split_long;
    h->overlap_on[0] = exact_hubs && (h->hub_overlap == 2 || (h->hub_overlap == 1 && (long_step || h->max_row_nnz >= 7000)));
    // The segment kernel gets a side stream of its own only where that was measured to pay (interleaved A/B over twelve graph shapes x three widths,
    // profiles/r04_side_streams.txt): narrow B (N <= 64: both the rows kernel, 8 lanes per row, and the segment chains are latency-bound and fill
    // each other's gaps -- citation- and wikikg2-shaped N = 32: -8 %) and no hub stream beside it.  Everywhere else a second stream is neutral to
    // harmful (+0 .. +16 %: protein-shaped N = 32, with 1 696 hubs on their stream): the segment kernel then stays on the caller's stream.
    // Round 5 (structured graphs, profiles/r05_regret.md): where the columns are local the rows kernel is fast (its neighbours share B rows in L2) and the
    // segment chains in front of it are what the step waits for -- beside it they cost nothing: segment_overlap = 1 is 0.78 - 0.94 of the time on
    // yelp- / products- / ppa- / youtube- / citation-community at every width; the round-4 A/B (uniformly random columns) found it neutral to harmful there.
    const bool seg_side = h->segment_overlap == 1 || (h->segment_overlap == 2 && ((!h->overlap_on[0] && h->feat <= 64) || h->local_pct >= 50));
    h->overlap_on[1] = seg_side && h->n_chunks > 0 && (h->hub_overlap == 2 || (h->hub_overlap == 1 && long_step));
    if (!(h->overlap_on[0] || h->overlap_on[1])) return MI_SPMM_OK;
    int lo = 0, hi = 0;
    HIP_TRY(hipDeviceGetStreamPriorityRange(&lo, &hi));
    if (!h->ev_fork) HIP_TRY(hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming));
    for (int i = 0; i < 2; ++i) {
        if (!h->overlap_on[i] || h->side[i]) continue;          // made once per handle, and only the ones this plan uses
        int ov = 0;
        const int rc = concurrent_stream(&h->side[i], ((h->side_priority >> i) & 1) ? hi : 0, &ov);
        if (rc != MI_SPMM_OK) return rc;
        h->side_overlaps[i] = ov;
        HIP_TRY(hipEventCreateWithFlags(&h->ev_join[i], hipEventDisableTiming));
    }
    return MI_SPMM_OK;
}

// The stream kernel class i of this run_rows call goes to: side stream i, made to wait for what the caller's stream holds
// so far (one fork event per call), or the caller's stream itself when overlap is off.
static int side_stream(mi_spmm_handle *h, int i, hipStream_t s, hipStream_t *out)
{
    *out = s;
    if (!h->overlap_on[i] || !h->side[i]) return MI_SPMM_OK;
    if (!h->fork_recorded) {
        HIP_TRY(hipEventRecord(h->ev_fork, s));
        h->fork_recorded = true;
    }
    if (!h->forked[i]) {
        HIP_TRY(hipStreamWaitEvent(h->side[i], h->ev_fork, 0));
        h->forked[i] = true;
    }
    *out = h->side[i];
    return MI_SPMM_OK;
}

namespace { int resolve_tile_cols(const mi_spmm_handle *h, int32_t N, int64_t ldb); }

// Column strips of the exact segments (plan.hpp), after either plan builder: the rule's cheap gates first, then one pass over
// the segments' columns (ascending? how many nonzeros?), then the S sub-segment tables.
static int plan_col_strips(mi_spmm_handle *h)
{
    h->n_strips = 1;
    h->seg_unsorted = -1;
    if (h->col_strips == 1 || h->n_chunks <= 0 || h->split_long || h->n_slots > 0 || h->feat <= 0 || h->num_cols < 2) return MI_SPMM_OK;
    // width of a column tile of the segment kernel at this handle's N (run_part: lane groups of N/4 lanes, capped by the tile rule)
    const int V = h->feat >= 4 ? 4 : 1;
    int lpr = 1;
    while (lpr < (h->feat + V - 1) / V) lpr <<= 1;
    lpr = lpr < 8 ? 8 : (lpr > 64 ? 64 : lpr);
    const int cap = resolve_tile_cols(h, h->feat, h->feat) / V;
    if (V == 4 && lpr > cap) lpr = cap;
    int tile = lpr * V;
    if (tile > h->feat) tile = h->feat;
    // (Round 4 left graphs with >= 50 % of their nonzeros near the diagonal alone: "neighbouring rows share their B rows through L2 already".  They do not
    //  once the rows are long: protein- / reddit-community (58 - 60 % local) 0.57 - 0.67 of the time with strips, and a BANDED matrix of 300-700-nonzero rows
    //  (100 % local: a row sits inside one or two strips) 0.54 - 0.69 -- the segment table is sorted by length, so a launch's rows come from all over the band.
    //  The gate is gone: profiles/r05_regret.md.)
    if (h->col_strips == 0 && h->local_pct >= 95) return MI_SPMM_OK;      // banded / mesh: the rows stay in the rows kernel (medium rule), nothing long is left to strip
    if (h->col_strips == 0 && resolve_col_strips(h->num_cols, tile, h->nnz, h->n_chunks, h->nnz, h->front_pct, h->local_pct) < 2) return MI_SPMM_OK;   // cannot pay whatever the survey says
    if (!h->d_col_bad) HIP_TRY(hipMalloc((void **)&h->d_col_bad, 256));
    void *d_sv = (char *)h->d_col_bad + 64;            // (the first bytes hold the column-range flag a second plan reads again)
    SegmentSurvey sv;
    if (h->strips_builder == 1) {                      // round 4: survey first (its nonzero count feeds the rule), tables afterwards
        const int rc = survey_segments(h->d_chunks, h->n_chunks, h->d_idx, d_sv, &sv);
        if (rc != 0) return rc;
        h->seg_unsorted = (int32_t)sv.unsorted;
        if (sv.unsorted) return MI_SPMM_OK;            // a row whose columns do not ascend cannot be cut by column without changing its order
        if ((int64_t)sv.nnz != h->seg_nnz) return MI_SPMM_ESTATE;      // the plan builders' count and the survey's are the same number
    }
    int64_t S = h->col_strips >= 2 ? h->col_strips : resolve_col_strips(h->num_cols, tile, h->seg_nnz, h->n_chunks, h->nnz, h->front_pct, h->local_pct);
    // local columns at a narrow B: a long row's nonzeros cluster in one or two strips, the other launches find it empty -- many strips only add launches
    // (ppa-community kLen 32: 14 strips DOUBLED the step, 4 are neutral, 2 take a fifth off; reddit- / protein-community kLen 32: 2 - 4 strips 0.88 - 0.95 of the time, 6 the same as none)
    if (h->col_strips == 0 && h->local_pct >= 50 && h->feat < 128 && S > 2) S = 2;
    // a strip is a launch: ~10 us of ramp each.  Not more strips than the step has 0.1 ms slices of work (hold-out: an R-MAT of scale 18 at kLen 32 -- 0.17 ms
    // of bytes -- lost 30 % to six strips; every fitted case has >= 0.14 ms per strip)
    if (h->col_strips == 0) {
        const double step = ((double)h->nnz * (4.0 * h->feat + 8.0) + 4.0 * (double)h->num_v * h->feat) / 6e12;
        const int64_t by_time = (int64_t)(step / 100e-6);
        if (S > by_time) S = by_time;
    }
    if (S > h->num_cols) S = h->num_cols;
    if (S > kMaxColStrips) S = kMaxColStrips;
    if (S < 2) return MI_SPMM_OK;
    const size_t need = (size_t)S * (size_t)h->n_chunks, bytes = need * sizeof(Chunk);
    if (h->strips_cap < need) {
        if (h->d_strips) (void)hipFree(h->d_strips);
        h->d_strips = nullptr;
        h->strips_cap = 0;
        if (hipMalloc((void **)&h->d_strips, bytes) != hipSuccess) { h->d_strips = nullptr; return MI_SPMM_ENOMEM; }
        h->strips_cap = need;
    }
    if (h->strips_builder == 1) {
        const int brc = build_col_strips(h->d_chunks, h->n_chunks, h->d_idx, h->num_cols, (int32_t)S, h->d_strips);
        if (brc != 0) return brc;
    } else {                                           // one pass: ascending? and the tables, which an unsorted segment leaves unusable
        const int brc = strip_segments(h->d_chunks, h->n_chunks, h->d_idx, h->num_cols, (int32_t)S, h->d_strips, d_sv, &sv);
        if (brc != 0) return brc;
        h->seg_unsorted = (int32_t)sv.unsorted;
        if (sv.unsorted) return MI_SPMM_OK;
    }
    h->n_strips = (int32_t)S;
    h->ws_bytes += bytes;
    return MI_SPMM_OK;
}

// preprocess with no host pass over the rows: column check, block detection, classification, scans,
// segment emission and the length sort all run on the device; one small copy comes back.
static int preprocess_on_gpu(mi_spmm_handle *h, std::chrono::steady_clock::time_point t0)
{
    const int32_t M = h->num_v;
    auto tp = t0;
    auto lap = [&](int i) {
        const auto now = std::chrono::steady_clock::now();
        h->phase_us[i] = std::chrono::duration<double, std::micro>(now - tp).count();
        tp = now;
    };
    lap(0);
    unsigned int *d_bad = nullptr;
    if (h->nnz > 0) {
        if (!h->d_col_bad) HIP_TRY(hipMalloc((void **)&h->d_col_bad, 256));
        d_bad = h->d_col_bad;
        hipError_t e = hipMemsetAsync(d_bad, 0, sizeof(unsigned int), 0);
        if (e == hipSuccess) {
            const int64_t want = (h->nnz + kBlockThreads * 8 - 1) / (kBlockThreads * 8);
            const int grid = (int)(want < 1 ? 1 : (want > 4096 ? 4096 : want));
            hipLaunchKernelGGL(csr_check_cols, dim3(grid), dim3(kBlockThreads), 0, 0, h->d_idx, h->nnz, h->num_cols, d_bad);
            e = hipGetLastError();
        }
        if (e != hipSuccess) return (int)e;
    }
    lap(1);
    if (h->block_path && !h->ftz && block_path_shape_ok(h->feat) && M >= 16 && h->nnz > 0) {
        const int32_t n_groups = (M + 15) / 16;
        if (hipMalloc((void **)&h->d_blk_flag, (size_t)n_groups) != hipSuccess) return MI_SPMM_ENOMEM;
        // a group's shared list may be as long as the longest exact segment (auto hub threshold: at most its largest candidate)
        hipLaunchKernelGGL(detect_row_blocks, dim3((n_groups + 3) / 4), dim3(kBlockThreads), 0, 0, h->d_ptr, h->d_idx, M, (int32_t)h->nnz,
                           (int32_t)h->block_min_len, (int32_t)(h->long_thr > 0 ? h->long_thr : hist_threshold(kHistN - 1)), h->d_blk_flag);
        const hipError_t e = hipGetLastError();
        if (e != hipSuccess) { free_plan(h); return (int)e; }
    }
    lap(2);
    // The plan, then the column strips of its segments -- and, where strips are in force and the hub threshold was ours to choose, a second plan WITHOUT
    // hubs when the longest row's sub-chains would hide inside the strips' launches: as stripped segments those rows gather out of L2 like the rest,
    // where the hub kernel gathers them out of all of B beside the strip launches (reddit-shaped N = 128 / 256: 5.07 -> 4.46, 8.97 -> 8.15 ms).  The longest
    // row's chain, at 100 ns per nonzero over all strips, has to fit into half of the stripped step's estimate: at N = 32 the same graph's 20 758-nonzero row
    // would be the step (1.08 -> 1.70 ms) and protein-shaped N = 32 loses 7 % (0.72 -> 0.78): both keep their hubs.  profiles/r04_col_strips.txt section 8
    // (attempt 2, ADVICE r4: the folded plan surveys the former hub rows for the first time; if one of them has non-ascending columns, or the rule on
    //  the new survey says no strips, the fold has nothing to stand on -- a 10^4-nonzero row would run as ONE unstripped segment chain -- and the plan
    //  with hubs is built again, this time to be kept)
    int32_t mthr_retry = 0;
    for (int attempt = 0; attempt < 3; ++attempt) {
        PlanOut po;
        const int64_t mthr_want = mthr_retry > 0 ? mthr_retry : h->medium_thr;
        const int32_t mthr = (int32_t)((h->long_thr == 0 || mthr_want < h->long_thr) ? mthr_want : h->long_thr);   // 0 = auto
        const int rc = build_plan_gpu(h->d_ptr, h->d_idx, M, h->num_cols, h->feat, h->nnz, h->d_blk_flag, d_bad, mthr, (int32_t)h->long_thr,
                                      (int32_t)h->long_chunk, (int32_t)h->split_long, (int32_t)h->seg_order, &h->scratch_a, &h->scratch_b, &po);
        h->d_chunks = po.d_chunks;
        h->d_long = po.d_long;
        h->d_blk_groups = po.d_blk_groups;
        if (rc != 0) { free_plan(h); return rc; }
        h->n_chunks = po.n_chunks;
        h->n_long = po.n_long;
        h->n_slots = po.n_slots;
        h->n_medium = po.n_medium;
        h->n_blk_groups = po.n_blk_groups;
        h->max_row_nnz = po.max_len;
        h->medium_res = po.mthr;
        h->long_thr = po.thr;
        h->local_pct = po.local_pct;
        h->front_pct = po.front_pct;
        h->seg_nnz = po.seg_nnz;
        {
            const int crc = plan_col_strips(h);
            if (crc != 0) { free_plan(h); return crc; }
        }
        const double step_s = ((double)h->nnz * (4.0 * h->feat + 8.0) + 4.0 * (double)M * h->feat) / 12e12;     // a stripped step: about twice the gather model's rate
        const bool fold_hubs = attempt == 0 && h->long_thr_user == 0 && !h->split_long && h->n_strips > 1 && h->n_long > 0 &&
                               (double)h->max_row_nnz * 100e-9 <= 0.5 * step_s;
        const bool unfold = attempt == 1 && h->n_strips <= 1;      // the fold lost its strips: back to the plan with hubs
        // Local columns, a wide B, segments that hold a quarter of the nonzeros and NO strips for them (columns out of order, or B beyond the rule): what is left
        // to them is the length-sorted table that scatters neighbours -- the rows kernel takes them instead, up to 1 024 nonzeros (protein-unsorted N = 128 / 256:
        // 0.83 - 0.85 of the time; profiles/r05_regret.md).  One more plan, on the device: ~0.3 ms of preprocess.
        // (narrow B: up to 512 -- protein-unsorted kLen 32: 0.85)
        //  and only on graphs whose ROWS are long, mean degree >= 64: citation-community, mean 10, lost 17 % with it)
        const int32_t keep_to = h->feat >= 128 ? 1024 : 512;
        const bool keep_rows = attempt == 0 && !fold_hubs && mthr_retry == 0 && h->medium_thr == 0 && !h->split_long && h->local_pct >= 50 && h->local_pct < 95 &&
                               M >= 65536 && h->n_strips <= 1 && h->medium_res < keep_to && h->seg_nnz * 4 >= h->nnz && (h->feat >= 128 || h->nnz / M >= 64);
        if (keep_rows) {
            mthr_retry = keep_to;
            if (h->d_chunks) (void)hipFree(h->d_chunks);
            if (h->d_long) (void)hipFree(h->d_long);
            if (h->d_blk_groups) (void)hipFree(h->d_blk_groups);
            h->d_chunks = nullptr; h->d_long = nullptr; h->d_blk_groups = nullptr;
            h->ws_bytes = 0;
            h->long_thr = h->long_thr_user;      // auto again (0) or the caller's
            attempt = -1;                        // the loop's ++ makes it 0: the same first attempt, with the other medium threshold
            continue;
        }
        if (!fold_hubs && !unfold) break;
        if (h->d_chunks) (void)hipFree(h->d_chunks);
        if (h->d_long) (void)hipFree(h->d_long);
        if (h->d_blk_groups) (void)hipFree(h->d_blk_groups);
        h->d_chunks = nullptr;
        h->d_long = nullptr;
        h->d_blk_groups = nullptr;
        h->ws_bytes = 0;
        if (fold_hubs) h->long_thr = 1 << 30;          // every row up to any length: one exact segment (stripped)
        else h->long_thr = 0;                           // auto threshold again: the first attempt's plan, kept this time
    }
    if (h->n_blk_groups == 0 && h->d_blk_flag) { (void)hipFree(h->d_blk_flag); h->d_blk_flag = nullptr; }
    h->n_rows_for_rows_kernel = (int64_t)M - 16 * (int64_t)h->n_blk_groups - (int64_t)h->n_long - (int64_t)h->n_medium;
    h->ldp = ((int64_t)h->feat + 3) / 4 * 4;
    lap(3);
    if (h->n_slots > 0) {
        const size_t pb = (size_t)h->n_slots * (size_t)h->ldp * sizeof(float);
        if (hipMalloc((void **)&h->d_partials, pb) != hipSuccess) { free_plan(h); return MI_SPMM_ENOMEM; }
        h->ws_bytes += pb;
    }
    h->ws_bytes += (size_t)h->n_chunks * sizeof(Chunk) + (size_t)h->n_long * sizeof(LongRow);
    {
        const int brc = build_block_items(h);
        if (brc != 0) { free_plan(h); return brc; }
    }
    lap(4);
    {
        int src = ensure_side_streams(h);
        if (src == 0) src = note_hub_rows(h);
        if (src != 0) { free_plan(h); return src; }
    }
    // mi_spmm.h: preprocess synchronises.  Everything above ran on the null stream and most of it was waited for by the copies back, but the last
    // launches (build_col_strips, the block items' device copy) were not: a run() on a non-blocking stream is not ordered behind the null stream.
    HIP_TRY(hipStreamSynchronize(0));
    h->prepared = true;
    h->preprocess_us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    return MI_SPMM_OK;
}

extern "C" {

int mi_spmm_abi_version(void) { return MI_SPMM_ABI_VERSION; }

const char *mi_spmm_build_info(void)
{
    return "mi_spmm gfx950 (CDNA4) hand-written HIP; abi "
#define MI_STR2(x) #x
#define MI_STR(x) MI_STR2(x)
        MI_STR(MI_SPMM_ABI_VERSION) "; hip " MI_STR(HIP_VERSION_MAJOR) "." MI_STR(HIP_VERSION_MINOR);
}

const char *mi_spmm_strerror(int code)
{
    switch (code) {
    case MI_SPMM_OK: return "ok";
    case MI_SPMM_EINVAL: return "mi_spmm: invalid argument";
    case MI_SPMM_ENOMEM: return "mi_spmm: out of memory";
    case MI_SPMM_ESTATE: return "mi_spmm: bad handle or run() before preprocess()";
    case MI_SPMM_ECSR: return "mi_spmm: malformed CSR (row_ptr not monotone, row_ptr[M] != nnz, or column out of range)";
    case MI_SPMM_EUNSUPPORTED: return "mi_spmm: unsupported option or shape";
    case MI_SPMM_ENODEVICE: return "mi_spmm: no usable HIP device";
    default: break;
    }
    if (code > 0) return hipGetErrorString((hipError_t)code);
    return "mi_spmm: unknown error";
}

int mi_spmm_create(mi_spmm_handle **out, const int32_t *d_row_ptr, const int32_t *d_col_idx,
                   const float *d_vals, int32_t num_v, int32_t num_cols, int64_t nnz, int32_t feat_in)
{
    if (!out) return MI_SPMM_EINVAL;
    *out = nullptr;
    if (num_v < 0 || num_cols < 0 || nnz < 0 || feat_in < 0) return MI_SPMM_EINVAL;
    if (nnz > INT32_MAX) return MI_SPMM_EUNSUPPORTED;  // int32 row_ptr cannot address more
    if (!d_row_ptr) return MI_SPMM_EINVAL;
    if (nnz > 0 && (!d_col_idx || !d_vals)) return MI_SPMM_EINVAL;
    mi_spmm_handle *h = new (std::nothrow) mi_spmm_handle();
    if (!h) return MI_SPMM_ENOMEM;
    std::memset(h, 0, sizeof(*h));
    h->magic = kMagic;
    h->d_ptr = d_row_ptr;
    h->d_idx = d_col_idx;
    h->d_val = d_vals;
    h->num_v = num_v;
    h->num_cols = num_cols;
    h->nnz = nnz;
    h->feat = feat_in;
    h->medium_thr = 0;     // auto: 64, or 32 on skewed degree distributions (resolved in preprocess)
    h->long_thr = 0;       // 0 = auto, resolved in preprocess: clamp(nnz / 8192, 256, 2048)
    h->long_thr_user = 0;
    h->split_long = 0;     // every row one chain in stored order (the reference's definition, spmm_ref.cu:10-14)
    h->hub_slice = 0;
    h->hub_overlap = 1;
    h->side_priority = 3;
    h->segment_overlap = 2;   // auto: a side stream for the segment kernel only where its longest chains are long (ensure_side_streams)
    h->long_chunk = 256;   // the reference's kBatchSize (spmm_opt.cu:6)
    h->rows_per_block = 0; // auto: one row per lane group (measured best at every N, profiles/r01_sweeps)
    h->xcd_remap = -1;     // auto (see run)
    h->nt_store = 1;       // C is write-once
    h->nt_stream = 0;      // (col,val) fetches straddle lines: nt would drop the line before its other half is used
    h->block_path = 1;
    h->block_min_len = 8;
    h->block_share = 2;
    h->block_max_pieces = kMaxPieces;
    h->block_run_min = 32;
    h->block_wg_waves = 1;
    h->kernel = 2;
    h->gpu_preprocess = 1;
    h->block_threads = 256;
    h->split_cols = 1;
    h->fused_step = 2;      // auto (run_part: short steps only)
    h->fused_order = 0;     // auto (run_part: the role whose longest chain lasts longest goes first)
    h->col_strips = 0;      // auto (plan.hpp resolve_col_strips)
    h->n_strips = 1;
    h->seg_unsorted = -1;
    h->segment_unroll = 0;  // auto: 32 (the whole 32-pair item in flight: 0-13 % faster than 8 on every shape, profiles/r01_segment_unroll.txt) -- 16 for
                            // column-strip launches (gathers served by L2 return sooner, occupancy buys more: -2 .. -13 %, profiles/r04_col_strips.txt)
    *out = h;
    return MI_SPMM_OK;
}

int mi_spmm_set_feat(mi_spmm_handle *h, int32_t feat_in)
{
    if (!good(h) || feat_in < 0) return MI_SPMM_EINVAL;
    if (feat_in != h->feat) {
        h->feat = feat_in;
        free_plan(h);
    }
    return MI_SPMM_OK;
}

int mi_spmm_destroy(mi_spmm_handle *h)
{
    if (!h) return MI_SPMM_OK;
    if (!good(h)) return MI_SPMM_ESTATE;
    free_plan(h);
    scratch_release(&h->scratch_a);
    scratch_release(&h->scratch_b);
    if (h->d_col_bad) (void)hipFree(h->d_col_bad);
    if (h->d_blk_items) (void)hipFree(h->d_blk_items);
    if (h->d_strips) (void)hipFree(h->d_strips);
    for (int i = 0; i < 2; ++i) {
        if (h->side[i]) (void)hipStreamDestroy(h->side[i]);
        if (h->ev_join[i]) (void)hipEventDestroy(h->ev_join[i]);
    }
    if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
    h->magic = 0;
    delete h;
    return MI_SPMM_OK;
}

int mi_spmm_set_option(mi_spmm_handle *h, const char *key, int64_t v)
{
    if (!good(h) || !key) return MI_SPMM_EINVAL;
    const std::string k(key);
    if (k == "use_graph") return MI_SPMM_EUNSUPPORTED;      // removed in round 5 (it only ever lost: profiles/r05_use_graph_experiment.md); a caller captures run() itself
    else if (k == "medium_row_threshold") { if (v < 0) return MI_SPMM_EINVAL; h->medium_thr = v > INT32_MAX ? INT32_MAX : v; h->tuned_mask &= ~4u; free_plan(h); }
    else if (k == "long_row_threshold") { if (v < 0) return MI_SPMM_EINVAL; h->long_thr_user = v > INT32_MAX ? INT32_MAX : v; h->long_thr = h->long_thr_user; h->tuned_mask &= ~64u; free_plan(h); }
    else if (k == "split_long_rows") { h->split_long = v ? 1 : 0; free_plan(h); }
    else if (k == "hub_slice") { if (v != 0 && v != 16 && v != 32 && v != 64) return MI_SPMM_EINVAL; h->hub_slice = v; }
    else if (k == "hub_overlap") { if (v < 0 || v > 2) return MI_SPMM_EINVAL; h->hub_overlap = v; free_plan(h); }
    else if (k == "segment_overlap") { if (v < 0 || v > 2) return MI_SPMM_EINVAL; h->segment_overlap = v; free_plan(h); }
    else if (k == "flush_denormals") { h->ftz = v ? 1 : 0; free_plan(h); }       // (the block path is not used with it: preprocess again)
    else if (k == "side_priority") {
        if (v < 0 || v > 3) return MI_SPMM_EINVAL;
        if (v != h->side_priority) {          // the streams are made by the next preprocess
            for (int i = 0; i < 2; ++i) {
                if (h->side[i]) { (void)hipStreamSynchronize(h->side[i]); (void)hipStreamDestroy(h->side[i]); h->side[i] = nullptr; }
                if (h->ev_join[i]) { (void)hipEventDestroy(h->ev_join[i]); h->ev_join[i] = nullptr; }
            }
            if (h->ev_fork) { (void)hipEventDestroy(h->ev_fork); h->ev_fork = nullptr; }
        }
        h->side_priority = v;
        free_plan(h);
    }
    else if (k == "long_row_chunk") { if (v < 1 || v > kMaxLongChunk) return MI_SPMM_EINVAL; h->long_chunk = v; free_plan(h); }
    else if (k == "rows_per_block") { if (v < 0 || v > (1 << 20)) return MI_SPMM_EINVAL; h->rows_per_block = v; }
    else if (k == "xcd_remap") h->xcd_remap = v < 0 ? -1 : (v ? 1 : 0);
    else if (k == "kernel") {
        if (v == 1) return MI_SPMM_EUNSUPPORTED;   // the first-generation rows kernel was retired in round 3 (git history: 0894343)
        if (v != 2) return MI_SPMM_EINVAL;
        h->kernel = v;
    }
    else if (k == "gpu_preprocess") { h->gpu_preprocess = v ? 1 : 0; free_plan(h); }
    else if (k == "split_cols") h->split_cols = v ? 1 : 0;
    else if (k == "tile_cols") { if (v != 0 && v != 32 && v != 64 && v != 128 && v != 256) return MI_SPMM_EINVAL; h->tile_cols = v; h->tuned_mask &= ~1u; }
    else if (k == "segment_unroll") { if (v != 0 && v != 8 && v != 16 && v != 32) return MI_SPMM_EINVAL; h->segment_unroll = v; }
    else if (k == "col_strips") { if (v < 0 || v > kMaxColStrips) return MI_SPMM_EINVAL; h->col_strips = v; h->tuned_mask &= ~2u; free_plan(h); }
    else if (k == "fused_step") { if (v < 0 || v > 2) return MI_SPMM_EINVAL; h->fused_step = v; h->tuned_mask &= ~8u; }
    else if (k == "fused_order") { if (v < 0 || v > 2) return MI_SPMM_EINVAL; h->fused_order = v; }
    else if (k == "segment_order") { if (v < 0 || v > 2) return MI_SPMM_EINVAL; h->seg_order = v; h->tuned_mask &= ~16u; free_plan(h); }
    else if (k == "autotune") { if (v != 0 && v != 1) return MI_SPMM_EINVAL; h->autotune = v; free_plan(h); }
    else if (k == "rows_unroll") { if (v != 0 && v != 8 && v != 16) return MI_SPMM_EINVAL; h->rows_unroll = v; }
    else if (k == "col_strips_builder") { if (v != 0 && v != 1) return MI_SPMM_EINVAL; h->strips_builder = v; free_plan(h); }
    else if (k == "block_threads") { if (v != 64 && v != 128 && v != 256) return MI_SPMM_EINVAL; h->block_threads = v; }
    else if (k == "nt_store") h->nt_store = v ? 1 : 0;
    else if (k == "nt_stream") h->nt_stream = v ? 1 : 0;
    else if (k == "block_path") { h->block_path = v ? 1 : 0; free_plan(h); }
    else if (k == "block_min_len") { if (v < 1) return MI_SPMM_EINVAL; h->block_min_len = v; free_plan(h); }
    else if (k == "block_share") { if (v < 1 || v > kMaxShare) return MI_SPMM_EINVAL; h->block_share = v; free_plan(h); }
    else if (k == "block_max_pieces") { if (v < 1 || v > kMaxPieces) return MI_SPMM_EINVAL; h->block_max_pieces = v; free_plan(h); }
    else if (k == "block_wg_waves") { if (v != 1 && v != 2 && v != 4) return MI_SPMM_EINVAL; h->block_wg_waves = v; }
    else if (k == "block_run_min") { if (v < 1) return MI_SPMM_EINVAL; h->block_run_min = v > INT32_MAX ? INT32_MAX : v; free_plan(h); }
    else return MI_SPMM_EUNSUPPORTED;
    return MI_SPMM_OK;
}

int mi_spmm_get_option(const mi_spmm_handle *h, const char *key, int64_t *value)
{
    if (!good(h) || !key || !value) return MI_SPMM_EINVAL;
    const std::string k(key);
    if (k == "long_row_threshold") *value = h->long_thr;
    else if (k == "medium_row_threshold") *value = h->prepared ? h->medium_res : h->medium_thr;
    else if (k == "n_medium_rows") *value = h->n_medium;
    else if (k == "n_partial_slots") *value = h->n_slots;
    else if (k == "long_row_chunk") *value = h->long_chunk;
    else if (k == "split_long_rows") *value = h->split_long;
    else if (k == "hub_slice") *value = h->hub_slice;
    else if (k == "hub_overlap") *value = h->hub_overlap;
    else if (k == "side_priority") *value = h->side_priority;
    else if (k == "side_stream_overlaps") *value = h->side[0] ? h->side_overlaps[0] : -1;
    else if (k == "segment_overlap") *value = h->segment_overlap;
    else if (k == "flush_denormals") *value = h->ftz;
    else if (k == "n_hub_rows") *value = h->split_long ? 0 : h->n_long;
    else if (k == "rows_per_block") *value = h->rows_per_block;
    else if (k == "xcd_remap") *value = h->xcd_remap;
    else if (k == "kernel") *value = h->kernel;
    else if (k == "gpu_preprocess") *value = h->gpu_preprocess;
    else if (k == "block_threads") *value = h->block_threads;
    else if (k == "segment_unroll") *value = h->segment_unroll;
    else if (k == "split_cols") *value = h->split_cols;
    else if (k == "tile_cols") *value = h->tile_cols;
    else if (k == "nt_store") *value = h->nt_store;
    else if (k == "nt_stream") *value = h->nt_stream;
    else if (k == "block_path") *value = h->block_path;
    else if (k == "n_long_rows") *value = h->n_long;
    else if (k == "n_chunks") *value = h->n_chunks;
    else if (k == "col_strips") *value = h->col_strips;
    else if (k == "n_col_strips") *value = h->n_strips;
    else if (k == "segments_unsorted") *value = h->seg_unsorted;
    else if (k == "col_strips_builder") *value = h->strips_builder;
    else if (k == "fused_step") *value = h->fused_step;
    else if (k == "segment_order") *value = h->seg_order;
    else if (k == "autotune") *value = h->autotune;
    else if (k == "rows_unroll") *value = h->rows_unroll;
    else if (k == "autotune_evals") *value = h->tune_evals;
    else if (k == "autotune_auto_us") *value = (int64_t)(h->tune_auto_ms * 1e3);
    else if (k == "autotune_best_us") *value = (int64_t)(h->tune_best_ms * 1e3);
    else if (k == "autotune_mask") *value = h->tuned_mask;
    else if (k == "fused_step_in_force") *value = h->last_fused;
    else if (k == "fused_order") *value = h->fused_order;
    else if (k == "rows_unroll_in_force") *value = h->last_rows_deep ? 16 : 8;
    else if (k == "fused_order_in_force") *value = h->last_fused ? (h->last_seg_first ? 2 : 1) : 0;
    else if (k == "segment_nnz") *value = h->seg_nnz;
    else if (k == "col_strips_table_hash") {           // FNV-1a over the strip tables (copied back: a test's question, not a step's)
        *value = 0;
        if (h->prepared && h->d_strips && h->n_strips > 1) {
            std::vector<Chunk> t((size_t)h->n_strips * (size_t)h->n_chunks);
            HIP_TRY(hipMemcpy(t.data(), h->d_strips, t.size() * sizeof(Chunk), hipMemcpyDeviceToHost));
            uint64_t f = 1469598103934665603ull;
            const unsigned char *b = reinterpret_cast<const unsigned char *>(t.data());
            for (size_t i = 0; i < t.size() * sizeof(Chunk); ++i) { f ^= b[i]; f *= 1099511628211ull; }
            *value = (int64_t)(f >> 1);
        }
    }
    else if (k == "workspace_bytes") *value = (int64_t)(h->ws_bytes + h->scratch_a.cap + h->scratch_b.cap);   // plan tables + partial sums + the two preprocess arenas (kept until destroy)
    else if (k == "feat") *value = h->feat;
    else if (k == "num_v") *value = h->num_v;
    else if (k == "num_cols") *value = h->num_cols;
    else if (k == "max_row_nnz") *value = h->max_row_nnz;
    else if (k == "column_locality_pct") *value = h->local_pct;
    else if (k == "column_front_pct") *value = h->front_pct;
    else if (k == "n_launches") *value = h->last_launches;
    else if (k == "lanes_per_row") *value = h->last_lpr;
    else if (k == "vector_width") *value = h->last_v;
    else if (k == "wide_addressing") *value = h->last_wide;
    else if (k == "n_block_groups") *value = h->n_blk_groups;
    else if (k == "block_min_len") *value = h->block_min_len;
    else if (k == "block_share") *value = h->block_share;
    else if (k == "block_max_pieces") *value = h->block_max_pieces;
    else if (k == "block_run_min") *value = h->block_run_min;
    else if (k == "block_wg_waves") *value = h->block_wg_waves;
    else if (k == "n_block_items") *value = h->n_blk_items;
    else if (k == "n_block_pieces") *value = h->n_blk_pieces;
    else if (k == "n_block_passes") *value = h->n_blk_passes;
    else if (k == "n_block_shared_items") *value = h->n_blk_shared_items;
    else if (k == "preprocess_us") *value = (int64_t)h->preprocess_us;
    else if (k == "pre_d2h_us") *value = (int64_t)h->phase_us[0];
    else if (k == "pre_colcheck_us") *value = (int64_t)h->phase_us[1];
    else if (k == "pre_detect_us") *value = (int64_t)h->phase_us[2];
    else if (k == "pre_table_us") *value = (int64_t)h->phase_us[3];
    else if (k == "pre_upload_us") *value = (int64_t)h->phase_us[4];
    else if (k == "prepared") *value = h->prepared ? 1 : 0;
    else return MI_SPMM_EUNSUPPORTED;
    return MI_SPMM_OK;
}

// The plan half of preprocess (mi_spmm_preprocess, below run_rows, adds the optional graph capture).
static int preprocess_plan(mi_spmm_handle *h)
{
    if (!good(h)) return MI_SPMM_ESTATE;
    const auto t0 = std::chrono::steady_clock::now();
    free_plan(h);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return MI_SPMM_ENODEVICE;

    const int32_t M = h->num_v;
    // Hub threshold.  Rows up to it stay ONE exact segment of the segment kernel; longer ones (hubs) go to the hub kernel
    // (stored order) or, with "split_long_rows", are cut into pieces.  An explicit value is taken as is.  auto:
    //   split mode: 2048 once there is enough work to hide a 2048-nonzero segment (nnz >= 2^24), proportionally lower for
    //     small matrices (profiles/r01_thresholds.txt);
    //   default (exact) mode: plan.hpp resolve_hub_threshold (needs the row-length histogram: resolved inside the plan builders).
    if (h->long_thr_user > 0) h->long_thr = h->long_thr_user;
    else if (h->split_long) {
        int64_t t = h->nnz / 8192;
        h->long_thr = t < 256 ? 256 : (t > 2048 ? 2048 : t);
    } else h->long_thr = 0;     // resolved by the plan builder from the row-length histogram (plan.hpp: resolve_hub_threshold)
    // The hub kernel moves 16-byte parts of B rows: a B narrower than one part keeps its long rows as single exact
    // segments of the segment kernel (slow on a real hub, but N < 4 has no bandwidth to speak of either way).
    if (!h->split_long && h->feat < 4) h->long_thr = INT32_MAX;
    if (h->gpu_preprocess) return preprocess_on_gpu(h, t0);
    std::vector<int32_t> ptr((size_t)M + 1);
    HIP_TRY(hipMemcpy(ptr.data(), h->d_ptr, sizeof(int32_t) * ((size_t)M + 1), hipMemcpyDeviceToHost));
    // data.cu:40-45 asserts ptr[num_v] == num_e; we also need monotone rows,
    // or the kernels would read outside col_idx/vals.
    if (ptr[0] < 0 || (int64_t)ptr[M] != h->nnz) return MI_SPMM_ECSR;
    int32_t max_len = 0;
    for (int32_t r = 0; r < M; ++r) {
        const int32_t len = ptr[r + 1] - ptr[r];
        if (len < 0) return MI_SPMM_ECSR;
        if (len > max_len) max_len = len;
    }
    if (ptr[0] != 0 && (int64_t)ptr[0] > h->nnz) return MI_SPMM_ECSR;
    h->max_row_nnz = max_len;
    // (as in the device builder: a block group's list may be as long as the largest candidate of the auto threshold)
    const int32_t detect_max = (int32_t)(h->long_thr > 0 ? h->long_thr : hist_threshold(kHistN - 1));
    // the column sample the device builder takes inside its own pass (locality: tile width, medium threshold, strips; front-loaded columns: strips)
    if (h->nnz > 0) {
        if (!h->d_col_bad) HIP_TRY(hipMalloc((void **)&h->d_col_bad, 256));
        const int src = sample_columns_gpu(h->d_ptr, h->d_idx, M, h->num_cols, h->nnz, (char *)h->d_col_bad + 64, &h->local_pct, &h->front_pct);
        if (src != 0) return src;
    } else h->local_pct = 0;
    if (h->long_thr == 0) {      // auto, exact mode: the same histogram and the same rule as the device builder
        LenHist hist;
        std::memset(&hist, 0, sizeof(hist));
        for (int32_t r = 0; r < M; ++r) {
            const int32_t len = ptr[r + 1] - ptr[r];
            for (int i = 0; i < kHistN && len > hist_threshold(i); ++i) { ++hist.cnt[i]; hist.nnz[i] += (unsigned long long)len; }
        }
        h->long_thr = resolve_hub_threshold(h->nnz, M, h->num_cols, h->feat, hist.nnz, max_len);
    }
    auto lap = [&](int i, std::chrono::steady_clock::time_point &from) {
        const auto now = std::chrono::steady_clock::now();
        h->phase_us[i] = std::chrono::duration<double, std::micro>(now - from).count();
        from = now;
    };
    auto tp = t0;
    lap(0, tp);

    // column range: one pass over col_idx on the device (an out-of-range
    // column is an out-of-bounds read of B, i.e. a GPU fault)
    if (h->nnz > 0) {
        unsigned int *d_bad = nullptr;
        HIP_TRY(hipMalloc((void **)&d_bad, 256));
        hipError_t e = hipMemset(d_bad, 0, sizeof(unsigned int));
        if (e == hipSuccess) {
            const int64_t want = (h->nnz + kBlockThreads * 8 - 1) / (kBlockThreads * 8);
            const int grid = (int)(want < 1 ? 1 : (want > 4096 ? 4096 : want));
            hipLaunchKernelGGL(csr_check_cols, dim3(grid), dim3(kBlockThreads), 0, 0, h->d_idx, h->nnz,
                               h->num_cols, d_bad);
            e = hipGetLastError();
        }
        unsigned int bad = 0;
        if (e == hipSuccess) e = hipMemcpy(&bad, d_bad, sizeof(bad), hipMemcpyDeviceToHost);
        (void)hipFree(d_bad);
        if (e != hipSuccess) return (int)e;
        if (bad) return MI_SPMM_ECSR;
    }

    lap(1, tp);
    // block path: 16-row groups with one shared column list (min_len <= L <= split threshold)
    {
        const int32_t N = h->feat;
        const bool n_ok = block_path_shape_ok(N);
        if (h->block_path && !h->ftz && n_ok && M >= 16 && h->nnz > 0) {
            const int32_t n_groups = (M + 15) / 16;
            if (hipMalloc((void **)&h->d_blk_flag, (size_t)n_groups) != hipSuccess) return MI_SPMM_ENOMEM;
            const int grid = (n_groups + 3) / 4;  // one wave per group
            hipLaunchKernelGGL(detect_row_blocks, dim3(grid), dim3(kBlockThreads), 0, 0, h->d_ptr, h->d_idx, M, (int32_t)h->nnz,
                               (int32_t)h->block_min_len, detect_max, h->d_blk_flag);
            hipError_t e = hipGetLastError();
            std::vector<uint8_t> flags((size_t)n_groups);
            if (e == hipSuccess) e = hipMemcpy(flags.data(), h->d_blk_flag, (size_t)n_groups, hipMemcpyDeviceToHost);
            if (e != hipSuccess) { free_plan(h); return (int)e; }
            std::vector<int32_t> groups;
            for (int32_t g = 0; g < n_groups; ++g) if (flags[(size_t)g]) groups.push_back(g);
            if (groups.empty()) {
                (void)hipFree(h->d_blk_flag);
                h->d_blk_flag = nullptr;
            } else {
                if (hipMalloc((void **)&h->d_blk_groups, groups.size() * sizeof(int32_t)) != hipSuccess) { free_plan(h); return MI_SPMM_ENOMEM; }
                e = hipMemcpy(h->d_blk_groups, groups.data(), groups.size() * sizeof(int32_t), hipMemcpyHostToDevice);
                if (e != hipSuccess) { free_plan(h); return (int)e; }
                h->n_blk_groups = (int32_t)groups.size();
            }
        }
    }

    lap(2, tp);
    // Segment table (the reference's Task list, spmm_opt.cu:43-54, kept only for rows that need it):
    //   len > long_thr                 -> pieces of long_chunk nonzeros, partial sums + ordered reduce
    //   medium_thr < len <= long_thr   -> ONE segment = the whole row, stored straight to C (exact order)
    // Rows of block-path groups are excluded (their L <= long_thr by construction).  Segments are
    // sorted longest first: neighbours in a wave have similar lengths and the tail is short.
    std::vector<uint8_t> flags_host;
    if (h->n_blk_groups > 0) {
        flags_host.resize((size_t)(M + 15) / 16);
        HIP_TRY(hipMemcpy(flags_host.data(), h->d_blk_flag, flags_host.size(), hipMemcpyDeviceToHost));
    }
    std::vector<Chunk> chunks;
    std::vector<LongRow> longs;
    const int32_t thr = (int32_t)h->long_thr, clen = (int32_t)h->long_chunk;
    // the same auto rule as the device builder (plan.hpp resolve_medium_threshold)
    const int32_t mthr = resolve_medium_threshold((int32_t)h->medium_thr, h->nnz, M, max_len,
                                                  (int32_t)(h->long_thr > INT32_MAX ? INT32_MAX : h->long_thr), h->local_pct, h->feat);
    h->medium_res = mthr;
    int32_t n_slots = 0, n_medium = 0;
    if (max_len > mthr) {
        chunks.reserve(1 << 16);
        for (int32_t r = 0; r < M; ++r) {
            const int32_t beg = ptr[r], end = ptr[r + 1], len = end - beg;
            if (len <= mthr) continue;
            if (!flags_host.empty() && flags_host[(size_t)(r >> 4)]) continue;      // block path owns it, whatever its length
            if (len <= thr) {
                Chunk c;
                c.beg = beg;
                c.end = end;
                c.slot = -1;
                c.row = r;
                chunks.push_back(c);
                ++n_medium;
                continue;
            }
            LongRow L;
            L.row = r;
            L.len = len;
            if (!h->split_long) {      // exact order: the hub kernel walks the row whole
                L.first_slot = -1;
                L.n_chunks = 0;
                longs.push_back(L);
                continue;
            }
            L.first_slot = n_slots;
            L.n_chunks = 0;
            for (int32_t b = beg; b < end; b += (end - b > clen ? clen : end - b)) {
                Chunk c;
                c.beg = b;
                c.end = (end - b > clen) ? b + clen : end;
                c.slot = n_slots++;
                c.row = r;
                chunks.push_back(c);
                ++L.n_chunks;
            }
            longs.push_back(L);
        }
        // hub rows longest first, ties in row order (as the device builder's stable radix sort leaves them)
        std::stable_sort(longs.begin(), longs.end(), [](const LongRow &x, const LongRow &y) { return x.len > y.len; });
        // longest first, stable: counting sort on the length (<= max(long_thr, long_chunk) by construction) -- unless the row order was asked for
        if (h->seg_order != 2) {
        int32_t lmax = 0;
        for (const Chunk &c : chunks) lmax = std::max(lmax, c.end - c.beg);
        std::vector<int32_t> start((size_t)lmax + 2, 0);
        for (const Chunk &c : chunks) ++start[(size_t)(lmax - (c.end - c.beg)) + 1];
        for (size_t i = 1; i < start.size(); ++i) start[i] += start[i - 1];
        std::vector<Chunk> sorted(chunks.size());
        for (const Chunk &c : chunks) sorted[(size_t)start[(size_t)(lmax - (c.end - c.beg))]++] = c;
        chunks.swap(sorted);
        }
    }
    h->n_chunks = (int32_t)chunks.size();
    h->n_long = (int32_t)longs.size();
    h->seg_nnz = 0;
    for (const Chunk &c : chunks) if (c.slot < 0) h->seg_nnz += c.end - c.beg;
    h->n_medium = n_medium;
    h->n_slots = n_slots;
    h->n_rows_for_rows_kernel = (int64_t)M - 16 * (int64_t)h->n_blk_groups - (int64_t)h->n_long - (int64_t)n_medium;
    h->ldp = ((int64_t)h->feat + 3) / 4 * 4;
    lap(3, tp);
    if (h->n_chunks > 0 || h->n_long > 0) {
        const size_t cb = chunks.size() * sizeof(Chunk), lb = longs.size() * sizeof(LongRow);
        const size_t pb = (size_t)n_slots * (size_t)h->ldp * sizeof(float);
        if (hipMalloc((void **)&h->d_chunks, cb ? cb : 16) != hipSuccess ||
            hipMalloc((void **)&h->d_long, lb ? lb : 16) != hipSuccess ||
            (pb && hipMalloc((void **)&h->d_partials, pb) != hipSuccess)) {
            free_plan(h);
            return MI_SPMM_ENOMEM;
        }
        hipError_t e = cb ? hipMemcpy(h->d_chunks, chunks.data(), cb, hipMemcpyHostToDevice) : hipSuccess;
        if (e == hipSuccess && lb) e = hipMemcpy(h->d_long, longs.data(), lb, hipMemcpyHostToDevice);
        if (e != hipSuccess) {
            free_plan(h);
            return (int)e;
        }
        h->ws_bytes = cb + lb + pb;
    }
    {
        const int brc = build_block_items(h);
        if (brc != 0) { free_plan(h); return brc; }
    }
    {
        const int crc = plan_col_strips(h);
        if (crc != 0) { free_plan(h); return crc; }
    }
    lap(4, tp);
    {
        int src = ensure_side_streams(h);
        if (src == 0) src = note_hub_rows(h);
        if (src != 0) { free_plan(h); return src; }
    }
    HIP_TRY(hipStreamSynchronize(0));      // (as in preprocess_on_gpu: build_col_strips is asynchronous)
    h->prepared = true;
    h->preprocess_us =
        std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    return MI_SPMM_OK;
}

}  // extern "C"

// ---- launch dispatch ------------------------------------------------------------
namespace {


// v2 rows kernel: the cache policy (nt C stores x nt (col,val) loads) and the workgroup size are
// instantiated for the 16-byte narrow path, which every benchmark shape uses; the dword and
// wide-address fallbacks get the default policy.
template <int LPR, int BT>
void launch_rows_v2_pol(int pol, const RowsArgs &a, dim3 grid, hipStream_t s)
{
    switch (pol & 3) {
    case 0: hipLaunchKernelGGL((spmm_rows_v2<4, LPR, 8, false, 0, BT>), grid, dim3(BT), 0, s, a); break;
    case 1: hipLaunchKernelGGL((spmm_rows_v2<4, LPR, 8, false, 1, BT>), grid, dim3(BT), 0, s, a); break;
    case 2: hipLaunchKernelGGL((spmm_rows_v2<4, LPR, 8, false, 2, BT>), grid, dim3(BT), 0, s, a); break;
    default: hipLaunchKernelGGL((spmm_rows_v2<4, LPR, 8, false, 3, BT>), grid, dim3(BT), 0, s, a); break;
    }
}
template <int LPR>
void launch_rows_v2_bt(int bt, int pol, const RowsArgs &a, dim3 grid, hipStream_t s)
{
    if (bt == 64) launch_rows_v2_pol<LPR, 64>(pol, a, grid, s);
    else if (bt == 128) launch_rows_v2_pol<LPR, 128>(pol, a, grid, s);
    else launch_rows_v2_pol<LPR, 256>(pol, a, grid, s);
}
template <int V, bool WIDE>
void launch_rows_v2_fixed(int lpr, const RowsArgs &a, dim3 grid, hipStream_t s)
{
    switch (lpr) {
    case 8: hipLaunchKernelGGL((spmm_rows_v2<V, 8, 8, WIDE, 1, 256>), grid, dim3(256), 0, s, a); break;
    case 16: hipLaunchKernelGGL((spmm_rows_v2<V, 16, 8, WIDE, 1, 256>), grid, dim3(256), 0, s, a); break;
    case 32: hipLaunchKernelGGL((spmm_rows_v2<V, 32, 8, WIDE, 1, 256>), grid, dim3(256), 0, s, a); break;
    default: hipLaunchKernelGGL((spmm_rows_v2<V, 64, 8, WIDE, 1, 256>), grid, dim3(256), 0, s, a); break;
    }
}
// "rows_unroll" = 16: sixteen gathers in flight per lane group (default cache policy, 256-thread workgroups only).  Half the round trips per row at two
// thirds of the occupancy: -13 % ... +23 % by graph (profiles/r05_rows_unroll_ab.txt), so never the rule's choice; "autotune" tries it
void launch_rows_v2_deep(int lpr, const RowsArgs &a, dim3 grid, hipStream_t s)
{
    switch (lpr) {
    case 8: hipLaunchKernelGGL((spmm_rows_v2<4, 8, 16, false, 1, 256>), grid, dim3(256), 0, s, a); break;
    case 16: hipLaunchKernelGGL((spmm_rows_v2<4, 16, 16, false, 1, 256>), grid, dim3(256), 0, s, a); break;
    case 32: hipLaunchKernelGGL((spmm_rows_v2<4, 32, 16, false, 1, 256>), grid, dim3(256), 0, s, a); break;
    default: hipLaunchKernelGGL((spmm_rows_v2<4, 64, 16, false, 1, 256>), grid, dim3(256), 0, s, a); break;
    }
}
void launch_rows_v2_any(bool vec4, bool wide, int lpr, int bt, int pol, const RowsArgs &a, dim3 grid, hipStream_t s)
{
    if (vec4 && !wide) {
        switch (lpr) {
        case 8: launch_rows_v2_bt<8>(bt, pol, a, grid, s); break;
        case 16: launch_rows_v2_bt<16>(bt, pol, a, grid, s); break;
        case 32: launch_rows_v2_bt<32>(bt, pol, a, grid, s); break;
        default: launch_rows_v2_bt<64>(bt, pol, a, grid, s); break;
        }
    } else if (vec4) launch_rows_v2_fixed<4, true>(lpr, a, grid, s);
    else if (wide) launch_rows_v2_fixed<1, true>(lpr, a, grid, s);
    else launch_rows_v2_fixed<1, false>(lpr, a, grid, s);
}

template <int V, int LPR, bool WIDE>
void launch_chunks(const ChunkArgs &a, dim3 grid, hipStream_t s, int deep)
{
    // deep = gathers in flight per lane group (16-byte narrow path; the fallbacks keep 8).  A segment is one
    // long dependent chain per lane group, so more loads in flight is what buys speed; 32 = a whole item.
    if (deep == 32 && V == 4 && !WIDE) hipLaunchKernelGGL((spmm_chunks<V, LPR, 32, false>), grid, dim3(kBlockThreads), 0, s, a);
    else if (deep == 16 && V == 4 && !WIDE) hipLaunchKernelGGL((spmm_chunks<V, LPR, 16, false>), grid, dim3(kBlockThreads), 0, s, a);
    else hipLaunchKernelGGL((spmm_chunks<V, LPR, 8, WIDE>), grid, dim3(kBlockThreads), 0, s, a);
}
template <int V, bool WIDE>
void launch_chunks_lpr(int lpr, const ChunkArgs &a, dim3 grid, hipStream_t s, int deep = 0)
{
    switch (lpr) {
    case 8: launch_chunks<V, 8, WIDE>(a, grid, s, deep); break;
    case 16: launch_chunks<V, 16, WIDE>(a, grid, s, deep); break;
    case 32: launch_chunks<V, 32, WIDE>(a, grid, s, deep); break;
    default: launch_chunks<V, 64, WIDE>(a, grid, s, deep); break;
    }
}

template <int G, bool WIDE, bool RUN>
void launch_block_items_g(int slab, const BlockArgs &a, dim3 grid, hipStream_t s, int bt)
{
    // slab = columns one wave covers: 256 / 128 / 64 as 4 / 2 / 1 chunks of 64 (16 bytes per lane), 32 as one chunk of
    // 32 (8 bytes per lane).  Shared items (G > 1) exist only for 256 and 128 (N % 128 == 0).
    if (slab == 256) hipLaunchKernelGGL((spmm_block_items<4, 4, G, WIDE, RUN>), grid, dim3(bt), 0, s, a);
    else if (slab == 128) hipLaunchKernelGGL((spmm_block_items<2, 4, G, WIDE, RUN>), grid, dim3(bt), 0, s, a);
    else if (G == 1 && slab == 64) hipLaunchKernelGGL((spmm_block_items<1, 4, 1, WIDE, RUN>), grid, dim3(bt), 0, s, a);
    else if (G == 1) hipLaunchKernelGGL((spmm_block_items<1, 2, 1, WIDE, RUN>), grid, dim3(bt), 0, s, a);
}
// cls: 0 = list items, 1 = run items holding one piece, 2 = run items sharing their B rows between two pieces
void launch_block_items(int slab, int cls, bool wide, const BlockArgs &a, dim3 grid, hipStream_t s, int bt)
{
    if (cls == 2) { if (wide) launch_block_items_g<2, true, true>(slab, a, grid, s, bt); else launch_block_items_g<2, false, true>(slab, a, grid, s, bt); }
    else if (cls == 1) { if (wide) launch_block_items_g<1, true, true>(slab, a, grid, s, bt); else launch_block_items_g<1, false, true>(slab, a, grid, s, bt); }
    else { if (wide) launch_block_items_g<1, true, false>(slab, a, grid, s, bt); else launch_block_items_g<1, false, false>(slab, a, grid, s, bt); }
}

void launch_small_step(int lpr, int deep, const SmallStepArgs &a, dim3 grid, hipStream_t s)
{
    // segments 16 gathers deep: 114 - 128 VGPRs beside the hub role's 112 (32 deep would make the whole kernel a 2-waves-per-SIMD kernel).  The 8-lane
    // instance is 3 dwords over 128 and spills them (12 bytes of scratch per lane); "segment_unroll" = 8 selects its 8-deep form, which does not.
    switch (lpr) {
    case 8:
        if (deep == 8) hipLaunchKernelGGL((spmm_small_step<8, 8>), grid, dim3(kBlockThreads), 0, s, a);
        else hipLaunchKernelGGL((spmm_small_step<8, 16>), grid, dim3(kBlockThreads), 0, s, a);
        break;
    case 16: hipLaunchKernelGGL((spmm_small_step<16, 16>), grid, dim3(kBlockThreads), 0, s, a); break;
    case 32: hipLaunchKernelGGL((spmm_small_step<32, 16>), grid, dim3(kBlockThreads), 0, s, a); break;
    default: hipLaunchKernelGGL((spmm_small_step<64, 16>), grid, dim3(kBlockThreads), 0, s, a); break;
    }
}

template <bool WIDE>
void launch_hub(int sw, bool excl, const HubArgs &a, dim3 grid, hipStream_t s)
{
    if (sw == 16 && excl) hipLaunchKernelGGL((spmm_hub<16, WIDE, true>), grid, dim3(64 * (1 + HubCfg<16>::L)), 0, s, a);
    else if (sw == 16) hipLaunchKernelGGL((spmm_hub<16, WIDE>), grid, dim3(64 * (1 + HubCfg<16>::L)), 0, s, a);
    else if (sw == 64) hipLaunchKernelGGL((spmm_hub<64, WIDE>), grid, dim3(64 * (1 + HubCfg<64>::L)), 0, s, a);
    else hipLaunchKernelGGL((spmm_hub<32, WIDE>), grid, dim3(64 * (1 + HubCfg<32>::L)), 0, s, a);
}

// Column-tile width of the rows / segment kernels when the caller leaves it to us (profiles/r02_wide_n_tiles.txt; round 5: profiles/r05_regret.md):
//   columns (almost) all near the row's own position (banded / mesh, >= 95 %): whole-wave tiles -- neighbouring rows share B rows through L2 and a
//     narrower tile only re-reads A (banded N=256: 1.75 ms at 256, 1.88 at 128, 2.15 at 64);
//   community orders (50 - 94 %: most of a row near the diagonal, the rest all over B): N >= 256 -> 128-column tiles, two rows per wavefront (round 5: six
//     community-ordered dataset shapes and the plain block model at N = 256: 0.82 - 0.95 of the time with 256; the others neutral); narrower N: the whole row;
//   otherwise (columns anywhere):
//     few nonzeros per row (mean < 8: youtube-, am-, arxiv-shaped; collab-shaped, 9.7, keeps 64: 0.87): the whole row / whole-wave tiles -- a narrower tile multiplies what a ROW costs
//       (row pointers, its lane group's start, its C store) and such a row has little else (round 5: youtube-shuffled N = 128 / 256 0.83 / 0.79,
//       arxiv-degree N = 256 0.74, am-degree N = 256 0.82 of the time with 64);
//     N >= 256: 64-column tiles -- the tiles are swept one after the other, so a sweep's B working set is K x 64 x 4 bytes (256 MiB at K = 2^20: the
//       Infinity Cache's size) instead of 1 GiB (uniform N=1024: 20.3 -> 19.4 ms, N=256: 5.44 -> 4.79; R-MAT N=512: 10.3 -> 8.3; dense-ish: +6 %);
//     N < 256 with hub rows (longest row > 64 x mean: R-MAT-like column reuse): 64 as well (R-MAT N=128: 2.09 -> 1.81 ms);
//     else the whole row (uniform N=128: 2.40 at 128, 2.42 at 64, 2.57 at 32).
int resolve_tile_cols(const mi_spmm_handle *h, int32_t N, int64_t ldb)
{
    (void)ldb;
    if (h->tile_cols > 0) return (int)h->tile_cols;
    if (h->local_pct < 0 || h->local_pct >= 95) return 256;
    if (h->local_pct >= 50) {
        if (N < 256) return 256;
        // (rows of a handful of nonzeros over a B far beyond the Infinity Cache -- youtube-, am-, wikikg2-community at N = 256: 0.91 - 0.94 with the whole wave)
        const int64_t mean_l = h->num_v > 0 ? h->nnz / h->num_v : 0;
        return (mean_l < 8 && 4.0 * (double)h->num_cols * (double)N > 256.0 * 1048576.0) ? 256 : 128;
    }
    // an L2-resident B (4 K N <= 6 MiB: ddi-shaped) has no working set to shrink: narrower tiles only re-read A (ddi-community N = 256: 0.76 with the whole wave)
    if (4.0 * (double)h->num_cols * (double)N <= 6.0 * 1048576.0) return 256;
    const int64_t mean = h->num_v > 0 ? h->nnz / h->num_v : 0;
    const bool hubs = (int64_t)h->max_row_nnz > 64 * (mean > 1 ? mean : 1);
    // (... except two tiles of 64 at N = 128 when the hubs come first in the vertex order: their B rows are the hot set, and half-width rows of it fit L2 --
    //  am-degree N = 128: 0.88 of the time with 64; the same graph at N = 256 prefers the whole wave by 0.82; arxiv-degree, whose B fits the Infinity Cache, 0.89 the other way)
    if (mean < 8) return (N <= 128 && N > 64 && hubs && h->front_pct >= 50 && 4.0 * (double)h->num_cols * (double)N > 256.0 * 1048576.0) ? 64 : 256;
    // (Hold-out graphs, round 5: sizing the tile so that a sweep's K x tile x 4 bytes just fit the Infinity Cache -- the whole wave once K <= 2^18 -- gains
    //  5 - 22 % on an R-MAT of scale 18 and LOSES 15 - 19 % on collab- / reddit- / protein-shaped graphs of the same K, whose strips or hub rows want the
    //  narrow tile: not adopted; profiles/r05_regret_holdout.md.)
    if (N >= 256) return 64;
    if (N >= 128 && hubs) return 64;
    return 256;
}

int pow2_ceil(int x)
{
    int p = 1;
    while (p < x) p <<= 1;
    return p;
}

}  // namespace

extern "C" {

// One column part [col0, col0 + N) of the step: d_vin / d_vout / partials already point at column col0.
// The MFMA block kernel works on the whole width and is launched with the first part only (full = the
// caller's original pointers and width).
struct FullView { const float *B; float *C; int32_t N; };

static int run_part(mi_spmm_handle *h, const float *d_vin, int64_t ldb, float *d_vout, int64_t ldc, int32_t row_begin,
                    int32_t row_end, hipStream_t s, int32_t N, int64_t col0, bool blocks_on, bool launch_blocks_here,
                    const FullView &full, int *launches_out, bool record, const PeerOut &po_full)
{
    // the extra destinations as this column part sees them (the hub and block kernels address the full width: po_full)
    PeerOut po = po_full;
    for (int q = 0; q < po.n; ++q) po.p[q] += col0;
    const int32_t M = h->num_v;
    // 16 bytes per lane whenever a row holds at least one float4: global dwordx4 accesses only need dword
    // alignment, and a row whose width is not a multiple of 4 gets its last lane shifted back to column N - 4
    // (spmm_kernels.hpp).  Only N < 4 takes the dword path.  The MFMA block path keeps the strict 16-byte rule.
    const bool vec4 = N >= 4;
    const int V = vec4 ? 4 : 1;
    // narrow addressing: byte offset of any B element < 2^32, column index and
    // row pitch in bytes < 2^24 (one v_mad_u32_u24 per gathered row)
    const bool wide = !((int64_t)h->num_cols <= (1 << 24) && ldb * 4 < (1 << 24) &&
                        ((int64_t)(h->num_cols > 0 ? h->num_cols - 1 : 0) * ldb + N) * 4 <= ((int64_t)1 << 32));
    int lpr = pow2_ceil((N + V - 1) / V);
    if (lpr < 8) lpr = 8;
    if (lpr > 64) lpr = 64;
    // Wide B (N >= 256): the column tiles are swept one after the other (blockIdx.x runs fastest), so a tile's
    // working set is K rows x tile_cols x 4 B.  Narrower tiles keep that set nearer the Infinity Cache's size.
    const int tile_cap = resolve_tile_cols(h, N, ldb) / V;
    if (vec4 && lpr > tile_cap) lpr = tile_cap;
    const int tile_w = lpr * V;
    const int col_tiles = (N + tile_w - 1) / tile_w;
    const int bt = (vec4 && !wide) ? (int)h->block_threads : kBlockThreads;
    const int gpb = bt / lpr;
    int rpb = (int)h->rows_per_block;
    // one row per lane group -- except narrow groups on very short rows (N <= 32, mean degree < 12), where a second
    // row per group lets the next row's pairs be fetched under the current row's few gathers (+3-6 %, profiles/r01_thresholds.txt)
    const bool short_rows = lpr == 8 && M > 0 && h->nnz / M < 12;
    if (rpb <= 0) rpb = short_rows ? 2 * gpb : gpb;
    if (rpb < gpb) rpb = gpb;
    int rpg = rpb / gpb;          // contiguous rows per lane group, at most LPR - 1
    if (rpg > lpr - 1) rpg = lpr - 1;
    if (rpg < 1) rpg = 1;
    rpb = rpg * gpb;
    const int64_t nblk64 = ((int64_t)(row_end - row_begin) + rpb - 1) / rpb;
    if (nblk64 > INT32_MAX || col_tiles > 65535) return MI_SPMM_EUNSUPPORTED;
    // auto: measured neutral-to-positive everywhere except one whole-wave row and a single column
    // tile (N = 256), where interleaving rows over the XCDs is ~3 % faster (profiles/r01_sweep_*)
    const bool remap = h->xcd_remap < 0 ? !(lpr == 64 && col_tiles == 1) : (h->xcd_remap != 0);
    const int flags = (remap ? kFlagXcdRemap : 0) | (h->ftz ? kFlagFtz : 0);
    const int pol = (h->nt_store ? kPolNtStore : 0) | (h->nt_stream ? kPolNtStream : 0);
    int launches = 0;

    // ---- small steps: one launch, three roles (spmm_kernels.hpp spmm_small_step) -------------------------------------------------------------
    // Eligible: the plain case of every role -- 16-byte lanes, narrow addressing, ONE column tile, no column strips, no split rows, no block groups, the
    // default workgroup size.  auto ("fused_step" = 2): only steps whose bytes take under 0.2 ms at 6 TB/s -- there the launch boundaries and the fork are
    // a third of the step and the rows role's lower occupancy (3 waves per SIMD, the hub role's footprint) costs nothing; longer steps keep their kernels.
    {
        const double step_bytes = (double)h->nnz * (4.0 * full.N + 8.0) + 4.0 * (double)h->num_v * full.N;
        const bool eligible = vec4 && !wide && col_tiles == 1 && launch_blocks_here && N == full.N && h->n_strips <= 1 && !h->split_long && h->n_blk_groups == 0 &&
                              bt == kBlockThreads && pol == kPolNtStore && (h->n_long > 0 || h->n_chunks > 0) &&
                              (int64_t)((full.N + 15) / 16) * h->n_long + (int64_t)h->n_chunks + nblk64 < (int64_t)INT32_MAX;
        // auto: the step's bytes take under 0.1 ms, or its longest row's chain (3.2 ns per nonzero) outlasts them anyway (am-shaped): then the rows role's
        // occupancy cannot matter.  (First rule, 0.2 ms: youtube-shaped kLen 32 -- 0.155 ms of bytes, a 75 us chain, 35 K rows workgroups -- lost 6 - 14 %.)
        // (an L2-resident B -- 4 K N <= 6 MiB, ddi-shaped -- moves its bytes three times faster: priced as in plan.hpp resolve_hub_threshold)
        const double t_bytes = step_bytes / (4.0 * (double)h->num_cols * (double)full.N <= 6.0 * 1048576.0 ? 18e12 : 6e12);
        // (local columns: the rows kernel alone is fast and wants its occupancy -- 0.1 ms; columns anywhere: latency-bound whatever the occupancy -- 0.16 ms:
        //  arxiv-community N = 128 loses 18 % fused, arxiv-rcm gains 16 %; youtube-community kLen 32 loses 14 %, youtube-shuffled gains 11 %)
        const double t_fused = h->local_pct >= 50 ? 100e-6 : 160e-6;
        // (a step that is ONE launch anyway -- one role only: ddi-shaped N = 256, every row a segment -- keeps that role's own kernel, which is the better
        //  instance of it: segments 32 gathers deep instead of 16 at the small-step kernel's footprint; ddi-community kLen 256: 85.7 -> 79 us)
        const int n_roles = ((h->n_long > 0 && range_has_hub(h, row_begin, row_end)) ? 1 : 0) + (h->n_chunks > 0 ? 1 : 0) + (h->n_rows_for_rows_kernel > 0 ? 1 : 0);
        const bool want = h->fused_step == 1 || (h->fused_step == 2 && n_roles >= 2 && (t_bytes < t_fused || (double)h->max_row_nnz * 3.2e-9 > t_bytes));
        h->last_fused = 0;
        if (eligible && want) {
            SmallStepArgs fa{};
            const bool hubs_here = h->n_long > 0 && range_has_hub(h, row_begin, row_end);
            if (hubs_here) {
                fa.h.rows = h->d_long; fa.h.row_ptr = h->d_ptr; fa.h.col_idx = h->d_idx; fa.h.vals = h->d_val;
                fa.h.B = full.B; fa.h.C = full.C; fa.h.ldb = ldb; fa.h.ldc = ldc; fa.h.n_hubs = h->n_long; fa.h.N = full.N;
                fa.h.slices = (full.N + 15) / 16; fa.h.row_lo = row_begin; fa.h.row_hi = row_end; fa.h.flags = h->ftz ? kFlagFtz : 0; fa.h.po = po_full;
                fa.hub_wgs = fa.h.slices * h->n_long;
            }
            if (h->n_chunks > 0) {
                fa.c.chunks = h->d_chunks; fa.c.col_idx = h->d_idx; fa.c.vals = h->d_val; fa.c.B = d_vin; fa.c.partials = nullptr; fa.c.C = d_vout;
                fa.c.ldb = ldb; fa.c.ldp = h->ldp; fa.c.ldc = ldc; fa.c.n_chunks = h->n_chunks; fa.c.N = N; fa.c.flags = flags;
                fa.c.row_lo = row_begin; fa.c.row_hi = row_end; fa.c.po = po;
                fa.seg_wgs = (h->n_chunks + kBlockThreads / lpr - 1) / (kBlockThreads / lpr);
            }
            fa.r.blk_flag = nullptr; fa.r.row_ptr = h->d_ptr; fa.r.col_idx = h->d_idx; fa.r.vals = h->d_val; fa.r.B = d_vin; fa.r.C = d_vout;
            fa.r.ldb = ldb; fa.r.ldc = ldc; fa.r.row0 = row_begin; fa.r.M = row_end; fa.r.N = N; fa.r.rows_per_block = rpg;
            fa.r.long_thr = (int32_t)h->medium_res; fa.r.nblk = (int)nblk64; fa.r.flags = flags; fa.r.po = po;
            const int rows_wgs = h->n_rows_for_rows_kernel > 0 ? (int)nblk64 : 0;       // every row may belong to the first two roles (ddi-shaped graphs)
            if (fa.hub_wgs + fa.seg_wgs + rows_wgs == 0) { *launches_out += 0; return MI_SPMM_OK; }
            // Which of the first two roles leads the grid (workgroups start in blockIdx order): the one whose longest chain LASTS longest.  A hub row costs
            // 3.2 ns per nonzero plus ~2 us of pipeline fill; a segment is one lane group's dependent gather chain, 47 ns per nonzero (30 out of an
            // L2-resident B), and the longest segment is as long as the hub threshold allows.  arxiv- / youtube- / am-shaped: one hub row of 10^4 - 10^5
            // nonzeros is the step -> hubs first; ddi- / collab-shaped: hub rows of 700 - 1 800 nonzeros (2 - 6 us) over 256 - 512-nonzero segments
            // (12 - 24 us) that used to START behind two rounds of hub workgroups -> segments first.
            {
                const bool l2_b = 4.0 * (double)h->num_cols * (double)full.N <= 6.0 * 1048576.0;
                const int64_t seg_max = h->long_thr > 0 && h->long_thr < (int64_t)h->max_row_nnz ? h->long_thr : (int64_t)h->max_row_nnz;
                const double t_seg = (double)seg_max * (l2_b ? 30e-9 : 47e-9), t_hub = 2e-6 + (double)h->max_row_nnz * 3.2e-9;
                const bool both = fa.hub_wgs > 0 && fa.seg_wgs > 0;
                fa.seg_first = both && (h->fused_order == 2 || (h->fused_order == 0 && t_seg > t_hub)) ? 1 : 0;
                h->last_seg_first = fa.seg_first;
            }
            dim3 fgrid((unsigned)(fa.hub_wgs + fa.seg_wgs + rows_wgs));
            launch_small_step(lpr, (int)h->segment_unroll, fa, fgrid, s);
            h->last_fused = 1;
            if (record) { h->last_wide = 0; h->last_lpr = lpr; h->last_v = V; }
            *launches_out += 1;
            return (int)hipGetLastError();
        }
    }

    // segment, hub, block and reduce kernels walk their whole tables and keep the rows of this call's range.
    // The hub kernel goes first (its longest row is the step's longest dependent chain) and, like the block kernel,
    // addresses the full width: it is launched with the first column part only.
    if (h->n_long > 0 && !h->split_long && launch_blocks_here && range_has_hub(h, row_begin, row_end)) {
        HubArgs ha{};
        ha.rows = h->d_long;
        ha.row_ptr = h->d_ptr;
        ha.col_idx = h->d_idx;
        ha.vals = h->d_val;
        ha.B = full.B;
        ha.C = full.C;
        ha.ldb = ldb;
        ha.ldc = ldc;
        ha.n_hubs = h->n_long;
        ha.N = full.N;
        ha.row_lo = row_begin;
        ha.row_hi = row_end;
        ha.flags = h->ftz ? kFlagFtz : 0;
        ha.po = po_full;
        // Slice width, when the caller leaves it to us: 32 columns per chain wave -- unless the longest row's chain alone
        // (3.3 ns per nonzero) is more than half of what the whole step's bytes take at 6 TB/s, i.e. that one chain is the
        // step: then 16, whose loaders put half as much through the LDS the chain wave reads from (am-shaped, N = 128:
        // 0.84 -> 0.72 ms; where the hub rows are many rather than one long, 32 is faster: R-MAT N = 32 0.47 vs 0.56;
        // profiles/r03_hub_experiments.txt)
        int sw = (int)h->hub_slice;
        bool excl = false;         // the chain-bound case: the hub workgroups keep their CUs to themselves (spmm_kernels.hpp EXCL)
        if (sw <= 0) {
            const double step_s = ((double)h->nnz * (4.0 * full.N + 8.0) + 4.0 * (double)h->num_v * full.N) / 6e12;
            // (round 4, new chain loop: 16 and 32 are within 1-3 % of each other on chain-bound graphs up to N = 128; at N = 256 the 16 workgroups
            //  of a row cost more than they save: am-shaped 1.164 -> 1.119 ms with 32; 64 loses everywhere it is the chain that counts)
            const bool chain_bound = full.N <= 128 && (double)h->max_row_nnz * 3.3e-9 > 0.5 * step_s;
            sw = (full.N <= 16 || chain_bound) ? 16 : 32;
            // ... and only while the hub workgroups are few: each keeps a whole CU, and 184 hubs x 8 slices of them (am-community N = 128) held the chip
            // until the last one had gone -- the rows kernel behind them started late: 0.43 -> 0.57 ms (profiles/r05_regret.md)
            excl = chain_bound && (int64_t)((full.N + 15) / 16) * h->n_long <= 128;
        }
        const bool wide_hub = !((int64_t)h->num_cols <= (1 << 24) && ldb * 4 < (1 << 24) &&
                                ((int64_t)(h->num_cols > 0 ? h->num_cols - 1 : 0) * ldb + full.N) * 4 <= ((int64_t)1 << 32));
        const int slices = (full.N + sw - 1) / sw;
        if ((int64_t)slices * h->n_long > INT32_MAX) return MI_SPMM_EUNSUPPORTED;
        ha.slices = slices;
        dim3 hgrid((unsigned)(slices * h->n_long));
        hipStream_t hs = s;
        {
            const int fr = side_stream(h, 0, s, &hs);
            if (fr != 0) return fr;
        }
        if (wide_hub) launch_hub<true>(sw, excl, ha, hgrid, hs); else launch_hub<false>(sw, excl, ha, hgrid, hs);
        ++launches;
    }
    if (h->n_chunks > 0) {
        ChunkArgs ca{};
        ca.chunks = h->d_chunks;
        ca.col_idx = h->d_idx;
        ca.vals = h->d_val;
        ca.B = d_vin;
        ca.partials = h->d_partials + col0;
        ca.C = d_vout;
        ca.ldb = ldb;
        ca.ldp = h->ldp;
        ca.ldc = ldc;
        ca.n_chunks = h->n_chunks;
        ca.N = N;
        ca.flags = flags;
        ca.row_lo = row_begin;
        ca.row_hi = row_end;
        ca.po = po;
        const int cgpb = kBlockThreads / lpr;  // the chunk kernel always runs 256-thread workgroups
        dim3 cgrid((h->n_chunks + cgpb - 1) / cgpb, col_tiles);
        // partial rows are ldp (multiple of 4) floats and hipMalloc-aligned, so only
        // the B side decides the vector width here
        // on side stream 1: a segment is one lane group's dependent chain (up to long_row_threshold nonzeros), and the rows
        // kernel behind it in the same stream would wait for the longest one
        hipStream_t cs = s;
        {
            const int fr = side_stream(h, 1, s, &cs);
            if (fr != 0) return fr;
        }
        // column strips (plan.hpp): one launch per strip in stream order, each over its own sub-segment table; every strip but the last
        // leaves the chains in the local C (plain stores, no extra destinations), the last one finishes them like an unstripped launch
        const int n_strips = (h->d_strips && h->n_strips > 1) ? h->n_strips : 1;
        for (int st = 0; st < n_strips; ++st) {
            if (n_strips > 1) {
                ca.chunks = h->d_strips + (size_t)st * (size_t)h->n_chunks;
                const bool last = st == n_strips - 1;
                ca.flags = flags | (last ? (po.n > 0 ? kFlagStripNoSkip : 0) : kFlagStripCarry);
                if (!last) ca.po.n = 0; else ca.po = po;
            }
            const int deep = h->segment_unroll > 0 ? (int)h->segment_unroll : (n_strips > 1 ? 16 : 32);
            if (vec4) { if (wide) launch_chunks_lpr<4, true>(lpr, ca, cgrid, cs); else launch_chunks_lpr<4, false>(lpr, ca, cgrid, cs, deep); }
            else { if (wide) launch_chunks_lpr<1, true>(lpr, ca, cgrid, cs); else launch_chunks_lpr<1, false>(lpr, ca, cgrid, cs); }
            ++launches;
        }
    }

    const bool remap_blocks = h->xcd_remap != 0;   // list is column-ordered: keep neighbours on one XCD
    // the block kernel addresses the FULL width (this part may be narrower when split_cols peels a remainder)
    const bool wide_full = !((int64_t)h->num_cols <= (1 << 24) && ldb * 4 < (1 << 24) &&
                             ((int64_t)(h->num_cols > 0 ? h->num_cols - 1 : 0) * ldb + full.N) * 4 <= ((int64_t)1 << 32));
    if (blocks_on && launch_blocks_here) {
        BlockArgs ba{};
        ba.col_idx = h->d_idx;
        ba.vals = h->d_val;
        ba.B = full.B;
        ba.C = full.C;
        ba.ldb = ldb;
        ba.ldc = ldc;
        ba.N = full.N;
        ba.remap = remap_blocks ? 1 : 0;
        ba.row_lo = row_begin;
        ba.row_hi = row_end;
        ba.po = po_full;
        const int slab = block_slab_width(full.N), slabs = full.N / slab;
        // pass p continues the fma chains pass p-1 left in C: stream order is the dependency
        for (int pass = 0; pass < h->n_blk_passes; ++pass) {
            for (int cls = 2; cls >= 0; --cls) {
                int32_t n = h->blk_launch[pass][cls].n;
                ba.items = h->d_blk_items + h->blk_launch[pass][cls].off;
                if (n == 0) continue;
                ba.n_items = n;
                // one item per wave, and ONE wave per workgroup by default: items differ 4x in length (64 or 128 rows, one or two
                // pieces), and a 4-wave workgroup holds its four wave slots until its longest item is done -- 1.38 of 2 possible
                // waves per SIMD were resident against 1.80 with single-wave workgroups (profiles/r03_c4_item_timeline.txt)
                const int wpw = (int)h->block_wg_waves;
                dim3 bgrid((unsigned)((n + wpw - 1) / wpw), slabs);
                launch_block_items(slab, cls, wide_full, ba, bgrid, s, 64 * wpw);
                ++launches;
            }
        }
    }

    RowsArgs a{};
    a.blk_flag = h->n_blk_groups > 0 ? h->d_blk_flag : nullptr;
    a.row_ptr = h->d_ptr;
    a.col_idx = h->d_idx;
    a.vals = h->d_val;
    a.B = d_vin;
    a.C = d_vout;
    a.ldb = ldb;
    a.ldc = ldc;
    a.row0 = row_begin;
    a.M = row_end;
    a.N = N;
    a.rows_per_block = rpg;
    // rows above the medium threshold were given to the segment or hub kernel -- except rows of block groups (of any
    // length up to the detection cap): when the block path cannot run on this call (pointers or pitches not 16-byte
    // aligned) the rows kernel takes exactly those rows as well (kFlagBlockFallback), nothing is computed twice
    const bool blocks_fallback = h->n_blk_groups > 0 && !blocks_on;
    a.long_thr = (int32_t)h->medium_res;
    a.nblk = (int)nblk64;
    a.flags = flags | (blocks_fallback ? kFlagBlockFallback : 0);
    a.po = po;
    dim3 grid((unsigned)nblk64, col_tiles);
    // Gathers in flight per lane group, auto: 8 -- except where the rows kernel carries LONG rows one wave per row: banded columns (the only plans that keep
    // rows of hundreds of nonzeros here: medium threshold 1 024), ONE whole-wave column tile (N = 193 ... 256: at N = 512 the second tile's wave shares the
    // row's B rows and 16 gains nothing) and a mean degree >= 256.  Then a row is one wave's chain of round trips and 16 halves them: banded 300-700 rows
    // 0.89 (band +-2048) / 0.90 (+-512), 600-1000 rows 0.93; mean degree 150 and below: 0.98 ... 1.06 (profiles/r05_banded_unroll_ab.jsonl)
    const bool deep_rows = h->rows_unroll == 16 || (h->rows_unroll == 0 && h->local_pct >= 95 && lpr == 64 && col_tiles == 1 && h->medium_res >= 512 &&
                                                    M > 0 && h->nnz / M >= 256);
    h->last_rows_deep = deep_rows && vec4 && !wide && bt == kBlockThreads && pol == kPolNtStore ? 1 : 0;
    // every row may already be owned by the segment, split and block paths: nothing left to launch
    const bool rows_needed = !((blocks_on || h->n_blk_groups == 0) && h->n_rows_for_rows_kernel == 0);
    if (!rows_needed) { /* skip */ }
    else if (deep_rows && vec4 && !wide && bt == kBlockThreads && pol == kPolNtStore) launch_rows_v2_deep(lpr, a, grid, s);
    else launch_rows_v2_any(vec4, wide, lpr, bt, pol, a, grid, s);
    if (rows_needed) ++launches;

    if (h->n_long > 0 && h->split_long) {
        ReduceArgs ra{};
        ra.rows = h->d_long;
        ra.partials = h->d_partials + col0;
        ra.C = d_vout;
        ra.ldp = h->ldp;
        ra.ldc = ldc;
        ra.n_long = h->n_long;
        ra.N = N;
        ra.flags = flags;
        ra.row_lo = row_begin;
        ra.row_hi = row_end;
        ra.po = po;
        const int64_t threads = (int64_t)h->n_long * ((N + V - 1) / V);
        dim3 rgrid((unsigned)((threads + kBlockThreads - 1) / kBlockThreads));
        hipStream_t cs = s;       // behind the segment kernel that produced the partial sums
        {
            const int fr = side_stream(h, 1, s, &cs);
            if (fr != 0) return fr;
        }
        if (vec4) hipLaunchKernelGGL((spmm_reduce_chunks<4>), rgrid, dim3(kBlockThreads), 0, cs, ra);
        else hipLaunchKernelGGL((spmm_reduce_chunks<1>), rgrid, dim3(kBlockThreads), 0, cs, ra);
        ++launches;
    }
    if (record) {
        h->last_wide = wide ? 1 : 0;
        h->last_lpr = lpr;
        h->last_v = V;
    }
    *launches_out += launches;
    return (int)hipGetLastError();
}

}  // extern "C"  (run_part is internal)

extern "C" {

static int launch_set(mi_spmm_handle *h, const float *d_vin, int64_t ldb, float *d_vout, int64_t ldc, int32_t row_begin,
                      int32_t row_end, const PeerOut &po, hipStream_t s);

int mi_spmm_run_rows(mi_spmm_handle *h, const float *d_vin, int64_t ldb, float *d_vout, int64_t ldc,
                     int32_t row_begin, int32_t row_end, void *stream)
{
    return mi_spmm_run_rows_multi(h, d_vin, ldb, d_vout, ldc, row_begin, row_end, 0, nullptr, stream);
}

int mi_spmm_run_rows_multi(mi_spmm_handle *h, const float *d_vin, int64_t ldb, float *d_vout, int64_t ldc,
                           int32_t row_begin, int32_t row_end, int32_t n_extra, float *const *d_extra, void *stream)
{
    if (!good(h) || !h->prepared) return MI_SPMM_ESTATE;
    if (n_extra < 0 || n_extra > kMaxPeerOut || (n_extra > 0 && !d_extra)) return MI_SPMM_EINVAL;
    PeerOut po;
    std::memset(&po, 0, sizeof(po));
    po.n = n_extra;
    for (int q = 0; q < n_extra; ++q) {
        if (!d_extra[q]) return MI_SPMM_EINVAL;
        po.p[q] = d_extra[q];
    }
    const int32_t M = h->num_v, N = h->feat;
    if (row_begin < 0 || row_end > M || row_begin > row_end) return MI_SPMM_EINVAL;
    if (row_begin == row_end || N == 0) return MI_SPMM_OK;
    if (!d_vout || ldc < N || ldb < N) return MI_SPMM_EINVAL;
    if (!d_vin && h->nnz > 0) return MI_SPMM_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    return launch_set(h, d_vin, ldb, d_vout, ldc, row_begin, row_end, po, s);
}

// The launch set of one step (or row panel) on stream s: what run() enqueues, directly or -- once -- under capture.
static int launch_set(mi_spmm_handle *h, const float *d_vin, int64_t ldb, float *d_vout, int64_t ldc, int32_t row_begin,
                      int32_t row_end, const PeerOut &po, hipStream_t s)
{
    const int32_t N = h->feat;
    const int n_extra = po.n;
    float *const *d_extra = po.p;
    bool aligned16 = (N % 4 == 0) && (ldb % 4 == 0) && (ldc % 4 == 0) &&
                     (((uintptr_t)d_vin | (uintptr_t)d_vout) & 15u) == 0;
    for (int q = 0; q < n_extra; ++q) aligned16 = aligned16 && ((uintptr_t)d_extra[q] & 15u) == 0;
    const bool blocks_on = h->n_blk_groups > 0 && aligned16;
    const FullView full = {d_vin, d_vout, N};
    // Widths just above a multiple of 256: the last 256-column tile would hold only a few columns yet run one
    // row per wave.  Up to 64 such columns go to a second set of launches with their own, narrower lane groups:
    // N = 257 / 260: -7 % time; wider remainders measured neutral, so they stay tiles of the one launch
    // (profiles/r01_odd_widths.txt -- the partial tile is bound by the extra cache lines, not by instruction issue).
    int32_t rem = (h->split_cols && N > 256) ? N % 256 : 0;
    if (rem > 64) rem = 0;
    int launches = 0;
    h->fork_recorded = h->forked[0] = h->forked[1] = false;
    int rc = run_part(h, d_vin, ldb, d_vout, ldc, row_begin, row_end, s, N - rem, 0, blocks_on, true, full, &launches, true, po);
    if (rc == 0 && rem > 0)
        rc = run_part(h, d_vin + (N - rem), ldb, d_vout + (N - rem), ldc, row_begin, row_end, s, rem, N - rem, blocks_on,
                      false, full, &launches, false, po);
    for (int i = 0; i < 2; ++i) {      // join: whatever the caller enqueues next is ordered after the side streams' rows too
        if (!h->forked[i]) continue;
        h->forked[i] = false;
        hipError_t je = hipEventRecord(h->ev_join[i], h->side[i]);
        if (je == hipSuccess) je = hipStreamWaitEvent(s, h->ev_join[i], 0);
        if (rc == 0 && je != hipSuccess) rc = (int)je;
    }
    h->last_launches = launches;
    return rc;
}


// "autotune": time the step under the plan in force (nothing else: no allocation beyond two events).  One warm-up, then enough runs for ~1 ms.
static int time_step(mi_spmm_handle *h, const float *d_vin, float *d_vout, hipEvent_t e0, hipEvent_t e1, double *ms_out)
{
    PeerOut po;
    std::memset(&po, 0, sizeof(po));
    const double est_ms = ((double)h->nnz * (4.0 * h->feat + 8.0) + 4.0 * (double)h->num_v * h->feat) / 8e9;
    int reps = (int)(1.0 / (est_ms > 0.01 ? est_ms : 0.01)) + 1;
    reps = reps < 3 ? 3 : (reps > 20 ? 20 : reps);
    double best = 1e30;
    for (int batch = 0; batch < 2; ++batch) {                  // (min of two batches; the first also warms the plan's tables up)
        int rc = launch_set(h, d_vin, h->feat, d_vout, h->feat, 0, h->num_v, po, nullptr);
        if (rc != MI_SPMM_OK) return rc;
        HIP_TRY(hipEventRecord(e0, nullptr));
        for (int i = 0; i < reps && rc == MI_SPMM_OK; ++i) rc = launch_set(h, d_vin, h->feat, d_vout, h->feat, 0, h->num_v, po, nullptr);
        if (rc != MI_SPMM_OK) return rc;
        HIP_TRY(hipEventRecord(e1, nullptr));
        HIP_TRY(hipEventSynchronize(e1));
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
        if ((double)ms / reps < best) best = (double)ms / reps;
    }
    *ms_out = best;
    return MI_SPMM_OK;
}

static int autotune_plan(mi_spmm_handle *h, const float *d_vin, float *d_vout)
{
    hipEvent_t e0 = nullptr, e1 = nullptr;
    HIP_TRY(hipEventCreate(&e0));
    if (hipEventCreate(&e1) != hipSuccess) { (void)hipEventDestroy(e0); return MI_SPMM_ENOMEM; }
    struct Cfg { int64_t tile, strips, mthr, fused, order, unroll, thr; };
    auto apply = [&](const Cfg &c) { h->tile_cols = c.tile; h->col_strips = c.strips; h->medium_thr = c.mthr; h->fused_step = c.fused; h->seg_order = c.order; h->rows_unroll = c.unroll; h->long_thr_user = c.thr; };
    // what the caller left to us (an explicit value of the caller's is never touched); the plan of these settings is the one in force
    const bool own_tile = h->tile_cols == 0, own_strips = h->col_strips == 0, own_mthr = h->medium_thr == 0, own_fused = h->fused_step == 2, own_order = h->seg_order == 0;
    const bool own_unroll = h->rows_unroll == 0, own_thr = h->long_thr_user == 0 && !h->split_long && h->feat >= 4;
    Cfg best = {h->tile_cols, h->col_strips, h->medium_thr, h->fused_step, h->seg_order, h->rows_unroll, h->long_thr_user};
    double best_ms = 0.0;
    int rc = time_step(h, d_vin, d_vout, e0, e1, &best_ms);
    h->tune_auto_ms = best_ms;
    h->tune_evals = 0;
    const int32_t N = h->feat, S_auto = h->n_strips, tile_auto = h->last_lpr * 4, mthr_auto = (int32_t)h->medium_res;
    const bool fused_auto = h->last_fused != 0, rows_deep_auto = h->last_rows_deep != 0;
    const int64_t thr_auto = h->long_thr;        // the resolved hub threshold of the auto plan (1 << 30: the hub fold; 8192 with no row above it: no hubs)
    auto consider = [&](Cfg c) {
        if (rc != MI_SPMM_OK) return;
        apply(c);
        rc = preprocess_plan(h);
        double ms = 0.0;
        if (rc == MI_SPMM_OK) rc = time_step(h, d_vin, d_vout, e0, e1, &ms);
        if (rc != MI_SPMM_OK) return;
        ++h->tune_evals;
        if (ms < 0.97 * best_ms) { best = c; best_ms = ms; }    // 3 %: a candidate has to beat the noise of a 1 ms measurement
    };
    // one option at a time, each sweep starting from the best so far (the options interact little once the plan class is fixed: scripts/regret.py)
    // the hub threshold one notch down and one up (a power-of-two ladder, 256 ... 8192): half of auto is 5 - 10 % faster on yelp- / youtube-shaped graphs and an
    // R-MAT of scale 18, 4 - 15 % slower on ddi- / protein-shaped ones and an R-MAT of scale 20 -- no plan statistic separates them (profiles/r05_regret*.md)
    if (own_thr && thr_auto >= 256 && thr_auto <= 8192 && h->max_row_nnz > 256)
        for (int64_t t : {thr_auto / 2, thr_auto * 2}) if (t >= 256 && t <= 8192 && (t < thr_auto || h->max_row_nnz > thr_auto)) { Cfg c = best; c.thr = t; consider(c); }
    if (own_mthr) for (int64_t m : {32, 64, 256, 1024}) if (m != mthr_auto) { Cfg c = best; c.mthr = m; consider(c); }
    if (own_strips) {
        const int64_t cand[3] = {1, S_auto > 1 ? (S_auto / 2 > 1 ? S_auto / 2 : 2) : 4, S_auto > 1 ? (2 * S_auto < kMaxColStrips ? 2 * S_auto : kMaxColStrips) : 12};
        for (int64_t sc : cand) if (sc != S_auto) { Cfg c = best; c.strips = sc; consider(c); }
    }
    if (own_tile && N > 64) for (int64_t t : {64, 128, 256}) if (t != tile_auto && (t < N || t == 256) && !(t == 256 && tile_auto >= N)) { Cfg c = best; c.tile = t; consider(c); }
    if (own_fused) { Cfg c = best; c.fused = fused_auto ? 0 : 1; consider(c); }
    if (own_order && h->n_chunks > 0) { Cfg c = best; c.order = 2; consider(c); }        // segments as the rows come (auto: by length)
    // 16 gathers in flight per lane group of the rows kernel (auto: 8): -13 % on banded 300-700 rows at N = 256, -8 % on rows of 4 - 12 at N = 128 / 256,
    // -1 % on C1 / C2, +4 ... +23 % on community-ordered graphs (occupancy): no rule of the plan's statistics separates them (profiles/r05_rows_unroll_ab.txt)
    if (own_unroll && !fused_auto) { Cfg c = best; c.unroll = rows_deep_auto ? 8 : 16; consider(c); }      // (the other depth than the rule's)
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (rc != MI_SPMM_OK) { apply({own_tile ? 0 : h->tile_cols, own_strips ? 0 : h->col_strips, own_mthr ? 0 : h->medium_thr, own_fused ? 2 : h->fused_step, own_order ? 0 : h->seg_order, own_unroll ? 0 : h->rows_unroll, own_thr ? 0 : h->long_thr_user}); return rc; }
    apply(best);
    h->tuned_mask = (own_tile && best.tile != 0 ? 1u : 0u) | (own_strips && best.strips != 0 ? 2u : 0u) | (own_mthr && best.mthr != 0 ? 4u : 0u) |
                    (own_fused && best.fused != 2 ? 8u : 0u) | (own_order && best.order != 0 ? 16u : 0u) | (own_unroll && best.unroll != 0 ? 32u : 0u) | (own_thr && best.thr != 0 ? 64u : 0u);
    h->tune_best_ms = best_ms;
    return preprocess_plan(h);              // the winner's plan (also when the winner is the auto plan: the last candidate's tables are in place otherwise)
}

int mi_spmm_preprocess(mi_spmm_handle *h, const float *d_vin, float *d_vout)
{
    // the reference zeroes vout here (spmm_opt.cu:67-68) because its kernel accumulates; ours overwrites, so vout is left alone -- unless "autotune" is on
    if (good(h) && h->tuned_mask) {         // what an earlier tuning set goes back to auto: tuned again below, or simply auto when the option is off now
        if (h->tuned_mask & 1u) h->tile_cols = 0;
        if (h->tuned_mask & 2u) h->col_strips = 0;
        if (h->tuned_mask & 4u) h->medium_thr = 0;
        if (h->tuned_mask & 8u) h->fused_step = 2;
        if (h->tuned_mask & 16u) h->seg_order = 0;
        if (h->tuned_mask & 32u) h->rows_unroll = 0;
        if (h->tuned_mask & 64u) h->long_thr_user = 0;
        h->tuned_mask = 0;
    }
    const int rc = preprocess_plan(h);
    if (rc != MI_SPMM_OK) return rc;
    if (h->autotune && d_vin && d_vout && h->num_v > 0 && h->feat >= 4 && h->nnz > 0 && !h->split_long) {
        const auto t0 = std::chrono::steady_clock::now();
        const int trc = autotune_plan(h, d_vin, d_vout);
        if (trc != MI_SPMM_OK) return trc;
        h->preprocess_us += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    }
    return MI_SPMM_OK;
}

int mi_spmm_run_ld(mi_spmm_handle *h, const float *d_vin, int64_t ldb, float *d_vout, int64_t ldc,
                   void *stream)
{
    if (!good(h)) return MI_SPMM_ESTATE;
    return mi_spmm_run_rows(h, d_vin, ldb, d_vout, ldc, 0, h->num_v, stream);
}

int mi_spmm_run(mi_spmm_handle *h, const float *d_vin, float *d_vout, void *stream)
{
    if (!good(h)) return MI_SPMM_ESTATE;
    return mi_spmm_run_rows(h, d_vin, h->feat, d_vout, h->feat, 0, h->num_v, stream);
}

static int compare_common(const void *a, const void *b, int64_t n, int mode, int64_t *count_out,
                          float *maxabs_out, void *stream)
{
    if (!count_out || n < 0) return MI_SPMM_EINVAL;
    *count_out = 0;
    if (maxabs_out) *maxabs_out = 0.f;
    if (n == 0) return MI_SPMM_OK;
    if (!a || !b) return MI_SPMM_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    unsigned long long *d = nullptr;  // [0] count, [1] max bits
    HIP_TRY(hipMalloc((void **)&d, 256));
    hipError_t e = hipMemsetAsync(d, 0, 16, s);
    if (e == hipSuccess) {
        const int64_t want = (n + kBlockThreads * 4 - 1) / (kBlockThreads * 4);
        const int grid = (int)(want < 1 ? 1 : (want > 8192 ? 8192 : want));
        hipLaunchKernelGGL(compare_kernel, dim3(grid), dim3(kBlockThreads), 0, s, a, b, n, mode, d,
                           (unsigned int *)(d + 1));
        e = hipGetLastError();
    }
    unsigned long long host[2] = {0, 0};
    if (e == hipSuccess) e = hipMemcpyAsync(host, d, 16, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    (void)hipFree(d);
    if (e != hipSuccess) return (int)e;
    *count_out = (int64_t)host[0];
    if (maxabs_out) {
        const unsigned int bits = (unsigned int)host[1];
        std::memcpy(maxabs_out, &bits, 4);
    }
    return MI_SPMM_OK;
}

int mi_spmm_valid_float(const float *d_y, const float *d_y2, int64_t n, int64_t *bad_out, void *stream)
{
    return compare_common(d_y, d_y2, n, 0, bad_out, nullptr, stream);
}

int mi_spmm_valid_int(const int32_t *d_y, const int32_t *d_y2, int64_t n, int64_t *bad_out, void *stream)
{
    return compare_common(d_y, d_y2, n, 1, bad_out, nullptr, stream);
}

int mi_spmm_count_bitdiff(const float *d_a, const float *d_b, int64_t n, int64_t *ndiff_out,
                          float *maxabs_out, void *stream)
{
    return compare_common(d_a, d_b, n, 2, ndiff_out, maxabs_out, stream);
}

int mi_spmm_stream_create_concurrent(void **stream_out, int high_priority, int *overlaps_out)
{
    if (!stream_out) return MI_SPMM_EINVAL;
    *stream_out = nullptr;
    int lo = 0, hi = 0;
    HIP_TRY(hipDeviceGetStreamPriorityRange(&lo, &hi));      // hi = numerically lowest = highest priority
    hipStream_t s = nullptr;
    int ov = 0;
    const int rc = concurrent_stream(&s, high_priority ? hi : 0, &ov);
    if (rc != MI_SPMM_OK) return rc;
    *stream_out = (void *)s;
    if (overlaps_out) *overlaps_out = ov;
    return MI_SPMM_OK;
}

int mi_spmm_fill_normal(float *d_out, int64_t n, uint64_t seed, uint64_t subsequence, float mean, float stddev,
                        void *stream)
{
    if (n < 0) return MI_SPMM_EINVAL;
    if (n == 0) return MI_SPMM_OK;
    if (!d_out) return MI_SPMM_EINVAL;
    const int64_t want = ((n + 3) / 4 + kBlockThreads - 1) / kBlockThreads;
    const int grid = (int)(want < 1 ? 1 : (want > 65536 ? 65536 : want));
    hipLaunchKernelGGL(fill_normal_kernel, dim3(grid), dim3(kBlockThreads), 0, (hipStream_t)stream, d_out, n, seed,
                       subsequence, mean, stddev);
    return (int)hipGetLastError();
}

int mi_spmm_fill_philox_u32(uint32_t *d_out, int64_t n, uint64_t seed, uint64_t subsequence, void *stream)
{
    if (n < 0) return MI_SPMM_EINVAL;
    if (n == 0) return MI_SPMM_OK;
    if (!d_out) return MI_SPMM_EINVAL;
    const int64_t want = ((n + 3) / 4 + kBlockThreads - 1) / kBlockThreads;
    const int grid = (int)(want < 1 ? 1 : (want > 65536 ? 65536 : want));
    hipLaunchKernelGGL(fill_philox_kernel, dim3(grid), dim3(kBlockThreads), 0, (hipStream_t)stream, d_out, n, seed,
                       subsequence);
    return (int)hipGetLastError();
}

int mi_spmm_unpack_gathered(const float *d_staging, float *d_C, int64_t rows, int32_t n_ranks,
                            int32_t n_loc, int64_t ldc, void *stream)
{
    if (rows < 0 || n_ranks < 1 || n_loc < 0 || ldc < (int64_t)n_ranks * n_loc) return MI_SPMM_EINVAL;
    if (rows == 0 || n_loc == 0) return MI_SPMM_OK;
    if (!d_staging || !d_C) return MI_SPMM_EINVAL;
    const bool vec4 = (n_loc % 4 == 0) && (ldc % 4 == 0) &&
                      (((uintptr_t)d_staging | (uintptr_t)d_C) & 15u) == 0;
    const int V = vec4 ? 4 : 1;
    const int64_t total = rows * (int64_t)n_ranks * (n_loc / V);
    const int64_t want = (total + kBlockThreads - 1) / kBlockThreads;
    const int grid = (int)(want < 1 ? 1 : (want > 16384 ? 16384 : want));
    hipStream_t s = (hipStream_t)stream;
    if (vec4) hipLaunchKernelGGL((unpack_gathered<4>), dim3(grid), dim3(kBlockThreads), 0, s, d_staging, d_C, rows, n_ranks, n_loc, ldc);
    else hipLaunchKernelGGL((unpack_gathered<1>), dim3(grid), dim3(kBlockThreads), 0, s, d_staging, d_C, rows, n_ranks, n_loc, ldc);
    return (int)hipGetLastError();
}

}  // extern "C"
