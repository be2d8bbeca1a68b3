// plan.hpp -- what preprocess produces for run(): the segment table (medium rows + pieces of split
// rows, longest first), the split-row table, the compacted list of block-path groups.  Built either
// on the GPU (preprocess_gpu.hip, default) or by the reference-style host loop (mi_spmm.hip,
// "gpu_preprocess" = 0, kept as the cross-check).  Internal to libmi_spmm.so.
#pragma once
#include <stddef.h>
#include <stdint.h>

#include "plan_types.hpp"

namespace mi {

// ---- auto hub threshold of the default (exact-order) mode ------------------------------------------------------
// Two kernels take rows in stored order: the segment kernel (one lane group per row, 32 gathers in flight: 47 ns per
// nonzero of ONE row on an idle chip, 140-430 ns beside a rows kernel that saturates the fabric; but thousands of rows at
// once, i.e. full memory throughput) and the hub kernel (3.2 ns per nonzero of one row, about half the segment kernel's
// throughput).  So: a row becomes a hub when, as a segment on its side stream, it could no longer hide behind the rest of
// the step -- L x (100 + 1.3 N) ns > half the step's estimated time (gather-model bytes at 6 TB/s) -- unless the rows above that
// length hold more than a quarter of all nonzeros: then the hub kernel would carry the step at its lower throughput, and the
// threshold moves up until they do not (ddi-shaped graphs: every row is long) -- as long as a segment of that length still fits
// inside the step (below).  Candidates 256 .. 8192, powers of two.
// Measured against fixed thresholds on eight graph shapes x three widths: profiles/r03_hub_thresholds.txt.
// Round 5 (profiles/r05_regret.md, ddi-community kLen 128 / 256: regret 16 / 33 %): when B is L2-RESIDENT (4 K N <= 6 MiB: a gather is an L2 hit, not a
// trip to the fabric) the step runs at the L2's gather rate (18 TB/s) and a segment's nonzero costs ~30 ns beside the others, ~16 ns alone -- the
// constants above price such a step 3x too long and its segments 5x too slow, and rows went to the hub kernel that were faster as segments.
// The histogram comes from the same pass over row_ptr that finds the longest row; both plan builders use this function.
constexpr int kHistN = 6;
__host__ __device__ inline int32_t hist_threshold(int i) { return 256 << i; }
struct LenHist {
    uint32_t cnt[kHistN];             // rows longer than hist_threshold(i)
    unsigned long long nnz[kHistN];   // ... and the nonzeros they hold
};
__host__ __device__ inline int32_t resolve_hub_threshold(int64_t nnz, int32_t M, int32_t K, int32_t N, const unsigned long long *nnz_above, int32_t max_len)
{
    const double bytes = (double)nnz * (4.0 * N + 8.0) + 4.0 * (double)M * N;
    const bool resident = 4.0 * (double)K * (double)N <= 6.0 * 1048576.0;
    const double step = bytes / (resident ? 18e12 : 6e12);
    // a segment's cost per nonzero beside a saturating rows kernel grows with the bytes per gather: ~140 ns at N = 32, ~270 at 128,
    // ~430 at 256 (am-shaped N = 256: spmm_chunks 0.88 ms for 2 048-nonzero rows, profiles/r03b_am_kernel_stats.csv)
    const double seg_ns = resident ? 30.0 : 100.0 + 1.3 * (double)(N < 256 ? N : 256);
    const double idle_ns = resident ? 16.0 : 47.0;
    const double t = 0.5 * step / (seg_ns * 1e-9);
    // the longest row itself hides behind the step as a segment: nothing needs the hub kernel (the candidates are powers of two, and the largest one
    // below t used to send rows between it and t to the hub kernel all the same -- ddi-community kLen 256: 264 rows, 33 % regret)
    if ((double)max_len <= t) return hist_threshold(kHistN - 1);
    int i = 0;
    while (i + 1 < kHistN && (double)hist_threshold(i + 1) <= t) ++i;
    const int i_lat = i;
    while (i + 1 < kHistN && (double)nnz_above[i] > 0.25 * (double)nnz) ++i;
    // ... but not so far up that ONE segment of that length, even on an idle chip (47 ns per nonzero), outlasts the whole step's
    // estimate: a small matrix of long rows (ddi-shaped, N = 32: 4 267 rows, 44 us of bytes, an 1 772-nonzero row = 83 us as a
    // segment) is then latency-bound whatever the hub kernel's throughput, and the shorter chain wins (69 -> 42 us)
    while (i > i_lat && (double)hist_threshold(i) * idle_ns * 1e-9 > step) --i;
    return hist_threshold(i);
}

// ---- auto medium threshold (rows above it leave the rows kernel for the segment kernel) -----------------------------------------------------
// 64 when the degrees are even (the rows kernel's lane groups finish together anyway), 32 when the longest row is more than 8x the mean -- skewed graphs,
// where neighbours in a wave differ widely (profiles/r01_medium_threshold.txt).  Round 5: that was measured on uniformly random columns only.  Where the
// columns are LOCAL (community / mesh order: >= 50 % of the sampled nonzeros near their row's own position) neighbouring rows gather the same B rows and
// meet in one L2 in the rows kernel, which walks the rows in order and many at a time; a row that leaves it loses that company.  So, on local orders:
//   N >= 128 (at most two rows per wavefront: a long row next to a short one idles little), a graph large enough to fill the chip with rows (M >= 65 536) and
//     a step long enough not to be latency-bound (>= 0.2 ms of bytes at 6 TB/s): the rows kernel keeps rows up to 512 nonzeros (profiles/r05_regret.md:
//     ppa- / products- / yelp- / citation-community 64 -> 0.85, 128 -> 0.72 - 0.77, 512 -> a further 0.91 - 0.95 of the time; ddi- (4 267 rows) and
//     arxiv-community (0.1 ms steps) LOSE 20 - 45 % with it: few rows, each needing the segment kernel's 32 gathers in flight);
//   N < 128 (eight or four rows per wavefront: a long row idles its neighbours' lanes): four times the mean degree, at most 256 -- on graphs whose ROWS ARE
//     long (ppa-, products-, protein-shaped: mean 48 - 600) a threshold of 32 sends nearly every row away (ppa-community kLen 32: 256 -> 0.79), on graphs of
//     short rows with a few long ones (collab-, youtube-, yelp-shaped: mean 5 - 19) the old rule stays (256 there loses 22 - 34 %).
// Non-local orders of the same graphs: 64 and 128 LOSE 16 - 38 % against 32.
__host__ __device__ inline int32_t resolve_medium_threshold(int32_t mthr_user, int64_t nnz, int32_t M, int32_t max_len, int32_t thr, int32_t local_pct, int32_t N)
{
    const int64_t mean_len = M > 0 ? nnz / M : 0;
    const int32_t old_rule = (int64_t)max_len > 8 * (mean_len > 1 ? mean_len : 1) ? 32 : 64;
    int32_t m = old_rule;
    if (mthr_user > 0) m = mthr_user;
    else if (M < 65536) { /* too few rows to fill the chip with lane groups: every long row needs the segment kernel's 32 gathers in flight (ddi-shaped) */ }
    else if (local_pct >= 95) m = 1024;       // banded / mesh: every row's B rows are its neighbours' -- the rows kernel keeps all but the longest (banded rows of
                                              // 300 - 700 nonzeros: 0.47 - 0.73 of the time at every width against 64 / 512)
    else if (local_pct >= 50) {
        const double step = ((double)nnz * (4.0 * N + 8.0) + 4.0 * (double)M * N) / 6e12;
        if (N >= 128) { if (step >= 200e-6) m = 512; }
        else {
            const int64_t by_mean = 4 * mean_len;
            if (by_mean > m) m = (int32_t)(by_mean < 256 ? by_mean : 256);
        }
    }
    return m < thr ? m : thr;
}

struct PlanOut {
    Chunk *d_chunks = nullptr;      // [n_chunks], sorted by length descending (stable in row order)
    LongRow *d_long = nullptr;      // [n_long], longest first (stable in row order)
    int32_t *d_blk_groups = nullptr;  // [n_blk_groups] (only when d_blk_flag was given)
    int32_t n_chunks = 0, n_long = 0, n_slots = 0, n_medium = 0, n_blk_groups = 0;
    int32_t max_len = 0;
    int32_t mthr = 0;               // resolved medium threshold (the rows kernel skips rows above it)
    int32_t thr = 0;                // resolved hub threshold (the caller's value, or the auto rule above)
    int32_t local_pct = 0;          // sampled nonzeros within a window of their row's own position, percent (column-tile, medium-threshold and strip rules)
    int32_t front_pct = 0;          // sampled nonzeros whose column lies in the first quarter of the columns, percent (uniform: 25; hubs-first vertex orders: 50+)
    int64_t seg_nnz = 0;            // nonzeros in the whole (exact) segments: lets the column-strip rule pick S before any column is read
};

// Returns 0, a negative MI_SPMM_E* code (malformed CSR, out of memory) or a positive hipError_t.
// d_blk_flag: per 16-row group, 1 = block path owns it (nullable).  col_bad: device flag written
// by csr_check_cols earlier on the same (null) stream; read back with the same single copy.
// split: 1 = hubs are cut into pieces of clen (summed piece by piece), 0 = hubs are only listed (spmm_hub keeps their order).
// mthr: medium threshold, 0 = auto (resolved on the device from the longest row, returned in PlanOut::mthr).
// A grow-only device arena owned by the handle: preprocess temporaries are carved out of it.
struct Scratch {
    void *p = nullptr;
    size_t cap = 0;
};
int scratch_reserve(Scratch *s, size_t bytes);   // 0 or MI_SPMM_ENOMEM; contents are lost when it grows
void scratch_release(Scratch *s);

// thr: hub threshold, 0 = auto (exact mode only; resolved on the device from the row-length histogram, returned in PlanOut::thr).
// N: dense width (the auto rule's byte estimate).
// the column sample on its own (the host plan builder has no device pass of its own over the rows): synchronises
int sample_columns_gpu(const int32_t *d_row_ptr, const int32_t *d_col_idx, int32_t M, int32_t K, int64_t nnz, void *d_scratch256, int32_t *local_pct, int32_t *front_pct);
// seg_order: 1 = segments longest first (stable), 2 = in row order (no sort)
int build_plan_gpu(const int32_t *d_row_ptr, const int32_t *d_col_idx, int32_t M, int32_t K, int32_t N, int64_t nnz,
                   const uint8_t *d_blk_flag, const unsigned int *d_col_bad, int32_t mthr, int32_t thr, int32_t clen, int32_t split, int32_t seg_order,
                   Scratch *sa, Scratch *sb, PlanOut *out);

// ---- column strips of the exact segments --------------------------------------------------------------------------
// A graph of long rows over few columns (protein-, reddit-shaped: hundreds of nonzeros per row, B a few tens of MiB) gathers out of a B that
// the Infinity Cache holds but an XCD's 4 MiB L2 does not.  Cut every segment at S - 1 column boundaries (columns ascending inside a row:
// checked) and run strip after strip: a launch then gathers out of K / S rows of B.  survey_segments reads the segments' columns once (are
// they ascending? how many nonzeros do the segments hold?); build_col_strips writes the S sub-segment tables (strip-major, each in the
// segment table's order; strip 0 starts the chains, later strips continue them through C: plan_types.hpp kSlotContinue).
struct SegmentSurvey {
    uint32_t unsorted;              // segments with a column smaller than its predecessor
    uint32_t pad;
    unsigned long long nnz;         // nonzeros in all segments
};
constexpr int kMaxColStrips = 64;   // "col_strips" accepts 2 .. 64; the auto rule stops at 32
// Round 5, the default builder: survey and tables in one pass over the segments' columns (S chosen beforehand from PlanOut::seg_nnz).  out->unsorted > 0:
// the tables are incomplete, drop them.  Synchronises (one 16-byte copy).  The two functions below are the round-4 builder, kept as the cross-check
// ("col_strips_builder" = 1; test_strip_builders_agree).
int strip_segments(const Chunk *d_chunks, int32_t n_chunks, const int32_t *d_col_idx, int32_t K, int32_t S, Chunk *d_strips, void *d_scratch256,
                   SegmentSurvey *out);
int survey_segments(const Chunk *d_chunks, int32_t n_chunks, const int32_t *d_col_idx, void *d_scratch256, SegmentSurvey *out);   // synchronises (one 16-byte copy)
int build_col_strips(const Chunk *d_chunks, int32_t n_chunks, const int32_t *d_col_idx, int32_t K, int32_t S, Chunk *d_strips);   // asynchronous on the null stream
// the rule: how many strips for this plan (1 = none)
// front_pct (round 5): share of sampled nonzeros in the first quarter of the columns.  A vertex order that puts the hubs first (un-permuted R-MAT, crawl
// orders) puts the hot rows of B into the FIRST strip whatever its width; strips then cannot reach L2 size by the rule above (sub-segments would get too
// short) and yet four wide ones pay: the first launch gathers the hot rows with nothing cold evicting them, the others stream (rmat20-unpermuted kLen 32 /
// 128 / 256: 0.88 / 0.95 / 0.87 of the time without, profiles/r05_regret.md).
// local_pct (round 5): where the columns are local a launch's rows already share B rows through L2, and half as many strips do (10 MiB each: reddit-community
// N = 128 23 -> 11 strips 0.93, protein-community 13 -> 6 strips 0.95 of the time).
inline int32_t resolve_col_strips(int64_t K, int32_t tile_cols, int64_t seg_nnz, int32_t n_segments, int64_t nnz, int32_t front_pct = 25, int32_t local_pct = 0)
{
    // Measured on protein- and reddit-shaped graphs at N = 32 / 128 / 256, strip counts interleaved in one process (profiles/r04_col_strips.txt, sections 2
    // and 9): the best strip holds 4 - 6 MiB of B per column tile (one to one and a half XCD L2s; smaller strips pay more launches and shorter sub-segments
    // than they gain in hits), sub-segments down to ~20 nonzeros still pay with 16 gathers in flight, and a strip that cannot get below 16 MiB buys nothing.
    // the segments are not where the step's bytes are.  (A quarter, round 5; it was half: reddit-community at N >= 128 keeps its rows up to 512 nonzeros
    // in the rows kernel -- the medium rule above -- and the longer ones, 40 % of the nonzeros, still gain 0.70 - 0.79 of the step's time from strips.)
    if (n_segments <= 0 || seg_nnz * 4 < nnz) return 1;
    const double strip_target = (local_pct >= 50 ? 10.0 : 5.0) * 1048576.0;
    const double b_bytes = (double)K * 4.0 * (double)tile_cols;
    int64_t s = (int64_t)(b_bytes / strip_target + 0.5);
    const int64_t by_len = seg_nnz / n_segments / 20;                   // sub-segments of >= 20 nonzeros on average
    if (s > by_len) s = by_len;
    if (s > 32) s = 32;
    const bool front_loaded = front_pct >= 50 && by_len >= 4;           // (sub-segments of >= 80 nonzeros on average with four strips)
    if (front_loaded && (s < 2 || b_bytes / (double)s > 8.0 * 1048576.0)) return 4;
    if (s < 2) return 1;
    if (b_bytes / (double)s > 16.0 * 1048576.0) return 1;               // B far beyond the caches: strips that large buy nothing
    return (int32_t)s;
}

// ---- block path: items assembled on the device ---------------------------------------------------
// d_gp / d_groups: analyze_group_runs' output for the ng qualifying groups.  Pieces are ordered by (pass, list pieces before
// run pieces, first column, shareable first, longest first, group) with one stable radix sort; run pieces with the same first
// column form items of up to `share` pieces.  The result -- item array and launch table -- is what the host assembler
// (mi_spmm.hip build_block_items_host, kept behind "gpu_preprocess" = 0) produces, record for record.
struct BlockPlanOut {
    BlockItem *d_items = nullptr;                       // in: the caller's grow-only item buffer (or null); out: the buffer that holds the items
    size_t items_cap = 0;                               // in / out: its capacity in items (re-allocated only when too small)
    int32_t n_items = 0, n_pieces = 0, n_passes = 0, n_shared = 0;
    struct { int32_t off, n; } launch[kMaxPieces][2];   // [pass][0: list items, 1: run items]
};
int build_block_items_gpu(const GroupPieces *d_gp, const int32_t *d_groups, int32_t ng, int32_t share, int32_t run_unit, Scratch *sc,
                          BlockPlanOut *out);

}  // namespace mi
