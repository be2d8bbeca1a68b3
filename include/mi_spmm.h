/*
 * mi_spmm.h -- C ABI of the MI355X-native CSR SpMM  (C = A_csr * B_dense).
 *
 * This is the drop-in boundary for the liblaf/hpc PA4 operator.  The
 * reference has no FFI of its own: its boundary is the abstract C++ class
 * `SpMM` (PA4/workspace/include/spmm_base.h:8-46) with
 *      SpMM(CSR *g, int feat_in)
 *      virtual void preprocess(float *vin, float *vout)
 *      virtual void run(float *vin, float *vout)
 * implemented by SpMMRef / SpMMOpt / SpMMCuSparse.  Each entry point below
 * names the reference member it replaces; include/spmm_adapter.hpp wraps them
 * back into a class with exactly that shape, INTEGRATION.md shows the binding.
 *
 * Conventions (all from the reference):
 *   - every buffer is a DEVICE pointer owned by the caller
 *     (test/test_spmm.cu:16-20, test/main.cpp:11-14); the handle never frees
 *     them and owns only what preprocess allocates;
 *   - CSR: int32 row_ptr[num_v+1], int32 col_idx[nnz], fp32 vals[nnz],
 *     0-based (src/spmm_cusparse.cu:6-9), row_ptr[num_v] == nnz
 *     (src/data.cu:40-45), column order inside a row unspecified;
 *   - dense B (vin) and C (vout) are row-major fp32 with leading dimension
 *     feat_in (src/spmm_cusparse.cu:11-15); the reference is square (K = M);
 *   - run() leaves vout = A*vin (overwrite, like spmm_ref.cu:15 and cuSPARSE
 *     beta = 0, include/spmm_cusparse.h:22), is idempotent, asynchronous on
 *     the given stream and performs no host synchronisation;
 *   - arithmetic per output element: fp32 fused multiply-add chain over the
 *     row's nonzeros in stored order starting from +0.0f (spmm_ref.cu:10-14
 *     under the reference's nvcc --use_fast_math build).  With default options
 *     EVERY row is that chain, bit for bit, whatever its length (short rows,
 *     single segments, hub rows through the hub kernel, block groups through
 *     the f32 MFMA).  Only the opt-in "split_long_rows" = 1 sums rows longer
 *     than the threshold piece by piece (tolerance: DESIGN.md).  The reference's BUILD
 *     additionally flushes fp32 subnormals (nvcc --use_fast_math => -ftz=true): opt-in
 *     "flush_denormals" = 1 reproduces that too, bit for bit.
 *
 * Errors: the reference aborts (include/util.h:63-84).  The C ABI never
 * aborts: every function returns 0 on success or a negative MI_SPMM_E* /
 * positive hipError_t code; the C++ adapter restores abort-on-error.
 *
 * Threading: a handle is not thread-safe; use one handle per host thread and
 * device (the reference is single-threaded with globals, src/util.cu:3-12).
 */
#ifndef MI_SPMM_H
#define MI_SPMM_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MI_SPMM_ABI_VERSION 1

/* error codes (negative = ours, positive = hipError_t passed through) */
#define MI_SPMM_OK            0
#define MI_SPMM_EINVAL       -1  /* bad argument (NULL pointer, negative size) */
#define MI_SPMM_ENOMEM       -2  /* host or device allocation failed */
#define MI_SPMM_ESTATE       -3  /* run() before preprocess(), bad handle */
#define MI_SPMM_ECSR         -4  /* row_ptr not monotone / row_ptr[M] != nnz / column out of range */
#define MI_SPMM_EUNSUPPORTED -5  /* option or shape not supported */
#define MI_SPMM_ENODEVICE    -6  /* no gfx950 device / HIP runtime unusable */

typedef struct mi_spmm_handle mi_spmm_handle;

/* Replaces SpMM::SpMM(CSR *g, int feat_in) and struct CSR
 * (spmm_base.h:14-21, util.h:120-129).  Stores the five scalars and three
 * device pointers; touches no device memory.  num_cols = K (rows of B); the
 * reference always has K = num_v. */
int mi_spmm_create(mi_spmm_handle **out,
                   const int32_t *d_row_ptr, const int32_t *d_col_idx,
                   const float *d_vals,
                   int32_t num_v, int32_t num_cols, int64_t nnz,
                   int32_t feat_in);

/* Replaces SpMM::set_feat (spmm_base.h:26-29).  Invalidates preprocess. */
int mi_spmm_set_feat(mi_spmm_handle *h, int32_t feat_in);

/* Replaces SpMMOpt::preprocess(vin, vout) (src/spmm_opt.cu:37-69): validates
 * the CSR, classifies the rows and builds the segment / split-row / block-group
 * tables -- on the device, one small copy comes back -- and allocates the
 * handle-owned workspace.  Synchronises the device (the reference's does:
 * cudaMemcpy D2H, :39).  Unlike the reference it does NOT need vout zeroed and
 * does not touch it. */
int mi_spmm_preprocess(mi_spmm_handle *h, const float *d_vin, float *d_vout);

/* Replaces SpMMOpt::run(vin, vout) (src/spmm_opt.cu:71-75).  stream is a
 * hipStream_t (NULL = the null stream the reference uses). */
int mi_spmm_run(mi_spmm_handle *h, const float *d_vin, float *d_vout,
                void *stream);

/* Same with explicit row pitches (in floats) for B and C: lets a column shard
 * read its slice of a wider row-major B and write its slice of a wider C
 * (multi-GPU column sharding; no reference counterpart). */
int mi_spmm_run_ld(mi_spmm_handle *h, const float *d_vin, int64_t ldb,
                   float *d_vout, int64_t ldc, void *stream);

/* Same for the row range [row_begin, row_end) only (d_vout is still the base of
 * the full-height C): lets the multi-GPU driver compute C in row panels and
 * overlap each panel's all-gather with the next panel's kernel.  Exactly the
 * rows of the range are written, whatever kernel class they belong to. */
int mi_spmm_run_rows(mi_spmm_handle *h, const float *d_vin, int64_t ldb,
                     float *d_vout, int64_t ldc, int32_t row_begin,
                     int32_t row_end, void *stream);

/* Same, and every finished row segment of C is ALSO stored at the same element offset of n_extra (<= 7) further buffers
 * of the same layout (same pitch ldc; d_extra[i] points where d_vout points in its buffer).  This is the multi-GPU
 * "peer_store" exchange (include/mi_spmm_dist.h): the extra buffers are the peers' C, mapped through HIP IPC, and the
 * kernels' epilogues write every result once locally and once per peer -- no staging, no copy, no re-layout.
 * Chains a later block pass continues (carried tiles) stay local.  No reference counterpart. */
int mi_spmm_run_rows_multi(mi_spmm_handle *h, const float *d_vin, int64_t ldb,
                           float *d_vout, int64_t ldc, int32_t row_begin,
                           int32_t row_end, int32_t n_extra, float *const *d_extra,
                           void *stream);

/* Replaces SpMMOpt::~SpMMOpt (include/spmm_opt.h:18-20). */
int mi_spmm_destroy(mi_spmm_handle *h);

/* Message for a code returned by any function here. */
const char *mi_spmm_strerror(int code);

/* Tuning / introspection.  Keys (all int64):
 *   "medium_row_threshold" rows longer than this run as ONE exact segment in the segment kernel (0 = auto:
 *                         64, or 32 when the longest row exceeds 8x the mean degree -- 256 where the columns are local
 *                         (>= 50 % of a row sample near the row's own position: the rows kernel keeps neighbours together,
 *                         the length-sorted segment table scatters them; plan.hpp resolve_medium_threshold); get returns
 *                         the resolved value after preprocess).  Scheduling only: results do not depend on it
 *   "long_row_threshold"  rows with more nonzeros are HUBS: they leave the segment kernel for the hub kernel (stored order:
 *                         loader waves + one chain wave per 32-column slice, 3.2 ns per nonzero of a row instead of 47).
 *                         0 = auto: a power of two in 256 .. 8192 from the row-length histogram, priced at the L2's rates
 *                         when B is L2-resident (hpc_amd/csrc/plan.hpp resolve_hub_threshold); get returns the resolved
 *                         value after preprocess.  Scheduling only in
 *                         the default mode: results do not depend on it
 *   "split_long_rows"     0 (default): hubs keep their stored order.  1: hubs are cut into pieces of "long_row_chunk"
 *                         nonzeros whose partial sums are added left to right (deterministic, not the reference's order;
 *                         auto threshold then clamp(nnz/8192, 256, 2048)).  Faster only when one row holds a few per cent
 *                         of a small matrix (an exact chain cannot run faster than ~5 cycles per nonzero of that row)
 *   "long_row_chunk"      piece length in nonzeros (split mode)
 *   "hub_slice"           columns per hub workgroup: 16, 32, 64; 0 = auto (32; 16 when N <= 16 or when the longest
 *                         row's chain alone is more than half of the step)
 *   "hub_overlap"         1 (default): the hub kernel runs on a handle-owned high-priority side stream forked from and joined
 *                         into the caller's stream inside every run call (events only), when the step is long enough to hide its
 *                         longest row behind the rows kernel (the fork costs ~20 us); 0: never; 2: always
 *   "segment_overlap"     2 (default, auto): the segment kernel gets a second side stream (under the "hub_overlap" rule) only where an interleaved
 *                         A/B measured a gain: N <= 64 and no hub stream in use (citation- / wikikg2-shaped kLen 32: -8 %); elsewhere it stays on
 *                         the caller's stream, in front of the rows kernel (a second stream there is neutral to +16 %).  0: never.  1: always.
 *                         Every side stream is TESTED to run beside the null stream before it is kept (mi_spmm_stream_create_concurrent):
 *                         read-only "side_stream_overlaps" says whether the hub stream passed
 *   "side_priority"       bit 0 / bit 1: the hub / segment side stream is a high-priority stream (default 3)
 *   "flush_denormals"     0 (default): IEEE fp32 arithmetic, subnormals kept -- the canonical definition (spmm_ref.cu:10-14 with fma contraction).
 *                         1: the arithmetic of the reference's actual BUILD: nvcc --use_fast_math (CMakeLists.txt:46) implies -ftz=true, so
 *                         its multiply-adds are fma.rn.ftz.f32 -- subnormal inputs count as sign-preserving zeros, subnormal results are
 *                         flushed to sign-preserving zeros.  Implemented with the wave's own mode register (MODE.FP_DENORM, set at kernel
 *                         entry); bit-identical to spmm_kernel_ref compiled with the matching switch (hipcc -fgpu-flush-denormals-to-zero)
 *                         on data full of subnormals.  The f32 MFMA block path is not used with it.  On data that holds no subnormals and
 *                         produces none (the reference's N(0, 0.1) inputs) the two settings give the same bits.
 *   "fused_step"          2 (default, auto) / 0 / 1: hub rows, segments and short rows as the three ROLES OF ONE LAUNCH (spmm_small_step: workgroups
 *                         take their role from blockIdx, hub slices first so that the step's longest chain starts first) instead of two or three
 *                         launches plus a side-stream fork and join.  Eligible: one column tile (N <= 256), no column strips, no split rows, no
 *                         block groups, default cache policy.  auto: only steps with at least two of the three roles present whose bytes take under 0.1 - 0.16 ms at 6 TB/s (there the launch
 *                         boundaries are a third of the step; every role runs at the kernel's footprint, 128 VGPRs = 4 waves per SIMD).  1: whenever
 *                         eligible.  Same device functions, same arithmetic: same bits.  Read-only "fused_step_in_force": the last run used it
 *   "fused_order"         0 (default, auto) / 1 / 2: which of the small-step kernel's first two roles leads its grid (workgroups start in blockIdx order):
 *                         1 = hub workgroups, 2 = segment workgroups.  auto: the role whose longest chain lasts longest -- a hub row at 3.2 ns per
 *                         nonzero + ~2 us of fill against the longest segment (as long as the hub threshold allows) at 47 ns per nonzero, 30 out of
 *                         an L2-resident B.  Scheduling only: same bits.  Read-only "fused_order_in_force": 0 (last run not fused) / 1 / 2
 *   "segment_order"       0 (default, auto) / 1: the segment table is sorted longest first (the lane groups of a wave carry similar lengths) / 2: it stays
 *                         in row order (neighbouring rows -- which gather the same B rows where the columns are local -- stay together).  auto = 1: row
 *                         order measured mixed on structured graphs (profiles/r05_regret.md); "autotune" tries it.  Scheduling only
 *   "rows_unroll"         0 (default, auto: 8; 16 for banded columns with a mean degree >= 256 at one whole-wave column tile) / 8 / 16: B-row gathers a lane group of the rows kernel keeps in flight per batch of its row's chain (16 only
 *                         on the plain path: 4-float lanes, 32-bit offsets, 256-thread workgroups, default cache policy; elsewhere 8 stays in force).  16 halves
 *                         the round trips of a row and costs occupancy: -13 % ... +23 % by graph (profiles/r05_rows_unroll_ab.txt); the one class a plan statistic
 *                         separates is the banded long-row one (profiles/r05_banded_unroll_ab.jsonl); "autotune" tries the other depth.  Same chain per row:
 *                         same bits.  Read-only "rows_unroll_in_force": 8 / 16 of the last rows launch
 *   "autotune"            0 (default) / 1: the rules behind the options above are guesses from a row sample and a histogram, and a wrong guess is silent
 *                         (same bits, slower).  With 1, preprocess MEASURES instead: the step is timed on the vin / vout it is given -- vout is written,
 *                         as the reference's preprocess does (spmm_opt.cu:67) -- under the auto plan and under a dozen forced settings of the options the
 *                         caller left at auto ("long_row_threshold" one notch down / up, "medium_row_threshold", "col_strips", "tile_cols", "fused_step", "segment_order", "rows_unroll"; an explicit value of the caller's is
 *                         never touched), one option at a time, and the fastest is kept (it has to win by 3 %).  Costs a dozen plans and ~50 steps of
 *                         preprocess time; scheduling only: same bits.  Afterwards the tuned options read back their chosen values; read-only
 *                         "autotune_evals", "autotune_auto_us", "autotune_best_us", "autotune_mask" (bit 0 tile, 1 strips, 2 medium, 3 fused, 4 segment order, 5 rows unroll, 6 hub threshold: what it changed)
 *   ("use_graph", round 4 -- the handle capturing its own launch set into a HIP graph and replaying it -- was removed in round 5: it lost on every graph,
 *    launched on the caller's stream or on a tested stream of its own, profiles/r05_use_graph_experiment.md; the key answers MI_SPMM_EUNSUPPORTED.  run()
 *    allocates nothing and synchronises nothing, so a caller can still capture it into a graph of its own: test_run_is_graph_capturable_and_stream_ordered.)
 *   "rows_per_block"      rows handled by one workgroup (0 = auto: one row per lane group)
 *   "block_threads"       workgroup size of the pipelined rows kernel (64, 128, 256)
 *   "segment_unroll"      B-row gathers in flight per lane group in the segment kernel: 8, 16, 32; 0 (default) = auto: 32, and 16 for the
 *                         launches of column strips (below)
 *   "split_cols"          1 (default): up to 64 columns past the last full 256-column tile get their own launches
 *   "tile_cols"           widest column tile of the rows / segment kernels: 256 (one row per wavefront), 128, 64, 32;
 *                         0 = auto.  Tiles are swept one after the other, so this sets the B working set of a sweep
 *                         (K x tile_cols x 4 bytes).  Scheduling only: results do not depend on it
 *   "col_strips"          column strips of the exact segments: every segment is cut at S - 1 column boundaries and the segment kernel runs
 *                         strip after strip (S launches in stream order), each continuing the rows' fma chains through C, so that a launch
 *                         gathers out of K / S rows of B -- an L2-sized piece when B is a few tens of MiB (graphs of long rows over few
 *                         columns: 1.2 - 1.6 x).  0 = auto (hpc_amd/csrc/plan.hpp resolve_col_strips: about 5 MiB of B per strip and column
 *                         tile, sub-segments of >= 20 nonzeros, at most 32 strips), 1 = off, 2 .. 64 = that many.  Needs ascending columns in
 *                         every segment (checked by preprocess; otherwise no strips).  Scheduling only: results do not depend on it.
 *                         auto also: none where >= 90 % of a row sample lies near the row's own position (banded / mesh); four wide strips where
 *                         the columns are front-loaded (hubs-first vertex orders: >= 50 % of the sample in the first quarter of the columns)
 *   "col_strips_builder"  0 (default): the tables are written in the same single pass over the segments' columns that checks their order
 *                         (strip_segments); 1: round 4's survey + S binary searches per segment (kept as the cross-check: same tables)
 *   "xcd_remap"           0/1: contiguous row ranges per XCD; -1 = auto
 *   "gpu_preprocess"      1 (default): segment table built on the device; 0: reference-style host loop
 *   "kernel"              2: pipelined items (spmm_rows_v2).  1 named the first-generation kernel: MI_SPMM_EUNSUPPORTED
 *   "nt_store"            0/1: non-temporal stores of C
 *   "nt_stream"           0/1: non-temporal loads of col_idx/vals
 *   "block_path"          0/1: 16-row groups sharing one column list go through the MFMA path
 *   "block_min_len"       shortest shared column list the block path takes
 *   "block_max_pieces"    most runs of consecutive columns a group's list is cut into (1..4; 1 = never cut).  Run p of a
 *                         group is processed in pass p, its fma chains continued through C: same bits, fewer B bytes
 *   "block_run_min"       shortest run worth a piece (and a pass) of its own; a list with a shorter run stays whole
 *   "block_share"         most pieces that share one fetch of their B rows (1 or 2)
 * set before preprocess; get any time.  Read-only keys after preprocess:
 *   "n_long_rows" (rows above the threshold), "n_hub_rows" (those of them the hub kernel takes: all, or 0 in split mode),
 *   "n_medium_rows", "n_chunks", "n_partial_slots", "workspace_bytes" (plan tables + partial sums + preprocess arenas),
 *   "n_launches", "lanes_per_row", "preprocess_us", "feat", "num_v", "num_cols", "max_row_nnz",
 *   "n_block_groups", "n_block_pieces", "n_block_items", "n_block_shared_items", "n_block_passes",
 *   "column_locality_pct" (share of sampled nonzeros near their row's own position; behind the "tile_cols" auto rule),
 *   "column_front_pct" (share of the sample in the first quarter of the columns; uniform: 25),
 *   "n_col_strips" (strips in force, 1 = none), "segments_unsorted" (segments whose columns do not ascend; -1 = not looked at),
 *   "segment_nnz" (nonzeros in the whole segments), "col_strips_table_hash" (FNV-1a of the strip tables, copied back: tests) */
int mi_spmm_set_option(mi_spmm_handle *h, const char *key, int64_t value);
int mi_spmm_get_option(const mi_spmm_handle *h, const char *key, int64_t *value);

/* Replaces valid(float*, float*, int) / validate_float
 * (src/valid.cu:3-11,36-51): counts elements with
 * |(y[i]-y2[i])/y[i]| > 1e-2 on the device.  n is 64-bit here (the
 * reference's int overflows at M*N >= 2^31). */
int mi_spmm_valid_float(const float *d_y, const float *d_y2, int64_t n,
                        int64_t *bad_out, void *stream);
/* Replaces valid(int*, int*, int) / validate_int (src/valid.cu:13-34). */
int mi_spmm_valid_int(const int32_t *d_y, const int32_t *d_y2, int64_t n,
                      int64_t *bad_out, void *stream);

/* Exact comparison helpers for parity tests: number of fp32 elements whose
 * BIT PATTERNS differ, and max |a-b| (both computed on the device). */
int mi_spmm_count_bitdiff(const float *d_a, const float *d_b, int64_t n,
                          int64_t *ndiff_out, float *maxabs_out, void *stream);

/* Replaces the fill half of allocate<T>() (include/data.h:24-37: cudaMalloc2 +
 * curandGenerateNormal(kCuRand, p, n, 0.f, 0.1), generator seeded 123 in
 * test/main.cpp:19-20): fp32 N(mean, stddev) written on the device.  Counter-based
 * Philox4x32-10 + Box-Muller: element i depends only on (seed, subsequence, i), so any
 * slice can be regenerated anywhere.  Same distribution as the reference; cuRAND's
 * XORWOW bit stream itself is not reproducible without cuRAND. */
int mi_spmm_fill_normal(float *d_out, int64_t n, uint64_t seed, uint64_t subsequence,
                        float mean, float stddev, void *stream);
/* The raw Philox4x32-10 words behind mi_spmm_fill_normal (known-answer testable). */
int mi_spmm_fill_philox_u32(uint32_t *d_out, int64_t n, uint64_t seed, uint64_t subsequence,
                            void *stream);

/* Column-shard plumbing for the multi-GPU all-gather.
 * RCCL all-gather delivers rank-major blocks  staging[G][rows][n_loc];
 * this writes them into row-major C[rows][ldc] at column g*n_loc. */
int mi_spmm_unpack_gathered(const float *d_staging, float *d_C, int64_t rows,
                            int32_t n_ranks, int32_t n_loc, int64_t ldc,
                            void *stream);

/* A stream whose kernels run BESIDE the null stream's, for callers that overlap work with run() (the multi-GPU step's exchange and
 * re-layout streams; the handle's own hub stream is made this way).  The runtime maps a process's streams onto a few hardware queues
 * per priority in creation order, and a queue that shares its command-processor pipe with the null stream's is served before or after
 * it, not alongside: whether a new stream overlaps is a matter of how many streams the process made before.  So candidates are tested
 * -- a 40 us spin kernel on the candidate and one on the null stream: ~65 us together when they run side by side, 90-120 when not --
 * and the first that passes is returned (at most four are tried; the others are destroyed).  *overlaps_out (may be NULL) says whether
 * the returned stream passed.  Synchronises (events): call at set-up time, not inside a step.  The caller destroys the stream
 * (hipStreamDestroy).  high_priority != 0: the device's highest stream priority.  No reference counterpart (single stream, util.h:133-136). */
int mi_spmm_stream_create_concurrent(void **stream_out, int high_priority, int *overlaps_out);

/* Library/ABI identification. */
int mi_spmm_abi_version(void);
const char *mi_spmm_build_info(void);

#ifdef __cplusplus
}
#endif
#endif /* MI_SPMM_H */
