// plan.hpp -- what preprocess produces for run(): the segment table (medium rows + pieces of split
// rows, longest first), the split-row table, the compacted list of block-path groups.  Built either
// on the GPU (preprocess_gpu.hip, default) or by the reference-style host loop (mi_spmm.hip,
// "gpu_preprocess" = 0, kept as the cross-check).  Internal to libmi_spmm.so.
#pragma once
#include <stddef.h>
#include <stdint.h>

#include "plan_types.hpp"

namespace mi {

struct PlanOut {
    Chunk *d_chunks = nullptr;      // [n_chunks], sorted by length descending (stable in row order)
    LongRow *d_long = nullptr;      // [n_long]
    int32_t *d_blk_groups = nullptr;  // [n_blk_groups] (only when d_blk_flag was given)
    int32_t n_chunks = 0, n_long = 0, n_slots = 0, n_medium = 0, n_blk_groups = 0;
    int32_t max_len = 0;
    int32_t mthr = 0;               // resolved medium threshold (the rows kernel skips rows above it)
    int32_t local_pct = 0;          // sampled nonzeros within a window of their row's own position, percent (column-tile rule)
};

// Returns 0, a negative MI_SPMM_E* code (malformed CSR, out of memory) or a positive hipError_t.
// d_blk_flag: per 16-row group, 1 = block path owns it (nullable).  col_bad: device flag written
// by csr_check_cols earlier on the same (null) stream; read back with the same single copy.
// mthr: medium threshold, 0 = auto (resolved on the device from the longest row, returned in PlanOut::mthr).
// A grow-only device arena owned by the handle: preprocess temporaries are carved out of it.
struct Scratch {
    void *p = nullptr;
    size_t cap = 0;
};
int scratch_reserve(Scratch *s, size_t bytes);   // 0 or MI_SPMM_ENOMEM; contents are lost when it grows
void scratch_release(Scratch *s);

int build_plan_gpu(const int32_t *d_row_ptr, const int32_t *d_col_idx, int32_t M, int32_t K, int64_t nnz, const uint8_t *d_blk_flag,
                   const unsigned int *d_col_bad, int32_t mthr, int32_t thr, int32_t clen, Scratch *sa, Scratch *sb, PlanOut *out);

}  // namespace mi
