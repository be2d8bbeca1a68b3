/*
 * mi_spmm_comparator.h -- C ABI of the vendor comparator (rocSPARSE SpMM).
 *
 * Replaces the reference's `SpMMCuSparse` (PA4/workspace/include/spmm_cusparse.h:6-24,
 * src/spmm_cusparse.cu:3-34): CSR (32-bit indices, base 0, fp32) x dense row-major
 * (ld = N) -> dense row-major, alpha = 1, beta = 0, default algorithm, external
 * buffer allocated in preprocess.  It is the "ref time" column of the reference's
 * report (PA4/report.md:41-73, first `time =` line of each log block) and an
 * independent GPU-side value check; it is NOT on the product path and lives in its
 * own library (hpc_amd/libmi_spmm_rocsparse.so) so libmi_spmm.so has no vendor
 * dependency.
 *
 * Unlike the reference (which ignores its members and reads the globals
 * kNumV/kNumE/kLen, spmm_cusparse.cu:6,11,14, and ignores every cusparse status)
 * the sizes are the handle's own and statuses are returned.
 */
#ifndef MI_SPMM_COMPARATOR_H
#define MI_SPMM_COMPARATOR_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct mi_rocsparse_spmm mi_rocsparse_spmm;

/* alg: 0 default (as the reference), 1 csr, 4 csr_row_split, 5 csr_merge (rocsparse_spmm_alg) */
int mi_rocsparse_spmm_create(mi_rocsparse_spmm **out, const int32_t *d_row_ptr, const int32_t *d_col_idx,
                             const float *d_vals, int32_t num_v, int32_t num_cols, int64_t nnz,
                             int32_t feat_in, int32_t alg);
/* SpMMCuSparse::preprocess (spmm_cusparse.cu:3-25): descriptors bound to vin/vout, buffer size, buffer */
int mi_rocsparse_spmm_preprocess(mi_rocsparse_spmm *h, const float *d_vin, float *d_vout, void *stream);
/* SpMMCuSparse::run (spmm_cusparse.cu:27-34) */
int mi_rocsparse_spmm_run(mi_rocsparse_spmm *h, const float *d_vin, float *d_vout, void *stream);
int mi_rocsparse_spmm_destroy(mi_rocsparse_spmm *h);
int64_t mi_rocsparse_spmm_buffer_bytes(const mi_rocsparse_spmm *h);

#ifdef __cplusplus
}
#endif
#endif
