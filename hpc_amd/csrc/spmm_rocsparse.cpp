// spmm_rocsparse.cpp -- vendor comparator: rocSPARSE SpMM behind include/mi_spmm_comparator.h.
// Mirrors SpMMCuSparse (PA4/workspace/src/spmm_cusparse.cu:3-34).  Not on the product path.
#include "../../include/mi_spmm_comparator.h"

#include <hip/hip_runtime_api.h>
#include <rocsparse/rocsparse.h>

#include <new>

struct mi_rocsparse_spmm {
    rocsparse_handle handle = nullptr;
    rocsparse_spmat_descr matA = nullptr;
    rocsparse_dnmat_descr matB = nullptr, matC = nullptr;
    const int32_t *d_ptr = nullptr, *d_idx = nullptr;
    const float *d_val = nullptr;
    int32_t num_v = 0, num_cols = 0, feat = 0;
    int64_t nnz = 0;
    rocsparse_spmm_alg alg = rocsparse_spmm_alg_default;
    float alpha = 1.0f, beta = 0.0f;  // spmm_cusparse.h:21-22
    void *buf = nullptr;
    size_t buf_bytes = 0;
    const float *bound_in = nullptr;
    float *bound_out = nullptr;
};

#define RS_TRY(x) do { rocsparse_status s_ = (x); if (s_ != rocsparse_status_success) return 1000 + (int)s_; } while (0)

static void drop_descr(mi_rocsparse_spmm *h)
{
    if (h->matA) rocsparse_destroy_spmat_descr(h->matA);
    if (h->matB) rocsparse_destroy_dnmat_descr(h->matB);
    if (h->matC) rocsparse_destroy_dnmat_descr(h->matC);
    h->matA = nullptr;
    h->matB = h->matC = nullptr;
    if (h->buf) (void)hipFree(h->buf);
    h->buf = nullptr;
    h->buf_bytes = 0;
}

extern "C" {

int mi_rocsparse_spmm_create(mi_rocsparse_spmm **out, const int32_t *d_row_ptr, const int32_t *d_col_idx,
                             const float *d_vals, int32_t num_v, int32_t num_cols, int64_t nnz, int32_t feat_in,
                             int32_t alg)
{
    if (!out || !d_row_ptr || num_v < 0 || num_cols < 0 || nnz < 0 || feat_in < 0) return -1;
    mi_rocsparse_spmm *h = new (std::nothrow) mi_rocsparse_spmm();
    if (!h) return -2;
    h->d_ptr = d_row_ptr;
    h->d_idx = d_col_idx;
    h->d_val = d_vals;
    h->num_v = num_v;
    h->num_cols = num_cols;
    h->nnz = nnz;
    h->feat = feat_in;
    h->alg = (rocsparse_spmm_alg)alg;
    rocsparse_status s = rocsparse_create_handle(&h->handle);
    if (s != rocsparse_status_success) {
        delete h;
        return 1000 + (int)s;
    }
    *out = h;
    return 0;
}

int mi_rocsparse_spmm_preprocess(mi_rocsparse_spmm *h, const float *d_vin, float *d_vout, void *stream)
{
    if (!h) return -3;
    drop_descr(h);
    RS_TRY(rocsparse_set_stream(h->handle, (hipStream_t)stream));
    RS_TRY(rocsparse_create_csr_descr(&h->matA, h->num_v, h->num_cols, h->nnz, (void *)h->d_ptr, (void *)h->d_idx,
                                      (void *)h->d_val, rocsparse_indextype_i32, rocsparse_indextype_i32,
                                      rocsparse_index_base_zero, rocsparse_datatype_f32_r));
    RS_TRY(rocsparse_create_dnmat_descr(&h->matB, h->num_cols, h->feat, h->feat, (void *)d_vin, rocsparse_datatype_f32_r,
                                        rocsparse_order_row));
    RS_TRY(rocsparse_create_dnmat_descr(&h->matC, h->num_v, h->feat, h->feat, (void *)d_vout, rocsparse_datatype_f32_r,
                                        rocsparse_order_row));
    size_t bytes = 0;
    RS_TRY(rocsparse_spmm(h->handle, rocsparse_operation_none, rocsparse_operation_none, &h->alpha, h->matA, h->matB,
                          &h->beta, h->matC, rocsparse_datatype_f32_r, h->alg, rocsparse_spmm_stage_buffer_size, &bytes,
                          nullptr));
    if (hipMalloc(&h->buf, bytes ? bytes : 16) != hipSuccess) return -2;
    h->buf_bytes = bytes;
    RS_TRY(rocsparse_spmm(h->handle, rocsparse_operation_none, rocsparse_operation_none, &h->alpha, h->matA, h->matB,
                          &h->beta, h->matC, rocsparse_datatype_f32_r, h->alg, rocsparse_spmm_stage_preprocess, &bytes,
                          h->buf));
    h->bound_in = d_vin;
    h->bound_out = d_vout;
    return 0;
}

int mi_rocsparse_spmm_run(mi_rocsparse_spmm *h, const float *d_vin, float *d_vout, void *stream)
{
    if (!h || !h->matA) return -3;
    if (d_vin != h->bound_in || d_vout != h->bound_out) {  // descriptors are bound to the buffers (spmm_cusparse.cu:11-15)
        RS_TRY(rocsparse_dnmat_set_values(h->matB, (void *)d_vin));
        RS_TRY(rocsparse_dnmat_set_values(h->matC, (void *)d_vout));
        h->bound_in = d_vin;
        h->bound_out = d_vout;
    }
    RS_TRY(rocsparse_set_stream(h->handle, (hipStream_t)stream));
    size_t bytes = h->buf_bytes;
    RS_TRY(rocsparse_spmm(h->handle, rocsparse_operation_none, rocsparse_operation_none, &h->alpha, h->matA, h->matB,
                          &h->beta, h->matC, rocsparse_datatype_f32_r, h->alg, rocsparse_spmm_stage_compute, &bytes,
                          h->buf));
    return 0;
}

int64_t mi_rocsparse_spmm_buffer_bytes(const mi_rocsparse_spmm *h) { return h ? (int64_t)h->buf_bytes : -1; }

int mi_rocsparse_spmm_destroy(mi_rocsparse_spmm *h)
{
    if (!h) return 0;
    drop_descr(h);
    if (h->handle) rocsparse_destroy_handle(h->handle);
    delete h;
    return 0;
}

}  // extern "C"
