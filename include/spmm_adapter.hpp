// spmm_adapter.hpp -- header-only C++ adapter: the reference's operator classes over the C ABI.
//
// The OPERATOR side of a harness written against the reference's headers compiles against this one (test/test_spmm.cu:33-36, 39-40, 43, 48-51,
// 57-60: tests/test_boundary_compiles.py); its own CUDA runtime calls (allocate<T>, cudaMemset, cudaFree, dbg: test_spmm.cu:16-19, 26, 37-38, 41) are
// the caller's to port:
//   struct CSR                      PA4/workspace/include/util.h:120-129
//   class SpMM (abstract)           PA4/workspace/include/spmm_base.h:8-46
//   class SpMMOpt : public SpMM     PA4/workspace/include/spmm_opt.h:12-29  (the drop-in)
//   class SpMMRef : public SpMM     PA4/workspace/include/spmm_ref.h:7-15   (exact-order configuration of the same library)
//   int valid(float*,float*,int), int valid(int*,int*,int)   PA4/workspace/include/valid.h
//   getCUDATime / getAverageTimeWithWarmUp                    PA4/workspace/include/util.h:131-151
//
// Error behaviour is the reference's (util.h:63-84): any non-zero status prints
// "Cuda failure: <code>" + file:line and exits(1).  The C ABI itself never aborts.
#ifndef MI_SPMM_ADAPTER_HPP
#define MI_SPMM_ADAPTER_HPP

#include <hip/hip_runtime_api.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <functional>

#include "mi_spmm.h"

#define MI_FATAL(code)                                                                         \
    do {                                                                                       \
        std::fprintf(stderr, "Cuda failure: %d (%s)\n%s:%d\nAborting...\n", (int)(code),       \
                     mi_spmm_strerror((int)(code)), __FILE__, __LINE__);                       \
        (void)hipDeviceReset();                                                                \
        std::exit(1);                                                                          \
    } while (0)
#define MI_CHECK(status)                     \
    do {                                     \
        int mi_s_ = (int)(status);           \
        if (mi_s_ != 0) MI_FATAL(mi_s_);     \
    } while (0)

// util.h:120-129 -- non-owning view of device arrays
struct CSR {
    CSR(int out_num_v, int out_num_e, int *outptr, int *outidx, float *outval)
        : num_v(out_num_v), num_e(out_num_e), ptr(outptr), idx(outidx), val(outval) {}
    int num_v = 0;
    int num_e = 0;
    int *ptr = nullptr;
    int *idx = nullptr;
    float *val = nullptr;
};

// spmm_base.h:8-46
class SpMM {
public:
    SpMM(int *dev_out_ptr, int *dev_out_idx, int out_num_v, int out_num_e, int out_feat_in)
        : d_ptr(dev_out_ptr), d_idx(dev_out_idx), feat_in(out_feat_in), num_v(out_num_v), num_e(out_num_e) {}
    SpMM(CSR *g, int out_feat_in) : feat_in(out_feat_in)
    {
        d_ptr = g->ptr;
        d_idx = g->idx;
        d_val = g->val;
        num_v = g->num_v;
        num_e = g->num_e;
    }
    virtual ~SpMM() {}
    virtual void set_feat(int given_feat) { this->feat_in = given_feat; }
    virtual void preprocess(float *vin, float *vout) = 0;
    virtual void run(float *vin, float *vout) = 0;

protected:
    int *d_ptr = nullptr;
    int *d_idx = nullptr;
    float *d_val = nullptr;
    int feat_in = 0;
    int num_v = 0;
    int num_e = 0;
};

// spmm_opt.h:12-29 -- the MI355X implementation
class SpMMOpt : public SpMM {
public:
    SpMMOpt(int *dev_out_ptr, int *dev_out_idx, int out_num_v, int out_num_e, int out_feat_in)
        : SpMM(dev_out_ptr, dev_out_idx, out_num_v, out_num_e, out_feat_in) {}
    SpMMOpt(CSR *g, int out_feat_in) : SpMM(g, out_feat_in) {}
    ~SpMMOpt() override
    {
        if (h_) (void)mi_spmm_destroy(h_);
    }
    void set_feat(int given_feat) override
    {
        SpMM::set_feat(given_feat);
        if (h_) MI_CHECK(mi_spmm_set_feat(h_, given_feat));
    }
    void set_stream(hipStream_t s) { stream_ = s; }  // reference: the null stream
    void set_option(const char *key, long long v)
    {
        ensure();
        MI_CHECK(mi_spmm_set_option(h_, key, v));
    }
    void preprocess(float *vin, float *vout) override
    {
        ensure();
        MI_CHECK(mi_spmm_preprocess(h_, vin, vout));
    }
    void run(float *vin, float *vout) override { MI_CHECK(mi_spmm_run(h_, vin, vout, (void *)stream_)); }

private:
    void ensure()
    {
        if (!h_) MI_CHECK(mi_spmm_create(&h_, d_ptr, d_idx, d_val, num_v, num_v /* square, spmm_cusparse.cu:6 */, num_e, feat_in));
    }
    mi_spmm_handle *h_ = nullptr;
    hipStream_t stream_ = nullptr;
};

// spmm_ref.h:7-15 / spmm_ref.cu:20-30 -- the course's reference operator.  The reference implements it with
// spmm_kernel_ref (one thread per row, spmm_ref.cu:3-17); here it is the product library's EXACT-ORDER
// configuration: every row, whatever its length, is one fma chain in stored order in the rows kernel (no row
// splitting, no segment kernel, no MFMA block path), which is bit-identical to spmm_kernel_ref -- pinned by
// tests/test_parity_gpu.py::test_reference_kernel_agrees_with_oracle_live and tests/test_fullsize_gpu.py against
// the reference kernel itself.  So the line test/test_spmm.cu:33 (`new SpMMRef(g, kLen)`) compiles as written (tests/test_boundary_compiles.py) and
// SpMMTest.validation compares SpMMOpt with SpMMRef on the device exactly as the reference does.
class SpMMRef : public SpMM {
public:
    SpMMRef(int *dev_out_ptr, int *dev_out_idx, int out_num_v, int out_num_e, int out_feat_in)
        : SpMM(dev_out_ptr, dev_out_idx, out_num_v, out_num_e, out_feat_in) {}
    SpMMRef(CSR *g, int out_feat_in) : SpMM(g, out_feat_in) {}
    ~SpMMRef() override
    {
        if (h_) (void)mi_spmm_destroy(h_);
    }
    void set_feat(int given_feat) override
    {
        SpMM::set_feat(given_feat);
        if (h_) MI_CHECK(mi_spmm_set_feat(h_, given_feat));
    }
    void preprocess(float *vin, float *vout) override
    {
        if (!h_) {
            MI_CHECK(mi_spmm_create(&h_, d_ptr, d_idx, d_val, num_v, num_v, num_e, feat_in));
            MI_CHECK(mi_spmm_set_option(h_, "long_row_threshold", 1LL << 30));    // never split a row
            MI_CHECK(mi_spmm_set_option(h_, "medium_row_threshold", 1LL << 30));  // every row stays in the rows kernel
            MI_CHECK(mi_spmm_set_option(h_, "block_path", 0));
        }
        MI_CHECK(mi_spmm_preprocess(h_, vin, vout));
    }
    void run(float *vin, float *vout) override { MI_CHECK(mi_spmm_run(h_, vin, vout, nullptr)); }

private:
    mi_spmm_handle *h_ = nullptr;
};

#ifdef MI_SPMM_WITH_COMPARATOR
#include "mi_spmm_comparator.h"
// spmm_cusparse.h:6-24 -- the vendor comparator (rocSPARSE here); link libmi_spmm_rocsparse.so
class SpMMRocSparse : public SpMM {
public:
    SpMMRocSparse(int *p, int *i, int nv, int ne, int f) : SpMM(p, i, nv, ne, f) {}
    SpMMRocSparse(CSR *g, int out_feat_in) : SpMM(g, out_feat_in) {}
    ~SpMMRocSparse() override
    {
        if (h_) (void)mi_rocsparse_spmm_destroy(h_);
    }
    void preprocess(float *vin, float *vout) override
    {
        if (!h_) MI_CHECK(mi_rocsparse_spmm_create(&h_, d_ptr, d_idx, d_val, num_v, num_v, num_e, feat_in, 0));
        MI_CHECK(mi_rocsparse_spmm_preprocess(h_, vin, vout, nullptr));
    }
    void run(float *vin, float *vout) override { MI_CHECK(mi_rocsparse_spmm_run(h_, vin, vout, nullptr)); }

private:
    mi_rocsparse_spmm *h_ = nullptr;
};
using SpMMCuSparse = SpMMRocSparse;  // the reference harness's name for it (test_spmm.cu:48)
#endif

// valid.h / valid.cu:22-51
inline int valid(float *y, float *y2, int num)
{
    int64_t bad = -1;
    MI_CHECK(mi_spmm_valid_float(y, y2, num, &bad, nullptr));
    return (int)bad;
}
inline int valid(int *y, int *y2, int num)
{
    int64_t bad = -1;
    MI_CHECK(mi_spmm_valid_int(y, y2, num, &bad, nullptr));
    return (int)bad;
}

// util.h:131-151
inline double getCUDATime(const std::function<void()> &f)
{
    MI_CHECK(hipDeviceSynchronize());
    auto t0 = std::chrono::system_clock::now();
    f();
    MI_CHECK(hipDeviceSynchronize());
    auto t1 = std::chrono::system_clock::now();
    return std::chrono::duration<double>(t1 - t0).count();
}
inline double getAverageTimeWithWarmUp(const std::function<void()> &f)
{
    const int n_warmup = 10, n_run = 20;
    for (int i = 0; i < n_warmup; ++i) f();
    double total_time = 0;
    for (int i = 0; i < n_run; ++i) total_time += getCUDATime(f);
    return total_time / n_run;
}

#endif  // MI_SPMM_ADAPTER_HPP
