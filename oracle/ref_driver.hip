// oracle/ref_driver.hip -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
//
// Launcher around the REFERENCE's own kernels, compiled for gfx950 by hipcc
// straight from the read-only reference tree (see oracle/Makefile, target
// _ref).  No reference text lives in this repository: the Makefile cuts the
// kernel definitions out of /root/reference at build time into a temporary
// file that is deleted after the compile; only the resulting shared object
// (oracle/_ref/libspmm_ref_gfx950.so, git-ignored) remains.
//
// What is and is not built from the reference:
//   built:      spmm_kernel_ref      PA4/workspace/src/spmm_ref.cu:3-17
//               struct Task          PA4/workspace/include/spmm_opt.h:6-10
//               kBatchSize, kTasksPerBlock, SpmmOptKernel
//                                    PA4/workspace/src/spmm_opt.cu:6-35
//               validate_float/int   PA4/workspace/src/valid.cu:3-20
//   not built:  everything that needs the CUDA runtime, cuRAND, cuSPARSE or
//               googletest (the class methods, the harness).  The kernels
//               themselves are plain CUDA-dialect device code that HIP accepts
//               as is (__global__, blockIdx, atomicAdd are HIP-native): no
//               stand-in header, library or macro is supplied for them.
//
// The host side below restates the launch geometry only:
//   SpMMRef::preprocess/run   spmm_ref.cu:20-30   block 128, grid ceil(M/128)
//   SpMMOpt::preprocess/run   spmm_opt.cu:37-75   tasks of <=256 nnz, block (1,N)
//   valid()                   valid.cu:22-51      grid ceil(n/128), block 128
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <vector>

#include REF_EXTRACT_INC  // supplied by oracle/Makefile; temporary, deleted after the build

#define REF_CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return (int)e_; } while (0)

extern "C" {

// SpMMRef::preprocess + run  (spmm_ref.cu:20-30)
int ref_spmm_ref_run(int *ptr, int *idx, float *val, float *vin, float *vout,
                     int num_v, int feat_in, void *stream)
{
    const int BLOCK_SIZE = 128;
    dim3 grid((num_v + BLOCK_SIZE - 1) / BLOCK_SIZE), block(BLOCK_SIZE);
    if (num_v <= 0) return 0;
    hipLaunchKernelGGL(spmm_kernel_ref, grid, block, 0, (hipStream_t)stream,
                       ptr, idx, val, vin, vout, num_v, feat_in);
    return (int)hipGetLastError();
}

// SpMMOpt::preprocess (spmm_opt.cu:37-69) minus the random_shuffle (order of
// tasks only changes which block does what, and -- for rows with more than one
// task -- the atomicAdd arrival order, which the GPU scheduler randomises
// anyway).  Returns a device task array the caller frees with ref_free.
int ref_spmm_opt_preprocess(const int *d_ptr, int num_v, void **d_tasks_out,
                            int *num_tasks_out)
{
    std::vector<int> h_ptr((size_t)num_v + 1);
    REF_CHECK(hipMemcpy(h_ptr.data(), d_ptr, sizeof(int) * ((size_t)num_v + 1),
                        hipMemcpyDeviceToHost));
    std::vector<Task> tasks;
    for (int row = 0; row < num_v; ++row) {
        const int begin = h_ptr[row], end = h_ptr[row + 1];
        for (int b = begin; b < end; b += kBatchSize) {
            Task t;
            t.row = row;
            t.ptr_begin = b;
            t.ptr_end = std::min(b + kBatchSize, end);
            tasks.push_back(t);
        }
    }
    void *d = nullptr;
    if (!tasks.empty()) {
        REF_CHECK(hipMalloc(&d, tasks.size() * sizeof(Task)));
        REF_CHECK(hipMemcpy(d, tasks.data(), tasks.size() * sizeof(Task),
                            hipMemcpyHostToDevice));
    }
    *d_tasks_out = d;
    *num_tasks_out = (int)tasks.size();
    return 0;
}

// SpMMOpt::run (spmm_opt.cu:71-75): ACCUMULATES into vout (atomicAdd, :34);
// the caller zeroes vout first, as preprocess does (:67-68).
int ref_spmm_opt_run(const void *d_tasks, int num_tasks, const int *idx,
                     const float *val, const float *vin, float *vout,
                     int feat_in, void *stream)
{
    if (num_tasks <= 0) return 0;
    dim3 block(kTasksPerBlock, feat_in);
    dim3 grid((num_tasks + block.x - 1) / block.x, feat_in / block.y);
    hipLaunchKernelGGL(SpmmOptKernel, grid, block, 0, (hipStream_t)stream,
                       (const Task *)d_tasks, idx, val, vin, vout, num_tasks,
                       feat_in);
    return (int)hipGetLastError();
}

int ref_free(void *d) { return d ? (int)hipFree(d) : 0; }

// valid(float*, float*, int)  (valid.cu:36-51)
int ref_valid_float(float *y, float *y2, int num, int *bad_out)
{
    int *diffnum = nullptr;
    REF_CHECK(hipMalloc((void **)&diffnum, 512));
    REF_CHECK(hipMemset(diffnum, 0, sizeof(int)));
    REF_CHECK(hipDeviceSynchronize());
    hipLaunchKernelGGL(validate_float, dim3((num + 127) / 128), dim3(128), 0, 0,
                       y, y2, num, diffnum);
    REF_CHECK(hipDeviceSynchronize());
    int ans = -1;
    REF_CHECK(hipMemcpy(&ans, diffnum, sizeof(int), hipMemcpyDeviceToHost));
    hipFree(diffnum);
    *bad_out = ans;
    return 0;
}

// valid(int*, int*, int)  (valid.cu:22-34)
int ref_valid_int(int *y, int *y2, int num, int *bad_out)
{
    int *diffnum = nullptr;
    REF_CHECK(hipMalloc((void **)&diffnum, 512));
    REF_CHECK(hipMemset(diffnum, 0, sizeof(int)));
    REF_CHECK(hipDeviceSynchronize());
    hipLaunchKernelGGL(validate_int, dim3((num + 127) / 128), dim3(128), 0, 0,
                       y, y2, num, diffnum);
    REF_CHECK(hipDeviceSynchronize());
    int ans = -1;
    REF_CHECK(hipMemcpy(&ans, diffnum, sizeof(int), hipMemcpyDeviceToHost));
    hipFree(diffnum);
    *bad_out = ans;
    return 0;
}

}  // extern "C"
