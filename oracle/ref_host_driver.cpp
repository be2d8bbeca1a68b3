// oracle/ref_host_driver.cpp -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
//
// Host build of the REFERENCE's arithmetic definition, spmm_kernel_ref (PA4/workspace/src/spmm_ref.cu:3-17), so that
// the provenance of tests/golden/*.npz can be checked in a container without a GPU (SURVEY.md 8c describes exactly this
// procedure; oracle/Makefile target _ref_host).  The 15 lines of the kernel are cut out of the read-only reference tree
// at build time into a temporary file (REF_EXTRACT_INC) that is deleted after the compile; no reference text lives here.
// The kernel body is plain C once the three CUDA index variables exist as host objects and __global__ means nothing:
// this file supplies those and the launch loop of SpMMRef::run (spmm_ref.cu:20-30: block 128, grid ceil(M / 128)).
// Built with -mfma -ffp-contract=fast: the reference's nvcc --use_fast_math build contracts a*b+c to fma
// (PA4/workspace/CMakeLists.txt:46); an unfused build of the same loop differs in ~50 % of the elements.
#include <cstdint>

#define __global__
struct Idx3 { int x, y, z; };
static thread_local Idx3 blockIdx, blockDim, threadIdx;

#include REF_EXTRACT_INC

extern "C" int ref_host_spmm(int *ptr, int *idx, float *val, float *vin, float *vout, int num_v, int feat_in)
{
    const int BLOCK_SIZE = 128;
    blockDim = {BLOCK_SIZE, 1, 1};
    const int grid = (num_v + BLOCK_SIZE - 1) / BLOCK_SIZE;
    for (int b = 0; b < grid; ++b)
        for (int t = 0; t < BLOCK_SIZE; ++t) {
            blockIdx = {b, 0, 0};
            threadIdx = {t, 0, 0};
            spmm_kernel_ref(ptr, idx, val, vin, vout, num_v, feat_in);
        }
    return 0;
}
