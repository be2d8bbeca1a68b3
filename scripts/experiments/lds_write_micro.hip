// lds_write_micro.hip -- EXPERIMENT, not part of the library.  What does SQ_LDS_BANK_CONFLICT count for ds_write_b128?
// k_linear: lane-linear 16-byte writes (conflict-free under any banking rule).  k_hub0 / k_hub1: the hub loaders' transposed writes with the
// round-3 lane map (4 parts x 4 groups per 16 lanes) and with 2 parts x 4 groups per 8 lanes.  k_b64 / k_b32: narrower linear writes.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
constexpr int ITERS = 4096;
template <int MODE> __global__ __launch_bounds__(64) void k(float *out)
{
    __shared__ __attribute__((aligned(16))) float lds[32 * 68 + 64];
    const int lane = threadIdx.x;
    int off;      // in floats
    if (MODE == 0) off = 4 * lane;
    else {
        const int LPS = 8, PH = MODE == 1 ? LPS / 4 : LPS / 2;
        const int part = MODE == 1 ? (lane & 3) + 4 * ((lane >> 4) % PH) : (lane & 1) + 2 * ((lane >> 3) % PH);
        const int g = MODE == 1 ? ((lane >> 2) & 3) + 4 * ((lane >> 4) / PH) : ((lane >> 1) & 3) + 4 * ((lane >> 3) / PH);
        off = (4 * part) * 68 + 4 * g;
    }
    f4 v = {1.f, 2.f, 3.f, (float)lane};
    for (int i = 0; i < ITERS; ++i) {
        if (MODE == 3) *reinterpret_cast<volatile f2 *>(&lds[2 * lane]) = (f2){v[0], v[1]};
        else if (MODE == 4) *reinterpret_cast<volatile float *>(&lds[lane]) = v[0];
        else *reinterpret_cast<volatile f4 *>(&lds[off]) = v;
    }
    __syncthreads();
    out[lane] = lds[lane];
}
int main()
{
    float *d; CK(hipMalloc(&d, 1024));
    hipLaunchKernelGGL(k<0>, dim3(1), dim3(64), 0, 0, d);
    hipLaunchKernelGGL(k<1>, dim3(1), dim3(64), 0, 0, d);
    hipLaunchKernelGGL(k<2>, dim3(1), dim3(64), 0, 0, d);
    hipLaunchKernelGGL(k<3>, dim3(1), dim3(64), 0, 0, d);
    hipLaunchKernelGGL(k<4>, dim3(1), dim3(64), 0, 0, d);
    CK(hipDeviceSynchronize());
    printf("%d writes per lane and kernel: k<0> b128 lane-linear, k<1> b128 hub map (4x4 per 16 lanes), k<2> b128 hub map (2x4 per 8 lanes), k<3> b64 linear, k<4> b32 linear\n", ITERS);
    return 0;
}
