// fma_chain_micro.hip -- EXPERIMENT: what does ONE dependent v_fmac_f32 cost a single wave on a SIMD, alone or next to
// LDS reads?  (The floor of an exact stored-order hub row: one dependent fma per nonzero.)
//   hipcc -O3 --offload-arch=gfx950 scripts/experiments/fma_chain_micro.hip -o scripts/experiments/build/fma_chain_micro
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
#define F4(a, b) asm volatile("v_fmac_f32 %0, %1, %2\n\tv_fmac_f32 %0, %1, %2\n\tv_fmac_f32 %0, %1, %2\n\tv_fmac_f32 %0, %1, %2" : "+v"(a) : "v"(b), "v"(c))
template <int MODE>
__global__ void k(float *out, unsigned long long *t, int iters)
{
    __shared__ float lds[4096];
    lds[threadIdx.x] = threadIdx.x;
    __syncthreads();
    float acc = out[threadIdx.x], acc2 = acc + 1.f, b = 1.0001f, c = 0.5f;
    typedef float f4x __attribute__((ext_vector_type(4)));
    f4x cur_b[16], cur_a[16], keep_b = {0, 0, 0, 0}, keep_a = {0, 0, 0, 0};
#pragma unroll
    for (int u = 0; u < 16; ++u) { cur_b[u] = *reinterpret_cast<const f4x *>(&lds[(threadIdx.x & 31) * 68 + 4 * u]); cur_a[u] = *reinterpret_cast<const f4x *>(&lds[2200 + 4 * u]); }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {          // 64 dependent fmas, nothing else
#pragma unroll
            for (int u = 0; u < 16; ++u) F4(acc, b);
        } else if (MODE == 1) {   // two independent chains interleaved (same instruction count per chain)
#pragma unroll
            for (int u = 0; u < 64; ++u) {
                asm volatile("v_fmac_f32 %0, %2, %3\n\tv_fmac_f32 %1, %2, %3" : "+v"(acc), "+v"(acc2) : "v"(b), "v"(c));
            }
        } else if (MODE == 2) {   // 64 dependent fmas fed by 32 ds_read_b128 (the chain wave's stage)
            typedef float f4 __attribute__((ext_vector_type(4)));
            f4 bb[16], aa[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) bb[u] = *reinterpret_cast<const f4 *>(&lds[(threadIdx.x & 31) * 68 + 4 * u]);
#pragma unroll
            for (int u = 0; u < 16; ++u) aa[u] = *reinterpret_cast<const f4 *>(&lds[2200 + 4 * u]);
#pragma unroll
            for (int u = 0; u < 64; ++u) acc = __builtin_fmaf(bb[u / 4][u % 4], aa[u / 4][u % 4], acc);
            asm volatile("" ::: "memory");
        } else if (MODE == 5 || MODE == 6) {   // software-pipelined: the NEXT trip's reads interleaved with this trip's fmas
            typedef float f4 __attribute__((ext_vector_type(4)));
            static_assert(true, "");
            f4 bb[16], aa[16], nb[16], na[16];
            if (i == 0) {
#pragma unroll
                for (int u = 0; u < 16; ++u) { bb[u] = *reinterpret_cast<const f4 *>(&lds[(threadIdx.x & 31) * 68 + 4 * u]); aa[u] = *reinterpret_cast<const f4 *>(&lds[2200 + 4 * u]); }
                keep_b = bb[0]; keep_a = aa[0];
            }
#pragma unroll
            for (int u = 0; u < 16; ++u) { bb[u] = cur_b[u]; aa[u] = cur_a[u]; }
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                nb[u] = *reinterpret_cast<const f4 *>(&lds[((threadIdx.x + i) & 31) * 68 + 4 * u]);
                if (MODE == 5) na[u] = *reinterpret_cast<const f4 *>(&lds[2200 + 4 * u + (i & 1)]);
                else na[u] = aa[u];
#pragma unroll
                for (int e = 0; e < 4; ++e) acc = __builtin_fmaf(bb[u][e], aa[u][e], acc);
                __builtin_amdgcn_sched_group_barrier(0x100, MODE == 5 ? 2 : 1, 0);   // DS reads
                __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);                    // VALU
            }
#pragma unroll
            for (int u = 0; u < 16; ++u) { cur_b[u] = nb[u]; cur_a[u] = na[u]; }
        } else if (MODE == 3) {   // v_fma_f32 (VOP3) dependent instead of v_fmac (VOP2)
#pragma unroll
            for (int u = 0; u < 64; ++u) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(b), "v"(c));
        } else if (MODE == 4) {   // dependent chain with an independent VALU op between the links
#pragma unroll
            for (int u = 0; u < 64; ++u) asm volatile("v_fmac_f32 %0, %2, %3\n\tv_mov_b32 %1, %2" : "+v"(acc), "+v"(acc2) : "v"(b), "v"(c));
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[threadIdx.x] = acc + acc2 + keep_b[0] + keep_a[0] + cur_b[3][1] + cur_a[2][2];
    if (threadIdx.x == 0) t[0] = t1 - t0;
}
int main()
{
    float *d; unsigned long long *t; CK(hipMalloc(&d, 4096)); CK(hipMalloc(&t, 8)); CK(hipMemset(d, 0, 4096));
    const int iters = 2000;
    const char *names[] = {"64 dependent v_fmac", "2 x 64 interleaved chains", "32 ds_read_b128 + 64 dependent fma", "64 dependent v_fma (VOP3)", "64 x (v_fmac + independent v_mov)", "pipelined: 32 ds_read_b128 (next) among 64 fma", "pipelined: 16 ds_read_b128 (next) among 64 fma"};
    for (int rep = 0; rep < 2; ++rep) {
        for (int m = 0; m < 7; ++m) {
            if (m == 0) hipLaunchKernelGGL(k<0>, dim3(1), dim3(64), 0, 0, d, t, iters);
            if (m == 1) hipLaunchKernelGGL(k<1>, dim3(1), dim3(64), 0, 0, d, t, iters);
            if (m == 2) hipLaunchKernelGGL(k<2>, dim3(1), dim3(64), 0, 0, d, t, iters);
            if (m == 3) hipLaunchKernelGGL(k<3>, dim3(1), dim3(64), 0, 0, d, t, iters);
            if (m == 4) hipLaunchKernelGGL(k<4>, dim3(1), dim3(64), 0, 0, d, t, iters);
            if (m == 5) hipLaunchKernelGGL(k<5>, dim3(1), dim3(64), 0, 0, d, t, iters);
            if (m == 6) hipLaunchKernelGGL(k<6>, dim3(1), dim3(64), 0, 0, d, t, iters);
            unsigned long long h; CK(hipMemcpy(&h, t, 8, hipMemcpyDeviceToHost));
            printf("%-40s %.1f ticks per 64-link trip = %.2f per link\n", names[m], (double)h / iters, (double)h / iters / 64);
        }
    }
    return 0;
}
