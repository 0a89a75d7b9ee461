// Do f32 MFMAs and f32 VALU work of ANOTHER wave on the same SIMD overlap on gfx950?
// 512-thread workgroups (2 waves per SIMD), one per CU.  Modes: 0 every wave MFMA only; 1 every wave VALU (fma) only;
// 2 waves 0-3 MFMA, waves 4-7 VALU fma; 3 waves 0-3 MFMA, 4-7 transcendental (v_exp); 4 only waves 0-3 MFMA (4-7 idle);
// 5 only waves 4-7 VALU.   build: hipcc --offload-arch=gfx950 -O3 -o tools/bin/mfma_valu_coexec tools/mfma_valu_coexec.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE>
__global__ void __launch_bounds__(512) k(float* out, int iters)
{
    const int wave = threadIdx.x >> 6;
    const bool mf = MODE == 0 || ((MODE == 2 || MODE == 3 || MODE == 4) && wave < 4);
    const bool va = MODE == 1 || ((MODE == 2 || MODE == 5) && wave >= 4);
    const bool tr = MODE == 3 && wave >= 4;
    float a = threadIdx.x * 1e-3f, b = 1.0001f;
    f32x16 acc0 = {0}, acc1 = {0}, acc2 = {0}, acc3 = {0};
    float v0 = a, v1 = a + 1, v2 = a + 2, v3 = a + 3, v4 = a + 4, v5 = a + 5, v6 = a + 6, v7 = a + 7;
    if (mf) {
        for (int i = 0; i < iters; i++) {
#pragma unroll
            for (int u = 0; u < 4; u++) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc1, 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc2, 0, 0, 0);
                acc3 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc3, 0, 0, 0);
            }
        }
    } else if (va) {
        for (int i = 0; i < iters; i++) {
#pragma unroll
            for (int u = 0; u < 32; u++) {   // 16 MFMAs x 64 cycles = 1024 cycles ~ 256 fma (4 cycles each for one wave)
                v0 = fmaf(v0, b, a); v1 = fmaf(v1, b, a); v2 = fmaf(v2, b, a); v3 = fmaf(v3, b, a);
                v4 = fmaf(v4, b, a); v5 = fmaf(v5, b, a); v6 = fmaf(v6, b, a); v7 = fmaf(v7, b, a);
            }
        }
    } else if (tr) {
        for (int i = 0; i < iters; i++) {
#pragma unroll
            for (int u = 0; u < 16; u++) {   // 128 v_exp x 8 cycles
                v0 = __builtin_amdgcn_exp2f(v0); v1 = __builtin_amdgcn_exp2f(v1); v2 = __builtin_amdgcn_exp2f(v2); v3 = __builtin_amdgcn_exp2f(v3);
                v4 = __builtin_amdgcn_exp2f(v4); v5 = __builtin_amdgcn_exp2f(v5); v6 = __builtin_amdgcn_exp2f(v6); v7 = __builtin_amdgcn_exp2f(v7);
            }
        }
    }
    float s = v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7;
    for (int q = 0; q < 16; q++) s += acc0[q] + acc1[q] + acc2[q] + acc3[q];
    if (s == 123.456f) out[threadIdx.x] = s;
}

template <int MODE> float run(float* d, int iters)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, d, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, d, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}

int main()
{
    float* d; hipMalloc(&d, 4096);
    const int iters = 20000;
    const double mfma_flop = 256.0 * 4 /*waves*/ * iters * 16.0 * 2 * 32 * 32 * 2;   // per 4 MFMA waves per CU
    const double fma_flop = 256.0 * 4 * iters * 256.0 * 64 * 2;
    float t0 = run<0>(d, iters), t1 = run<1>(d, iters), t2 = run<2>(d, iters), t3 = run<3>(d, iters), t4 = run<4>(d, iters), t5 = run<5>(d, iters);
    printf("0 all 8 waves MFMA        %.3f ms  %.1f TF\n", t0, 2 * mfma_flop / t0 / 1e9);
    printf("1 all 8 waves fma         %.3f ms  %.1f TF\n", t1, 2 * fma_flop / t1 / 1e9);
    printf("4 waves 0-3 MFMA alone    %.3f ms  %.1f TF\n", t4, mfma_flop / t4 / 1e9);
    printf("5 waves 4-7 fma alone     %.3f ms  %.1f TF\n", t5, fma_flop / t5 / 1e9);
    printf("2 MFMA || fma             %.3f ms  (sum of alone %.3f, max %.3f)\n", t2, t4 + t5, t4 > t5 ? t4 : t5);
    printf("3 MFMA || v_exp           %.3f ms\n", t3);
    return 0;
}
