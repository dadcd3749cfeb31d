// Does fp32 VALU work overlap with v_mfma_f32_32x32x2_f32 on gfx950?  One wave per SIMD; each loop iteration issues 4 independent
// MFMAs and K independent v_fma_f32 per MFMA.  Also: two waves per SIMD, one MFMA-only and one VALU-only.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int K> __global__ __launch_bounds__(256) void k(float* out, long long* cyc, int iters)
{
    f32x16 a0 = {}, a1 = {}, a2 = {}, a3 = {};
    float x = threadIdx.x * 1e-3f, y = 1.0001f;
    float v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = x + i;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#define STEP(acc)                                                            \
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc, 0, 0, 0);      \
        _Pragma("unroll") for (int i = 0; i < K; ++i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[i]) : "v"(y));
        STEP(a0) STEP(a1) STEP(a2) STEP(a3)
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += v[i] + a0[i] + a1[i] + a2[i] + a3[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int K> void run(const char* name, int blocks_per_cu)
{
    float* out; long long* cyc;
    int nb = 256 * blocks_per_cu;
    hipMalloc(&out, nb * 256 * 4); hipMalloc(&cyc, nb * 8);
    int iters = 20000;
    hipLaunchKernelGGL(k<K>, dim3(nb), dim3(256), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    long long h[4096];
    hipMemcpy(h, cyc, nb * 8, hipMemcpyDeviceToHost);
    double m = 0; for (int i = 0; i < nb; ++i) m += h[i]; m /= nb;
    printf("%s K=%2d VALU/MFMA, %d wave(s)/SIMD: %.1f cycles per MFMA (per wave)\n", name, K, blocks_per_cu, m / (iters * 4.0));
    hipFree(out); hipFree(cyc);
}
int main()
{
    run<0>("mfma+valu", 1); run<4>("mfma+valu", 1); run<8>("mfma+valu", 1); run<12>("mfma+valu", 1); run<16>("mfma+valu", 1);
    run<0>("mfma+valu", 2); run<8>("mfma+valu", 2); run<16>("mfma+valu", 2);
    return 0;
}
