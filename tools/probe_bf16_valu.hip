// How do v_mfma_f32_32x32x16_bf16 and VALU work share a SIMD on gfx950?  Each loop iteration issues 4 MFMAs on NACC accumulators
// (NACC = 1: a dependent chain) and K VALU instructions of kind OP after each MFMA.  Reports core cycles per MFMA per wave.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int K, int NACC, int OP> __global__ __launch_bounds__(256) void k(float* out, long long* cyc, int iters)
{
    f32x16 a[4] = {};
    const float x = threadIdx.x * 1e-3f, y = 1.0001f;
    u32x4 ab = {__float_as_uint(x), 0x3f803f80u, 0x3f803f80u, 0x3f803f80u}, bb = {0x3c003c00u, 0x3c003c00u, 0x3c003c00u, __float_as_uint(y)};
    float v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = x + i;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            a[m % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ab), __builtin_bit_cast(bf16x8, bb), a[m % NACC], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < K; ++i) {
                if constexpr (OP == 0) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[i]) : "v"(y));
                else if constexpr (OP == 1) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(v[i]) : "v"(y));
                else if constexpr (OP == 2) asm volatile("v_and_b32 %0, %0, %1" : "+v"(v[i]) : "v"(y));
                else asm volatile("v_exp_f32 %0, %0" : "+v"(v[i]));
            }
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += v[i] + a[0][i] + a[1][i] + a[2][i] + a[3][i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int K, int NACC, int OP> void run(int blocks_per_cu)
{
    float* out; long long* cyc;
    int nb = 256 * blocks_per_cu;
    hipMalloc(&out, nb * 256 * 4); hipMalloc(&cyc, nb * 8);
    int iters = 20000;
    hipLaunchKernelGGL((k<K, NACC, OP>), dim3(nb), dim3(256), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    static long long h[4096];
    hipMemcpy(h, cyc, nb * 8, hipMemcpyDeviceToHost);
    double m = 0; for (int i = 0; i < nb; ++i) m += h[i]; m /= nb;
    const char* ops[] = {"v_fma_f32", "v_cvt_pk_bf16_f32", "v_and_b32", "v_exp_f32"};
    printf("accs %d  K=%2d x %-18s %d wave(s)/SIMD: %6.1f cycles per MFMA per wave\n", NACC, K, ops[OP], blocks_per_cu, m / (iters * 4.0));
    hipFree(out); hipFree(cyc);
}
int main()
{
    run<0, 4, 0>(1); run<0, 1, 0>(1); run<0, 2, 0>(1);
    run<4, 4, 0>(1); run<8, 4, 0>(1); run<16, 4, 0>(1);
    run<4, 1, 0>(1); run<8, 1, 0>(1); run<16, 1, 0>(1);
    run<8, 4, 1>(1); run<8, 4, 2>(1); run<8, 4, 3>(1); run<8, 1, 1>(1); run<8, 1, 2>(1);
    run<0, 4, 0>(2); run<0, 1, 0>(2); run<8, 4, 0>(2); run<16, 4, 0>(2); run<8, 1, 0>(2); run<16, 1, 0>(2); run<8, 1, 1>(2);
    return 0;
}
