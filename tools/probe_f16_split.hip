// Can the operand split of the three-product MFMA scheme run on fp16 instead of bf16?  (x = hi + lo, hi = f16(x), lo = f16(x - hi))
//   1. numerics of v_mfma_f32_32x32x16_f16 on SUBNORMAL fp16 inputs (the lo parts of small activations are subnormal in fp16)
//   2. the 3-instruction split  v_cvt_pk_f16_f32 / v_fma_mixlo_f16 / v_fma_mixhi_f16  against fp64
//   3. cycles per MFMA with P operand pairs split beside each MFMA: bf16 (cvt_pk, shl, and, sub, sub, cvt_pk) against fp16 (cvt_pk, mixlo, mixhi)
// build: hipcc --offload-arch=gfx950 -O3 tools/probe_f16_split.hip -o /tmp/probe_f16_split
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void split_f16(float x0, float x1, unsigned& hi, unsigned& lo)
{
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(hi) : "v"(x0), "v"(x1));
    asm("v_fma_mixlo_f16 %0, %1, 1.0, -%2 op_sel_hi:[0,0,1]" : "=v"(lo) : "v"(x0), "v"(hi));
    asm("v_fma_mixhi_f16 %0, %1, 1.0, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(lo) : "v"(x1), "v"(hi));
}

// D[i][j] = sum_k A[i][k] B[k][j]; lane l supplies A[i = l & 31][k = 8 (l >> 5) + 0..7] and B[k = 8 (l >> 5) + 0..7][j = l & 31]
__global__ void numerics(const float* __restrict__ xa, const float* __restrict__ xb, float* __restrict__ d, float* __restrict__ split_err)
{
    const int l = threadIdx.x;
    unsigned ah[4], al[4], bh[4], bl[4];
    float worst = 0.0f;
    for (int p = 0; p < 4; ++p) {
        const float a0 = xa[l * 8 + 2 * p], a1 = xa[l * 8 + 2 * p + 1], b0 = xb[l * 8 + 2 * p], b1 = xb[l * 8 + 2 * p + 1];
        split_f16(a0, a1, ah[p], al[p]);
        split_f16(b0, b1, bh[p], bl[p]);
        const _Float16 h0 = __builtin_bit_cast(_Float16, (unsigned short)(ah[p] & 0xffff)), l0 = __builtin_bit_cast(_Float16, (unsigned short)(al[p] & 0xffff));
        const _Float16 h1 = __builtin_bit_cast(_Float16, (unsigned short)(ah[p] >> 16)), l1 = __builtin_bit_cast(_Float16, (unsigned short)(al[p] >> 16));
        const float e0 = fabsf((float)((double)a0 - (double)(float)h0 - (double)(float)l0)) / fmaxf(fabsf(a0), 1e-30f);
        const float e1 = fabsf((float)((double)a1 - (double)(float)h1 - (double)(float)l1)) / fmaxf(fabsf(a1), 1e-30f);
        worst = fmaxf(worst, fmaxf(e0, e1));
    }
    split_err[l] = worst;
    const u32x4 AH = {ah[0], ah[1], ah[2], ah[3]}, AL = {al[0], al[1], al[2], al[3]}, BH = {bh[0], bh[1], bh[2], bh[3]}, BL = {bl[0], bl[1], bl[2], bl[3]};
    f32x16 acc = {};
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, AH), __builtin_bit_cast(f16x8, BH), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, AH), __builtin_bit_cast(f16x8, BL), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, AL), __builtin_bit_cast(f16x8, BH), acc, 0, 0, 0);
    for (int r = 0; r < 16; ++r) d[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = acc[r];
}

template <int P, int KIND> __global__ __launch_bounds__(256) void timing(float* out, long long* cyc, int iters)
{
    f32x16 a[4] = {};
    const float x = threadIdx.x * 1e-3f + 0.37f;
    u32x4 ab = {0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u}, bb = ab;
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = x + i;
    unsigned sink = 0;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            if constexpr (KIND == 0) a[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ab), __builtin_bit_cast(bf16x8, bb), a[m], 0, 0, 0);
            else a[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, ab), __builtin_bit_cast(f16x8, bb), a[m], 0, 0, 0);
#pragma unroll
            for (int p = 0; p < P; ++p) {
                unsigned hi, lo;
                float x0 = v[(2 * p) & 7], x1 = v[(2 * p + 1) & 7];
                if constexpr (KIND == 0) {
                    float t0_, t1_;
                    asm volatile("v_cvt_pk_bf16_f32 %0, %4, %5\n\ts_nop 0\n\tv_lshlrev_b32 %2, 16, %0\n\tv_and_b32 %3, 0xffff0000, %0\n\tv_sub_f32 %2, %4, %2\n\tv_sub_f32 %3, %5, %3\n\t"
                                 "v_cvt_pk_bf16_f32 %1, %2, %3"
                                 : "=&v"(hi), "=&v"(lo), "=&v"(t0_), "=&v"(t1_) : "v"(x0), "v"(x1));
                } else {
                    asm volatile("v_cvt_pk_f16_f32 %0, %2, %3\n\ts_nop 0\n\tv_fma_mixlo_f16 %1, %2, 1.0, -%0 op_sel_hi:[0,0,1]\n\tv_fma_mixhi_f16 %1, %3, 1.0, -%0 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
                                 : "=&v"(hi), "=&v"(lo) : "v"(x0), "v"(x1));
                }
                sink ^= hi ^ lo;
            }
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    float s = __uint_as_float(sink & 0xff);
#pragma unroll
    for (int i = 0; i < 16; ++i) s += a[0][i] + a[1][i] + a[2][i] + a[3][i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int P, int KIND> void run_timing()
{
    float* out; long long* cyc;
    const int nb = 256, iters = 20000;
    hipMalloc(&out, nb * 256 * 4); hipMalloc(&cyc, nb * 8);
    hipLaunchKernelGGL((timing<P, KIND>), dim3(nb), dim3(256), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    static long long h[256];
    hipMemcpy(h, cyc, nb * 8, hipMemcpyDeviceToHost);
    double m = 0; for (int i = 0; i < nb; ++i) m += h[i]; m /= nb;
    printf("%s split, %d pairs per MFMA: %6.1f cycles per MFMA per wave (one wave per SIMD)\n", KIND ? "fp16 (3 instr/pair)" : "bf16 (6 instr/pair)", P, m / (iters * 4.0));
    hipFree(out); hipFree(cyc);
}

int main()
{
    // ---- numerics: scale sweeps down into the fp16 subnormal range (2^-14 = 6.1e-5 is the smallest normal fp16) ----
    for (float scale : {1.0f, 1e-2f, 1e-4f, 3e-6f}) {
        static float ha[512], hb[512], hd[1024], he[64];
        srand(7);
        for (int i = 0; i < 512; ++i) { ha[i] = scale * (rand() / (float)RAND_MAX - 0.5f); hb[i] = rand() / (float)RAND_MAX - 0.5f; }
        float *da, *db, *dd, *de;
        hipMalloc(&da, sizeof ha); hipMalloc(&db, sizeof hb); hipMalloc(&dd, sizeof hd); hipMalloc(&de, sizeof he);
        hipMemcpy(da, ha, sizeof ha, hipMemcpyHostToDevice); hipMemcpy(db, hb, sizeof hb, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(numerics, dim3(1), dim3(64), 0, 0, da, db, dd, de);
        hipMemcpy(hd, dd, sizeof hd, hipMemcpyDeviceToHost); hipMemcpy(he, de, sizeof he, hipMemcpyDeviceToHost);
        double worst = 0, ref_max = 0, serr = 0;
        for (int i = 0; i < 32; ++i)
            for (int j = 0; j < 32; ++j) {
                double r = 0;
                for (int k = 0; k < 16; ++k) r += (double)ha[(i + 32 * (k / 8)) * 8 + k % 8] * (double)hb[(j + 32 * (k / 8)) * 8 + k % 8];
                worst = fmax(worst, fabs(r - hd[i * 32 + j])); ref_max = fmax(ref_max, fabs(r));
            }
        for (int i = 0; i < 64; ++i) serr = fmax(serr, he[i]);
        printf("A scale %-8g: max |D - fp64| = %.3e (max |D| %.3e, ratio %.2e); worst relative split residual |x - hi - lo| / |x| = %.2e\n", scale, worst, ref_max,
               worst / ref_max, serr);
        hipFree(da); hipFree(db); hipFree(dd); hipFree(de);
    }
    run_timing<0, 0>(); run_timing<0, 1>();
    run_timing<1, 0>(); run_timing<1, 1>();
    run_timing<2, 0>(); run_timing<2, 1>();
    run_timing<3, 0>(); run_timing<3, 1>();
    return 0;
}
