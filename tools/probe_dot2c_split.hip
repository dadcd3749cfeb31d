// Is x - bf16_rne(x) computed exactly by ONE v_dot2c_f32_bf16 (acc = x, A = packed (hi0, hi1), B = packed (-1, 0) or (0, -1))?
// Compares against the shift / and / v_sub_f32 form of query_kernel's split_pair over random and edge-case floats.  Run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
#include <random>
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
__global__ void k(const float* x, unsigned* out, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (2 * i + 1 >= n) return;
    const float x0 = x[2 * i], x1 = x[2 * i + 1];
    const bf16x2 hv = {(__bf16)x0, (__bf16)x1};
    unsigned hpk = __builtin_bit_cast(unsigned, hv);
    asm volatile("" : "+v"(hpk));
    const float r0 = x0 - __uint_as_float(hpk << 16), r1 = x1 - __uint_as_float(hpk & 0xffff0000u);
    float d0 = x0, d1 = x1;
    const unsigned c0 = 0x0000bf80u, c1 = 0xbf800000u; // (-1, 0) and (0, -1) as packed bf16 (low half first)
    asm volatile("v_dot2c_f32_bf16 %0, %2, %3\n\tv_dot2c_f32_bf16 %1, %2, %4" : "+v"(d0), "+v"(d1) : "v"(hpk), "v"(c0), "v"(c1));
    out[4 * i] = __float_as_uint(r0); out[4 * i + 1] = __float_as_uint(d0); out[4 * i + 2] = __float_as_uint(r1); out[4 * i + 3] = __float_as_uint(d1);
}
int main()
{
    const int n = 1 << 22;
    std::vector<float> h(n);
    std::mt19937 g(1);
    for (int i = 0; i < n; ++i) {
        unsigned u = g();
        if (i % 4 == 0) { float f = (float)((int)(u % 2000001) - 1000000) * 1e-6f; h[i] = f; }          // [-1, 1]
        else if (i % 4 == 1) { float f = std::ldexp((float)(u & 0xffffff) / 16777216.0f + 1.0f, (int)((u >> 24) % 40) - 20); h[i] = (u & 0x80000000u) ? -f : f; }
        else { u &= 0xbfffffffu; if ((u & 0x7f800000u) == 0x7f800000u) u &= 0xbf7fffffu; std::memcpy(&h[i], &u, 4); } // any finite below 2.0, incl. denormals
    }
    h[0] = 0.0f; h[1] = -0.0f; h[2] = 1.0f; h[3] = 1.00390625f; h[4] = 3.0e-39f; h[5] = 65504.0f;
    float* dx; unsigned* dout;
    hipMalloc(&dx, n * 4); hipMalloc(&dout, n * 2 * 4);
    hipMemcpy(dx, h.data(), n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 2 / 256), dim3(256), 0, 0, dx, dout, n);
    std::vector<unsigned> o(n * 2);
    hipMemcpy(o.data(), dout, n * 2 * 4, hipMemcpyDeviceToHost);
    long bad = 0, bad_norm = 0;
    for (int i = 0; i < n / 2; ++i)
        for (int e = 0; e < 2; ++e) {
            const unsigned r = o[4 * i + 2 * e], d = o[4 * i + 2 * e + 1];
            if (r != d) {
                float rf, df; std::memcpy(&rf, &r, 4); std::memcpy(&df, &d, 4);
                const bool tiny = std::fabs(rf) < 1.2e-38f; // a denormal difference flushed to zero would not matter (it is below bf16's range for `lo` anyway)
                if (bad < 8) printf("x = %.9g: sub %.9g (0x%08x)  dot2c %.9g (0x%08x)%s\n", h[2 * i + e], rf, r, df, d, tiny ? "  [denormal]" : "");
                ++bad; if (!tiny && !(rf == 0.0f && df == 0.0f)) ++bad_norm;
            }
        }
    printf("%ld of %d differ (%ld outside the denormal range / signed zeros)\n", bad, n, bad_norm);
    return bad_norm != 0;
}
