// Is private (scratch) memory reliable with two 256-VGPR waves per SIMD on this box?  Every lane keeps NS values in a dynamically indexed
// private array (forced to scratch), runs MFMA + VALU filler between the store and the reload, and checks what comes back.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr int NS = 10;

template <int WAVES_PER_SIMD>
__global__ __launch_bounds__(256, WAVES_PER_SIMD) void k(unsigned long long* bad, unsigned* first, float* sink, int iters, int rot)
{
    volatile unsigned priv[NS]; // volatile + dynamic index: stays in scratch
    const unsigned uid = blockIdx.x * 256u + threadIdx.x;
    f32x16 acc[12] = {}; // 192 accumulator registers: with the filler below the kernel sits at the 256-VGPR budget
    u32x4 a = {uid, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u}, b = {0x3c003c00u, 0x3c003c00u, 0x3c003c00u, uid};
    unsigned long long nbad = 0;
    for (int it = 0; it < iters; ++it) {
        for (int s = 0; s < NS; ++s) priv[(s + rot) % NS] = uid * 131u + (unsigned)it * 7919u + (unsigned)s;
#pragma unroll
        for (int m = 0; m < 12; ++m)
            acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc[m], 0, 0, 0);
        for (int s = 0; s < NS; ++s) {
            const unsigned v = priv[(s + rot) % NS], want = uid * 131u + (unsigned)it * 7919u + (unsigned)s;
            if (v != want) { if (!nbad) { first[0] = uid; first[1] = it; first[2] = s; first[3] = v; first[4] = want; } ++nbad; }
        }
    }
    float t = 0;
#pragma unroll
    for (int m = 0; m < 12; ++m)
#pragma unroll
        for (int i = 0; i < 16; ++i) t += acc[m][i];
    sink[uid] = t;
    if (nbad) atomicAdd(bad + (threadIdx.x & 63), nbad);
}

template <int W> void run(int iters)
{
    unsigned long long* bad; unsigned* first; float* sink;
    const int nb = 256 * W;
    hipMalloc(&bad, 64 * 8); hipMalloc(&first, 32); hipMalloc(&sink, nb * 256 * 4);
    hipMemset(bad, 0, 64 * 8); hipMemset(first, 0, 32);
    hipLaunchKernelGGL(k<W>, dim3(nb), dim3(256), 0, 0, bad, first, sink, iters, 3);
    hipError_t e = hipDeviceSynchronize();
    unsigned long long h[64]; unsigned f[8];
    hipMemcpy(h, bad, sizeof h, hipMemcpyDeviceToHost); hipMemcpy(f, first, sizeof f, hipMemcpyDeviceToHost);
    unsigned long long tot = 0; for (int i = 0; i < 64; ++i) tot += h[i];
    printf("%d wave(s)/SIMD, %d blocks, %d iterations x %d slots per lane: %s, mismatches %llu", W, nb, iters, NS, hipGetErrorString(e), tot);
    if (tot) { printf("  per 16-lane quarter [%llu %llu %llu %llu]  first: thread %u it %u slot %u got %u want %u", 0ull, 0ull, 0ull, 0ull, f[0], f[1], f[2], f[3], f[4]);
        unsigned long long q[4] = {}; for (int i = 0; i < 64; ++i) q[i / 16] += h[i]; printf("  quarters %llu %llu %llu %llu", q[0], q[1], q[2], q[3]); }
    printf("\n");
    hipFree(bad); hipFree(first); hipFree(sink);
}
int main() { run<1>(20000); run<2>(20000); run<2>(20000); return 0; }
