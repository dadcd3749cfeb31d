// How fast can the forward-spill's store pattern go?  256 blocks x 4 waves, every wave writes ROWS rows of 32 floats per 32-sample group
// (lanes 0..31 row r, lanes 32..63 row r + 1: two 128-byte segments per store instruction), two layouts:
//   channel-major  xs[row][npad]      (rows 4 npad bytes apart: what query_kernel<0, SPILL> does today)
//   group-major    xs[group][row][32] (a group's rows contiguous)
// and the read side (the backward chain's loads) of both.  Build: hipcc --offload-arch=gfx950 -O2 tools/probe_spill_layout.hip -o exp/probe_spill_layout
#include <hip/hip_runtime.h>
#include <cstdio>
constexpr int ROWS = 2074;
template <int GROUP_MAJOR, int READ>
__global__ __launch_bounds__(256) void k(float* xs, long long npad, float* sink)
{
    const int lane = threadIdx.x & 63, j = lane & 31, h = lane >> 5;
    const long long ngroups = npad / 32, nwaves = (long long)gridDim.x * 4;
    float acc = 0.0f;
    for (long long g = (long long)blockIdx.x * 4 + (threadIdx.x >> 6); g < ngroups; g += nwaves) {
        float* base = GROUP_MAJOR ? xs + (size_t)g * ROWS * 32 + j : xs + (size_t)g * 32 + j;
        const size_t rs = GROUP_MAJOR ? 32 : (size_t)npad;
#pragma unroll 8
        for (int r = 0; r < ROWS; r += 2) {
            if (READ) acc += base[(size_t)(r + h) * rs];
            else base[(size_t)(r + h) * rs] = (float)(r + lane);
        }
    }
    if (READ && acc == 123.456f) sink[0] = acc;
}
template <int GM, int RD> float run(float* xs, long long npad, float* sink)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k<GM, RD>), dim3(256), dim3(256), 0, 0, xs, npad, sink);
    hipEventRecord(a);
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL((k<GM, RD>), dim3(256), dim3(256), 0, 0, xs, npad, sink);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms / 10;
}
int main()
{
    for (long long npad : {65536LL, 262144LL}) {
        float *xs, *sink; hipMalloc(&xs, (size_t)ROWS * npad * 4); hipMalloc(&sink, 4);
        hipMemset(xs, 0, (size_t)ROWS * npad * 4);
        const double gb = (double)ROWS * npad * 4 / 1e9;
        float t;
        t = run<0, 0>(xs, npad, sink); printf("npad %7lld  write channel-major %.3f ms  %.2f TB/s\n", npad, t, gb / t);
        t = run<1, 0>(xs, npad, sink); printf("npad %7lld  write group-major   %.3f ms  %.2f TB/s\n", npad, t, gb / t);
        t = run<0, 1>(xs, npad, sink); printf("npad %7lld  read  channel-major %.3f ms  %.2f TB/s\n", npad, t, gb / t);
        t = run<1, 1>(xs, npad, sink); printf("npad %7lld  read  group-major   %.3f ms  %.2f TB/s\n", npad, t, gb / t);
        hipFree(xs); hipFree(sink);
    }
    return 0;
}
