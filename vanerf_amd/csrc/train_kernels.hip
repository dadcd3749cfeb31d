// train_kernels.hip -- kernels of the training step's backward pass.
//
// vanerf_scatter_add_rows: table[idx[i]][:] += w[i] * g[i][:] for N samples -- the backward of the row gathers of the per-sample networks
// (bilinear taps of the feature maps, reference src/utils.py:136-151; nearest / twin vertex rows, src/networks.py:27-33).  N is ~8e5 per
// training step and the tables have 1e3..1.6e4 rows: as global atomics (torch's index_add_) every row is hit by hundreds of samples and
// the step spent 20 ms (28 %) there.  Here a block keeps a slice of the table's channels in LDS, adds its share of the samples with LDS
// atomics (a contended LDS atomic costs cycles, not a trip to the memory side), and flushes the slice once with global atomics.
#include <algorithm>

#include <cstdlib>

#include "common.h"

using namespace vanerf;

namespace {

constexpr int SC_BLOCK = 256;

// grid: (sample chunks, channel slices).  s_tab: [R][CS] floats.
// TAPS rows per sample (idx / w are [TAPS][ld]): the four bilinear taps of a pixel feature read its gradient row once
template <int CS, int TAPS = 1>
__global__ __launch_bounds__(SC_BLOCK) void scatter_rows_kernel(const int32_t* __restrict__ idx, const float* __restrict__ w, const float* __restrict__ g,
                                                                long long n, int C, int R, float* __restrict__ table, long long ld, long long g_ld,
                                                                const int32_t* __restrict__ idx2, const float* __restrict__ w2, const float* __restrict__ g2)
{
    extern __shared__ float s_tab[];
    const int c0 = blockIdx.y * CS;
    for (int k = threadIdx.x; k < R * CS; k += SC_BLOCK) s_tab[k] = 0.0f;
    __syncthreads();
    const int cl = threadIdx.x % CS;        // channel within the slice
    const int sub = threadIdx.x / CS;       // sample within a step of SC_BLOCK / CS samples
    constexpr int PER = SC_BLOCK / CS;
    const long long chunk = (n + gridDim.x - 1) / gridDim.x;
    const long long i0 = (long long)blockIdx.x * chunk, i1 = i0 + chunk < n ? i0 + chunk : n;
    if (c0 + cl < C) {
        // A thread owns one channel of a RUN of consecutive samples: consecutive depths of a ray keep their nearest vertex for many samples and move their
        // bilinear taps by a fraction of a pixel, so a running (row, sum) per tap stays in registers and goes to the LDS table only when the row changes
        // (the LDS float adds were the kernel's time: ~65 k of them per block and channel slice).  U samples per trip: all their loads (gradient row
        // element, indices, weights) are issued before the first use -- one sample per trip was a chain of dependent global loads.
        constexpr int U = 4, NS = TAPS + 1; // running sums: the taps, and the second source
        const long long run = (chunk + PER - 1) / PER;
        const long long a0 = i0 + (long long)sub * run, a1 = a0 + run < i1 ? a0 + run : i1;
        int cur[NS];
        float sum[NS];
#pragma unroll
        for (int k = 0; k < NS; ++k) { cur[k] = -1; sum[k] = 0.0f; }
        auto push = [&](int k, int r, float v) {
            if (r != cur[k]) {
                if ((unsigned)cur[k] < (unsigned)R) atomicAdd(&s_tab[cur[k] * CS + cl], sum[k]); // (a row outside the table contributes nothing)
                cur[k] = r; sum[k] = v;
            } else {
                sum[k] += v;
            }
        };
        for (long long ib = a0; ib < a1; ib += U) {
            float v0[U], wk[U][TAPS], v2[U], w2k[U];
            int rk[U][TAPS], r2[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const long long i = ib + u;
                const bool on = i < a1;
                const long long ic = on ? i : a0; // (a valid address for the trips past the run's end; their rows are turned off)
                v0[u] = g[ic * g_ld + c0 + cl];
#pragma unroll
                for (int k = 0; k < TAPS; ++k) {
                    rk[u][k] = on ? idx[ic + k * ld] : -1;
                    wk[u][k] = w ? w[ic + k * ld] : 1.0f;
                }
                if (TAPS == 1 && idx2) {
                    r2[u] = on ? idx2[ic] : -1;
                    v2[u] = g2[ic * g_ld + c0 + cl];
                    w2k[u] = w2 ? w2[ic] : 1.0f;
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
#pragma unroll
                for (int k = 0; k < TAPS; ++k) push(k, rk[u][k], w ? v0[u] * wk[u][k] : v0[u]);
                if (TAPS == 1 && idx2) push(TAPS, r2[u], w2 ? v2[u] * w2k[u] : v2[u]); // the nearest and the twin vertex row of a sample
            }
        }
#pragma unroll
        for (int k = 0; k < NS; ++k)
            if ((unsigned)cur[k] < (unsigned)R) atomicAdd(&s_tab[cur[k] * CS + cl], sum[k]);
    }
    __syncthreads();
    for (int k = threadIdx.x; k < R * CS; k += SC_BLOCK) {
        const float v = s_tab[k];
        const int c = c0 + k % CS;
        if (v != 0.0f && c < C) atomicAdd(&table[(size_t)(k / CS) * C + c], v);
    }
}

// the four bilinear taps of feat_sample (src/utils.py:136-151: border padding, align_corners) at xy in [-1, 1]: row indices of the channel-last
// map and weights, [4][n] each -- the arithmetic of the forward kernels' bilin_setup
__global__ __launch_bounds__(256) void bilinear_taps_kernel(const float* __restrict__ xy, long long n, int H, int W, int32_t* __restrict__ idx4,
                                                            float* __restrict__ w4)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float wm1 = (float)(W - 1), hm1 = (float)(H - 1);
    float x = (xy[2 * i] + 1.0f) * (0.5f * wm1), y = (xy[2 * i + 1] + 1.0f) * (0.5f * hm1);
    x = fminf(fmaxf(x, 0.0f), wm1); y = fminf(fmaxf(y, 0.0f), hm1);
    const float xf = floorf(x), yf = floorf(y), wx = x - xf, wy = y - yf;
    const int x0 = (int)xf, y0 = (int)yf, x1 = x0 + 1 < W ? x0 + 1 : W - 1, y1 = y0 + 1 < H ? y0 + 1 : H - 1;
    idx4[i] = y0 * W + x0; idx4[n + i] = y0 * W + x1; idx4[2 * n + i] = y1 * W + x0; idx4[3 * n + i] = y1 * W + x1;
    w4[i] = (1.0f - wx) * (1.0f - wy); w4[n + i] = wx * (1.0f - wy); w4[2 * n + i] = (1.0f - wx) * wy; w4[3 * n + i] = wx * wy;
}

} // namespace

extern "C" int vanerf_bilinear_taps(const float* xy, int64_t n, int H, int W, int32_t* idx4, float* w4, void* stream)
{
    return guarded([&] {
        if (n == 0) return;
        if (!xy || !idx4 || !w4 || n < 0 || H <= 0 || W <= 0) throw_error("vanerf_bilinear_taps: n=%lld H=%d W=%d or a null argument", (long long)n, H, W);
        hipLaunchKernelGGL(bilinear_taps_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, xy, (long long)n, H, W, idx4, w4);
        HIP_CHECK(hipGetLastError());
    });
}

// table[R][C] += scatter of w[i] * g[i][C] at rows idx[i] (w may be NULL = 1).  All device pointers; `table` is accumulated into (zero it first
// for a plain gradient).  Rows outside [0, R) are ignored.
namespace {
void scatter_rows(const int32_t* idx, const float* w, const float* g, int64_t g_ld, int64_t n, int C, float* table, int R, int taps, int64_t ld,
                  void* stream, const int32_t* idx2 = nullptr, const float* w2 = nullptr, const float* g2 = nullptr);
}

extern "C" int vanerf_scatter_add_rows(const int32_t* idx, const float* w, const float* g, int64_t g_ld, int64_t n, int C, float* table, int R,
                                       void* stream)
{
    return guarded([&] { scatter_rows(idx, w, g, g_ld, n, C, table, R, 1, 0, stream); });
}

// two sources into one table in one launch: table[idx[i]] += w[i] g[i], table[idx2[i]] += w2[i] g2[i] (g, g2 with the same row distance g_ld): the
// nearest and the twin vertex row of every sample -- one pass over the LDS slices instead of two
extern "C" int vanerf_scatter_add_rows2(const int32_t* idx, const float* w, const float* g, const int32_t* idx2, const float* w2, const float* g2,
                                        int64_t g_ld, int64_t n, int C, float* table, int R, void* stream)
{
    return guarded([&] {
        if (!idx2 || !g2) throw_error("vanerf_scatter_add_rows2: null argument");
        scatter_rows(idx, w, g, g_ld, n, C, table, R, 1, 0, stream, idx2, w2, g2);
    });
}

// the same with FOUR (index, weight) pairs per sample, idx4 / w4 = [4][ld] (ld >= n): table[idx4[k][i]] += w4[k][i] * g[i] for k < 4 -- the
// backward of a bilinear tap gather (src/utils.py:136-151) in one launch that reads every gradient row once
extern "C" int vanerf_scatter_add_taps(const int32_t* idx4, const float* w4, int64_t ld, const float* g, int64_t g_ld, int64_t n, int C, float* table,
                                       int R, void* stream)
{
    return guarded([&] {
        if (ld < n || !w4) throw_error("vanerf_scatter_add_taps: ld = %lld < n = %lld or no weights", (long long)ld, (long long)n);
        scatter_rows(idx4, w4, g, g_ld, n, C, table, R, 4, ld, stream);
    });
}

namespace {
void scatter_rows(const int32_t* idx, const float* w, const float* g, int64_t g_ld, int64_t n, int C, float* table, int R, int taps, int64_t ld,
                  void* stream, const int32_t* idx2, const float* w2, const float* g2)
{
    {
        if (n == 0) return;
        if (!idx || !g || !table) throw_error("vanerf_scatter_add_rows: null argument");
        if (n < 0 || C <= 0 || R <= 0) throw_error("vanerf_scatter_add_rows: n=%lld C=%d R=%d", (long long)n, C, R);
        if (g_ld == 0) g_ld = C;
        if (g_ld < C) throw_error("vanerf_scatter_add_rows: rows of %d channels, %lld floats apart", C, (long long)g_ld);
        // channel slice: the widest power of two <= 16 whose [R][CS] floats fit 128 KB of LDS
        int cs = 16;
        while (cs > 1 && ((size_t)R * cs * 4 > 128 * 1024 || cs / 2 >= C)) cs /= 2;
        if ((size_t)R * cs * 4 > 128 * 1024) throw_error("vanerf_scatter_add_rows: a table of %d rows does not fit the LDS slice", R);
        const int slices = (C + cs - 1) / cs;
        // blocks = chunks x slices: 4 096 samples per chunk for a whole training patch (~5e5 samples), but never fewer than ~512 blocks' worth of
        // work in flight -- the fused backward calls this once per block of 65 536 samples, where 16 chunks x 4 slices left three quarters of the chip idle
        int chunks = (int)((n + 4095) / 4096);
        static const int fill_blocks = [] { const char* e = getenv("VANERF_SCATTER_FILL"); return e ? atoi(e) : 256; }(); // (256 / 512 / 1024 blocks' worth measured: 12.2 / 12.6 / 12.6 ms of the step's fused stage)
        const int fill = (int)std::min<long long>((n + 511) / 512, fill_blocks / slices > 0 ? fill_blocks / slices : 1);
        if (chunks < fill) chunks = fill;
        if (chunks > 1024 / slices) chunks = 1024 / slices > 0 ? 1024 / slices : 1;
        if (chunks < 1) chunks = 1;
        const size_t lds = (size_t)R * cs * 4;
        const dim3 grid((unsigned)chunks, (unsigned)slices);
        const hipStream_t st = (hipStream_t)stream;
#define LAUNCH1(CS, TAPS)                                                                                                                  \
    do {                                                                                                                                   \
        if (lds > 64 * 1024) HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(scatter_rows_kernel<CS, TAPS>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
        hipLaunchKernelGGL((scatter_rows_kernel<CS, TAPS>), grid, dim3(SC_BLOCK), lds, st, idx, w, g, (long long)n, C, R, table, (long long)ld, (long long)g_ld, idx2, w2, g2); \
    } while (0)
#define LAUNCH(CS) do { if (taps == 4) LAUNCH1(CS, 4); else LAUNCH1(CS, 1); } while (0)
        switch (cs) {
        case 16: LAUNCH(16); break;
        case 8: LAUNCH(8); break;
        case 4: LAUNCH(4); break;
        case 2: LAUNCH(2); break;
        default: LAUNCH(1); break;
        }
#undef LAUNCH
#undef LAUNCH1
        HIP_CHECK(hipGetLastError());
    }
}
} // namespace
