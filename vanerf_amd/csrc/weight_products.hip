// weight_products.hip -- the weight gradients of the fused backward pass: dW'_l[out][slot] += Ys_l Xs_l^T over a block's samples, all twenty layers
// in ONE launch (hip_backward.py did this with one sliced torch.baddbmm per layer: 160 launches and 3.7 ms per training step, the library GEMMs
// at ~1.7 TB/s on a 128 x 360 output over 65 536 samples).
// * Shape of the work: outputs are tiny (965 x ~2 074 values in all), the reduction dimension is the block's samples.  The samples are cut into
//   `slices`; a wave owns (a group of up to 2 x 4 tiles of 32 x 32 outputs of one layer, one slice) and accumulates in place into
//   dw[layer][slice][out][slot] -- exactly one wave per accumulator per launch: no atomics, sums in a fixed order, reproducible run to run.
// * Operands: v_mfma_f32_32x32x2_f32, A = Ys rows (lane = out row, k half), B = Xs rows (lane = slot, k half); both spills are channel-major with the
//   samples contiguous, so a lane streams ITS row 16 bytes at a time: 64 bytes at a time.
// * Rows beyond a layer's own (partial tiles) are loaded from a clamped row and zeroed; tiles beyond a task's count are skipped wave-uniformly.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <mutex>
#include <vector>

#include "common.h"
#include "mfma_chain.h"

using namespace vanerf;
using namespace vanerf_chain;

namespace {

constexpr int MG = 2, NG = 4; // tiles of a wave's group: 64 output rows x 128 slots
struct WpTask {
    int y_row, x_row;   // first row of the group in Ys / Xs
    int n_out, n_slot;  // rows of the layer left from there (the group uses min(., 32 MG) / min(., 32 NG))
    int ld;             // slots of the layer (row length of its accumulator)
    unsigned dw_off;    // offset of the layer's accumulator [n_out_layer][ld] inside a slice of the accumulator, in floats ...
    unsigned dw_layer;  // ... and the floats per slice (all layers)
    int out0, slot0;    // the group's first output row / slot inside the layer
};

__global__ __launch_bounds__(256, 1) void weight_products_kernel(const WpTask* __restrict__ tasks, int ntasks, const float* __restrict__ xs,
                                                                 const float* __restrict__ ys, long long npad, int slices, float* __restrict__ dw)
{
    const int lane = threadIdx.x & 63, i = lane & 31, kh = lane >> 5;
    const int tg = blockIdx.x / slices, slice = blockIdx.x % slices;
    const int ti = __builtin_amdgcn_readfirstlane(tg * 4 + (int)(threadIdx.x >> 6));
    if (ti >= ntasks) return;
    const WpTask T = tasks[ti];
    const int mt_n = (min(T.n_out, 32 * MG) + 31) / 32, nt_n = (min(T.n_slot, 32 * NG) + 31) / 32; // wave-uniform
    const long long per = npad / slices, s0 = (long long)slice * per;
    // whether this lane's operand rows exist in the layer (partial tiles: the loads below run over the next layer's rows, the operands are zeroed)
    bool av[MG], bv[NG];
#pragma unroll
    for (int m = 0; m < MG; ++m) av[m] = 32 * m + i < T.n_out;
#pragma unroll
    for (int n = 0; n < NG; ++n) bv[n] = 32 * n + i < T.n_slot;
    f32x16 acc[MG][NG];
#pragma unroll
    for (int m = 0; m < MG; ++m)
#pragma unroll
        for (int n = 0; n < NG; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.0f;
    // One batch = 32 samples of the group's 64 + 128 operand rows.  Global side: instruction t fetches rows 8 t .. 8 t + 7, eight lanes per row, each lane
    // 16 bytes -- whole 128-byte lines (a lane streaming its own row 16 bytes per visit fetched every line four times: 0.80 ms per block; 64 bytes per
    // visit still left each instruction touching 32 lines: 0.65 ms).  The wave's own LDS window turns the rows around: lane (i, kh) reads samples 16 kh .. + 15
    // of row i as its MFMA operands (rows padded to 144 bytes: the 16-byte reads of sixteen lanes cover all 64 banks once).  The next batch's loads
    // are in flight under this batch's 128 products (the kernel takes the whole register file: one wave per SIMD).
    const WRsrc ys_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(ys), 0, (unsigned)(Y_ROWS * 4ll * npad), 0x00020000);
    const WRsrc xs_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xs), 0, (unsigned)(X_ROWS * 4ll * npad), 0x00020000);
    const unsigned row4 = (unsigned)npad * 4u;
    const unsigned piece = 16u * (unsigned)(lane & 7) + 4u * (unsigned)s0;
    const unsigned va = (unsigned)(T.y_row + (lane >> 3)) * row4 + piece, vb = (unsigned)(T.x_row + (lane >> 3)) * row4 + piece;
    constexpr int ROW_F4 = 9, TA = 8 * MG / 2, TB = 8 * NG / 2; // float4 per LDS row (8 + 1 pad); load instructions of the A / B part (4 per tile)
    extern __shared__ float4 s_rows[];
    float4* const mine = s_rows + (threadIdx.x >> 6) * ((32 * (MG + NG)) * ROW_F4);
    float4* const put = mine + (lane >> 3) * ROW_F4 + (lane & 7);   // + 8 t rows
    const float4* const get = mine + i * ROW_F4 + 4 * kh;         // + 32 tile rows, + q
    u32x4 g0[TA + TB], g1[TA + TB]; // two batches of loads in flight (a batch of products lasts ~3.4 us: one batch ahead did not always cover the latency)
    auto fetch = [&](u32x4 (&g)[TA + TB], unsigned sbyte) {
#pragma unroll
        for (int t = 0; t < TA; ++t) if (t / 4 < mt_n) g[t] = __builtin_amdgcn_raw_buffer_load_b128(ys_rs, va, (unsigned)(8 * t) * row4 + sbyte, 0);
#pragma unroll
        for (int t = 0; t < TB; ++t) if (t / 4 < nt_n) g[TA + t] = __builtin_amdgcn_raw_buffer_load_b128(xs_rs, vb, (unsigned)(8 * t) * row4 + sbyte, 0);
    };
    auto stage = [&](const u32x4 (&g)[TA + TB]) { // registers -> LDS rows
#pragma unroll
        for (int t = 0; t < TA; ++t)
            if (t / 4 < mt_n) put[(8 * t) * ROW_F4] = make_float4(__uint_as_float(g[t].x), __uint_as_float(g[t].y), __uint_as_float(g[t].z), __uint_as_float(g[t].w));
#pragma unroll
        for (int t = 0; t < TB; ++t)
            if (t / 4 < nt_n) put[(32 * MG + 8 * t) * ROW_F4] = make_float4(__uint_as_float(g[TA + t].x), __uint_as_float(g[TA + t].y), __uint_as_float(g[TA + t].z), __uint_as_float(g[TA + t].w));
    };
    const float4 zero4 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    auto products = [&]() {
        float4 a[MG][4], b[NG][4];
#pragma unroll
        for (int m = 0; m < MG; ++m)
#pragma unroll
            for (int q = 0; q < 4; ++q) a[m][q] = (m < mt_n && av[m]) ? get[(32 * m) * ROW_F4 + q] : zero4;
#pragma unroll
        for (int n = 0; n < NG; ++n)
#pragma unroll
            for (int q = 0; q < 4; ++q) b[n][q] = (n < nt_n && bv[n]) ? get[(32 * (MG + n)) * ROW_F4 + q] : zero4;
        // (one accumulator's sixteen products in a row; k-steps outermost -- eight independent chains -- measured slower: 0.446 against 0.419 ms)
#pragma unroll
        for (int m = 0; m < MG; ++m) {
            if (m >= mt_n) break;
#pragma unroll
            for (int n = 0; n < NG; ++n) {
                if (n >= nt_n) break;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m][q].x, b[n][q].x, acc[m][n], 0, 0, 0);
                    acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m][q].y, b[n][q].y, acc[m][n], 0, 0, 0);
                    acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m][q].z, b[n][q].z, acc[m][n], 0, 0, 0);
                    acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m][q].w, b[n][q].w, acc[m][n], 0, 0, 0);
                }
            }
        }
    };
#ifdef VANERF_EXP_WP_NOLOAD // timing experiments, wrong results: 1 = no global loads and no staging (products on whatever the LDS holds), 2 = no products
#define VANERF_WP_FETCH(g, s) ((void)0)
#define VANERF_WP_STAGE(g) ((void)0)
#else
#define VANERF_WP_FETCH(g, s) fetch(g, s)
#define VANERF_WP_STAGE(g) stage(g)
#endif
#ifdef VANERF_EXP_WP_NOMFMA
#define VANERF_WP_PRODUCTS() ((void)0)
#else
#define VANERF_WP_PRODUCTS() products()
#endif
    VANERF_WP_FETCH(g0, 0);
    if (32 < per) VANERF_WP_FETCH(g1, 4 * 32);
    for (long long s = 0; s < per; s += 64) {
        VANERF_WP_STAGE(g0);                       // (LDS operations of a wave execute in order: the previous batch's reads are ahead of these writes)
        if (s + 64 < per) VANERF_WP_FETCH(g0, (unsigned)(4 * (s + 64)));
        VANERF_WP_PRODUCTS();
        if (s + 32 >= per) break;
        VANERF_WP_STAGE(g1);
        if (s + 96 < per) VANERF_WP_FETCH(g1, (unsigned)(4 * (s + 96)));
        VANERF_WP_PRODUCTS();
    }
    // accumulate: register r of lane l holds row (r & 3) + 8 (r >> 2) + 4 (l >> 5), column l & 31 of the tile
    float* out = dw + (size_t)slice * T.dw_layer + T.dw_off; // dw[slice][all layers]: T.dw_layer = floats per slice, T.dw_off = the layer's offset in it
#pragma unroll
    for (int m = 0; m < MG; ++m) {
        if (m >= mt_n) break;
#pragma unroll
        for (int n = 0; n < NG; ++n) {
            if (n >= nt_n) break;
            const int c = 32 * n + i;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = 32 * m + (r & 3) + 8 * (r >> 2) + 4 * kh;
                if (row < T.n_out && c < T.n_slot) {
                    float* q = out + (size_t)(T.out0 + row) * T.ld + T.slot0 + c;
                    *q += acc[m][n][r];
                }
            }
        }
    }
}

struct TaskTable {
    std::vector<WpTask> host;
    unsigned total = 0; // floats per slice of the whole accumulator
};
const TaskTable& task_table()
{
    static const TaskTable t = [] {
        TaskTable tt;
        unsigned off = 0;
        struct Lay { unsigned off; int n_out, n_slot; };
        std::vector<Lay> lays;
        for (int l = 0; l < NUM_LAYERS; ++l) {
            lays.push_back({off, kNOUT[l], 2 * kT[l]});
            off += (unsigned)(kNOUT[l] * 2 * kT[l]);
        }
        tt.total = off;
        const unsigned total = off;
        for (int l = 0; l < NUM_LAYERS; ++l)
            for (int o = 0; o < kNOUT[l]; o += 32 * MG)
                for (int c = 0; c < 2 * kT[l]; c += 32 * NG) {
                    WpTask t{};
                    t.y_row = y_row_base(l) + o; t.x_row = x_row_base(l) + c;
                    t.n_out = kNOUT[l] - o; t.n_slot = 2 * kT[l] - c; t.ld = 2 * kT[l];
                    t.dw_off = 0; t.dw_layer = 0; t.out0 = o; t.slot0 = c;
                    t.x_row = x_row_base(l) + c;
                    t.dw_off = lays[l].off;
                    t.dw_layer = total;
                    tt.host.push_back(t);
                }
        // heavy groups first: the four waves of a block then carry similar work, and the launch ends on the light ones
        auto weight = [](const WpTask& t) { return ((std::min(t.n_out, 32 * MG) + 31) / 32) * ((std::min(t.n_slot, 32 * NG) + 31) / 32); };
        std::stable_sort(tt.host.begin(), tt.host.end(), [&](const WpTask& a, const WpTask& b) { return weight(a) > weight(b); });
        return tt;
    }();
    return t;
}

// per (device, slices): the table with the accumulator offsets of that slicing
struct DevTable { WpTask* dev = nullptr; int n = 0; };
const DevTable& device_table(int device, int slices)
{
    static std::mutex mu;
    static std::vector<std::pair<std::pair<int, int>, DevTable>> cache;
    std::lock_guard<std::mutex> lock(mu);
    for (auto& e : cache)
        if (e.first == std::make_pair(device, slices)) return e.second;
    std::vector<WpTask> h = task_table().host;
    DevTable d;
    d.n = (int)h.size();
    HIP_CHECK(hipMalloc(&d.dev, h.size() * sizeof(WpTask)));
    HIP_CHECK(hipMemcpy(d.dev, h.data(), h.size() * sizeof(WpTask), hipMemcpyHostToDevice));
    cache.emplace_back(std::make_pair(device, slices), d);
    return cache.back().second;
}

} // namespace

// dw: [layout_slices][sum over the layers of n_out x n_slots] floats (slice-major: one sum over the first dimension closes a step), accumulated into.
// slices <= layout_slices: a block too short to be cut `layout_slices` times accumulates into the first `slices` parts of the same accumulator.
extern "C" int vanerf_weight_products(const float* xs, const float* ys, int64_t npad, int slices, int layout_slices, float* dw, void* stream)
{
    return guarded([&] {
        if (!xs || !ys || !dw) throw_error("vanerf_weight_products: null argument");
        if (slices <= 0 || slices > layout_slices || npad <= 0 || npad % ((int64_t)slices * 32) != 0)
            throw_error("vanerf_weight_products: npad = %lld must be a multiple of 32 x slices = %d, slices <= layout_slices = %d", (long long)npad, 32 * slices, layout_slices);
        int device = 0;
        HIP_CHECK(hipGetDevice(&device));
        const DevTable& t = device_table(device, layout_slices);
        const unsigned blocks = (unsigned)((t.n + 3) / 4) * (unsigned)slices;
        constexpr int lds = 4 * 32 * (MG + NG) * 9 * 16; // four waves x 192 rows x 144 bytes
        static_assert(lds <= 160 * 1024, "the four waves' windows fit the CU's LDS");
        if ((long long)X_ROWS * 4 * npad >= (1ll << 32)) throw_error("vanerf_weight_products: spills of more than 4 GB (32-bit offsets)");
        HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(weight_products_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        hipLaunchKernelGGL(weight_products_kernel, dim3(blocks), dim3(256), lds, (hipStream_t)stream, t.dev, t.n, xs, ys, (long long)npad, slices, dw);
        HIP_CHECK(hipGetLastError());
    });
}

// floats per slice of the accumulator (= the sum over the layers of n_out x n_slots)
extern "C" int vanerf_weight_products_size(int64_t* floats_per_slice)
{
    return guarded([&] {
        if (!floats_per_slice) throw_error("vanerf_weight_products_size: null argument");
        *floats_per_slice = (int64_t)task_table().total;
    });
}
