// weights_update.hip -- re-packs a weight handle IN PLACE from parameters that live on the device: the training step's path.
// vanerf_weights_pack copies every parameter to the host, folds and permutes there and uploads (~100 blocking copies, two hipMalloc / hipFree
// pairs: 4.8 ms of host time per training step for the two handles of the step, and a drained GPU at the start of every step).  Here the
// packer's two stages (weights_pack.cpp) run as two launches per stream on the caller's stream, nothing blocks and no address changes:
//   fold_kernel    eff = [every layer's [out][in] matrix with weight-norm folded | its bias]   (same arithmetic, same order as fold_host)
//   place_*_kernel out[i] = eff[id[i] - 1] through the placement tables (uploaded once per device), the bf16x3 stream split into its parts
// Bit-identical to a fresh vanerf_weights_pack of the same parameters (tests/test_hip_parity.py).
#include <hip/hip_runtime.h>

#include <mutex>

#include "common.h"

using namespace vanerf;

namespace {

struct FoldArgs {
    LayerSrc l[NUM_LAYERS];
};

// one block per (row, layer); weight-norm: W[o][:] = v[o][:] * (g[o] / ||v[o][:]||), the norm summed in double in index order by one lane
__global__ __launch_bounds__(64) void fold_kernel(FoldArgs a, float* __restrict__ eff, const float* __restrict__ beta_src, float* __restrict__ beta_dst)
{
    const LayerSrc L = a.l[blockIdx.y];
    const int o = blockIdx.x;
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0 && beta_src) beta_dst[0] = fmaxf(beta_src[0], 2e-3f); // sdf_activation clamp (src/model.py:880)
    if (o >= L.nout) return;
    const float* v = L.w + (size_t)o * L.kin;
    float* e = eff + L.eff + (size_t)o * L.kin;
    __shared__ float s_scale;
    if (L.g) {
        if (threadIdx.x == 0) {
            double sum = 0.0;
            for (int k = 0; k < L.kin; ++k) sum += (double)v[k] * (double)v[k]; // the product of two floats is exact in double: no contraction issue
            s_scale = L.g[o] / (float)sqrt(sum);
        }
        __syncthreads();
        const float scale = s_scale;
        for (int k = threadIdx.x; k < L.kin; k += 64) e[k] = v[k] * scale;
    } else {
        for (int k = threadIdx.x; k < L.kin; k += 64) e[k] = v[k];
    }
    if (L.b && threadIdx.x == 0) eff[L.eff + (size_t)L.nout * L.kin + o] = L.b[o];
}

__global__ void place_f32_kernel(const int* __restrict__ tab, const float* __restrict__ eff, float* __restrict__ out, unsigned n)
{
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int id = tab[i];
    out[i] = id ? eff[id - 1] : 0.0f;
}

__device__ inline unsigned bf16_rne(float f)
{
    const unsigned u = __float_as_uint(f);
    return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}
__device__ inline unsigned bf16_part(float v, int part)
{
    const unsigned hi = bf16_rne(v);
    return (part ? bf16_rne(v - __uint_as_float(hi << 16)) : hi) & 0xffffu;
}

__global__ void place_b_kernel(const int2* __restrict__ tab, const float* __restrict__ eff, unsigned* __restrict__ out, unsigned n)
{
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int2 t = tab[i];
    const int part = (t.x >> 30) & 1, id0 = t.x & 0x3fffffff, id1 = t.y;
    const float v0 = id0 ? eff[id0 - 1] : 0.0f, v1 = id1 ? eff[id1 - 1] : 0.0f;
    out[i] = bf16_part(v0, part) | (bf16_part(v1, part) << 16);
}

// the placement tables on each device, uploaded at the first update there (blocking, once)
struct DeviceTables {
    int* fwd = nullptr;
    int* bwd = nullptr;
    int* fwd_b = nullptr;
};
const DeviceTables& device_tables(int device)
{
    static std::mutex mu;
    static DeviceTables per_device[64];
    if (device < 0 || device >= 64) throw_error("vanerf_weights_update: device %d", device);
    std::lock_guard<std::mutex> lock(mu);
    DeviceTables& t = per_device[device];
    if (!t.fwd) {
        const PackTables& p = pack_tables();
        auto upload = [](const std::vector<int>& v) {
            int* d = nullptr;
            HIP_CHECK(hipMalloc(&d, v.size() * sizeof(int)));
            HIP_CHECK(hipMemcpy(d, v.data(), v.size() * sizeof(int), hipMemcpyHostToDevice));
            return d;
        };
        t.bwd = upload(p.bwd);
        t.fwd_b = upload(p.fwd_b);
        t.fwd = upload(p.fwd);
    }
    return t;
}

} // namespace

extern "C" int vanerf_weights_update(VanerfWeights* h, const VanerfWeightTable* dev, const float* sigmoid_beta_dev, void* stream)
{
    return guarded([&] {
        if (!h || !dev) throw_error("vanerf_weights_update: null argument");
        int device = -1;
        HIP_CHECK(hipGetDevice(&device));
        if (device != h->device) throw_error("vanerf_weights_update: handle packed on device %d, current device %d", h->device, device);
        const PackTables& p = pack_tables();
        if (h->n_floats != (h->mode ? p.fwd_b.size() / 2 : p.fwd.size()) || (h->dev_bwd && h->n_floats_bwd != p.bwd.size()))
            throw_error("internal: handle and placement tables disagree on the stream sizes");
        FoldArgs a;
        layer_sources(*dev, a.l); // checks the pointers the layers need
        const DeviceTables& t = device_tables(device);
        if (!h->dev_eff) HIP_CHECK(hipMalloc(&h->dev_eff, (size_t)p.n_eff * sizeof(float)));
        hipStream_t s = (hipStream_t)stream;
        hipLaunchKernelGGL(fold_kernel, dim3(128, NUM_LAYERS), dim3(64), 0, s, a, h->dev_eff, sigmoid_beta_dev, h->dev_beta);
        const unsigned n = (unsigned)h->n_floats;
        if (h->mode)
            hipLaunchKernelGGL(place_b_kernel, dim3((n + 255) / 256), dim3(256), 0, s, reinterpret_cast<const int2*>(t.fwd_b), h->dev_eff,
                               reinterpret_cast<unsigned*>(h->dev), n);
        else
            hipLaunchKernelGGL(place_f32_kernel, dim3((n + 255) / 256), dim3(256), 0, s, t.fwd, h->dev_eff, h->dev, n);
        if (h->dev_bwd) {
            const unsigned nb = (unsigned)h->n_floats_bwd;
            hipLaunchKernelGGL(place_f32_kernel, dim3((nb + 255) / 256), dim3(256), 0, s, t.bwd, h->dev_eff, h->dev_bwd, nb);
        }
        HIP_CHECK(hipGetLastError());
        if (!sigmoid_beta_dev) { // the table's own value (host float), as vanerf_weights_pack takes it
            h->beta = dev->sigmoid_beta < 2e-3f ? 2e-3f : dev->sigmoid_beta;
            HIP_CHECK(hipMemcpyAsync(h->dev_beta, &h->beta, sizeof(float), hipMemcpyHostToDevice, s));
        }
    });
}
