// api.cpp -- ABI bookkeeping: version, thread-local last error, packed-weight handle.
#include <algorithm>
#include <cstddef>

#include <cstring>

#include "common.h"

namespace vanerf {
static thread_local std::string g_last_error;
void set_last_error(const char* msg) { g_last_error = msg ? msg : ""; }
} // namespace vanerf

using namespace vanerf;

extern "C" int vanerf_abi_version(void) { return VANERF_ABI_VERSION; }

extern "C" const char* vanerf_last_error(void) { return g_last_error.c_str(); }

extern "C" int vanerf_weights_pack(const VanerfWeightTable* w, int mode, VanerfWeights** out)
{
    return guarded([&] {
        if (!w || !out) throw_error("vanerf_weights_pack: null argument");
        if (mode != 0 && mode != 1) throw_error("vanerf_weights_pack: mode %d not supported (0 = fp32 MFMA, 1 = split-bf16 x3)", mode);
        const float* const* ptrs = reinterpret_cast<const float* const*>(w);
        constexpr size_t n_ptrs = offsetof(VanerfWeightTable, sigmoid_beta) / sizeof(const float*);
        for (size_t i = 0; i < n_ptrs; ++i)
            if (!ptrs[i]) throw_error("vanerf_weights_pack: weight pointer #%d is null", (int)i);
        std::vector<float> host, host_bwd;
        LayerOffsets offs{};
        pack_weights_host(*w, host, offs, mode, mode == 0 ? &host_bwd : nullptr);
        auto* h = new VanerfWeights();
        h->n_floats_bwd = host_bwd.size();
        h->n_floats = host.size();
        h->offs = offs;
        h->mode = mode;
        h->beta = w->sigmoid_beta < 2e-3f ? 2e-3f : w->sigmoid_beta; // sdf_activation clamp (src/model.py:880)
        hipError_t e = hipGetDevice(&h->device);
        if (e == hipSuccess) e = hipMalloc(&h->dev, host.size() * sizeof(float));
        if (e == hipSuccess) e = hipMemcpy(h->dev, host.data(), host.size() * sizeof(float), hipMemcpyHostToDevice);
        if (e == hipSuccess && !host_bwd.empty()) e = hipMalloc(&h->dev_bwd, host_bwd.size() * sizeof(float));
        if (e == hipSuccess && !host_bwd.empty()) e = hipMemcpy(h->dev_bwd, host_bwd.data(), host_bwd.size() * sizeof(float), hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMalloc(&h->stats, sizeof(unsigned long long));
        if (e == hipSuccess) e = hipMemset(h->stats, 0, sizeof(unsigned long long));
        if (e == hipSuccess) e = hipMalloc(&h->dev_beta, sizeof(float));
        if (e == hipSuccess) e = hipMemcpy(h->dev_beta, &h->beta, sizeof(float), hipMemcpyHostToDevice);
        if (e != hipSuccess) {
            if (h->dev) (void)hipFree(h->dev);
            if (h->dev_bwd) (void)hipFree(h->dev_bwd);
            if (h->stats) (void)hipFree(h->stats);
            if (h->dev_beta) (void)hipFree(h->dev_beta);
            delete h;
            hip_check(e, "vanerf_weights_pack: device upload");
        }
        *out = h;
    });
}

extern "C" int vanerf_weights_free(VanerfWeights* w)
{
    return guarded([&] {
        if (!w) return;
        if (w->dev) HIP_CHECK(hipFree(w->dev));
        if (w->dev_bwd) HIP_CHECK(hipFree(w->dev_bwd));
        if (w->stats) HIP_CHECK(hipFree(w->stats));
        if (w->dev_beta) HIP_CHECK(hipFree(w->dev_beta));
        if (w->dev_eff) HIP_CHECK(hipFree(w->dev_eff));
        delete w;
    });
}

// Host-only view of the packed stream, used by the CPU tests of the packer (no GPU needed).
extern "C" int vanerf_weights_pack_host(const VanerfWeightTable* w, float* out, int64_t cap, int64_t* n_out, unsigned* offsets)
{
    return guarded([&] {
        if (!w || !n_out) throw_error("vanerf_weights_pack_host: null argument");
        std::vector<float> host;
        LayerOffsets offs{};
        pack_weights_host(*w, host, offs);
        *n_out = (int64_t)host.size();
        if (offsets) std::copy_n(offs.off, (size_t)NUM_LAYERS, offsets);
        if (out) {
            if (cap < (int64_t)host.size()) throw_error("vanerf_weights_pack_host: buffer too small");
            std::copy(host.begin(), host.end(), out);
        }
    });
}

// Host-only view of any of the three streams a handle can carry: which = 0 the fp32 forward stream, 1 the bf16x3 forward stream (32-bit words
// of two bf16 each, returned as raw floats), 2 the transposed fp32 stream of the fused backward pass.
extern "C" int vanerf_weights_stream_host(const VanerfWeightTable* w, int which, float* out, int64_t cap, int64_t* n_out)
{
    return guarded([&] {
        if (!w || !n_out) throw_error("vanerf_weights_stream_host: null argument");
        if (which < 0 || which > 2) throw_error("vanerf_weights_stream_host: which = %d (0 fp32, 1 bf16x3, 2 backward)", which);
        std::vector<float> host, bwd;
        LayerOffsets offs{};
        pack_weights_host(*w, host, offs, which == 1 ? 1 : 0, which == 2 ? &bwd : nullptr);
        const std::vector<float>& src = which == 2 ? bwd : host;
        *n_out = (int64_t)src.size();
        if (out) {
            if (cap < (int64_t)src.size()) throw_error("vanerf_weights_stream_host: buffer too small");
            std::memcpy(out, src.data(), src.size() * sizeof(float)); // raw copy: the bf16x3 words are not numbers
        }
    });
}

// The streams a handle holds on the device, copied back (blocking; tests of vanerf_weights_update): which = 0 the forward stream of the handle's
// mode, 2 the backward stream (fp32 handles).
extern "C" int vanerf_weights_download(const VanerfWeights* w, int which, float* out, int64_t cap, int64_t* n_out)
{
    return guarded([&] {
        if (!w || !n_out) throw_error("vanerf_weights_download: null argument");
        if (which != 0 && which != 2) throw_error("vanerf_weights_download: which = %d (0 forward, 2 backward)", which);
        const float* src = which ? w->dev_bwd : w->dev;
        const size_t n = which ? w->n_floats_bwd : w->n_floats;
        if (!src) throw_error("vanerf_weights_download: the handle carries no such stream");
        *n_out = (int64_t)n;
        if (out) {
            if (cap < (int64_t)n) throw_error("vanerf_weights_download: buffer too small");
            HIP_CHECK(hipDeviceSynchronize());
            HIP_CHECK(hipMemcpy(out, src, n * sizeof(float), hipMemcpyDeviceToHost));
        }
    });
}

// Running count (since the handle was packed) of 32-sample groups for which vanerf_query_samples skipped GeoVisFusion / mlp_geo
// because every sample of the group was invalid.  Synchronises the device; meant for benchmarks and tests, not for the render loop.
extern "C" int vanerf_weights_short_groups(const VanerfWeights* w, uint64_t* count)
{
    return guarded([&] {
        if (!w || !count) throw_error("vanerf_weights_short_groups: null argument");
        unsigned long long v = 0;
        HIP_CHECK(hipMemcpy(&v, w->stats, sizeof v, hipMemcpyDeviceToHost));
        *count = v;
    });
}
