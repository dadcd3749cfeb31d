// api.cpp -- ABI bookkeeping: version, thread-local last error, packed-weight handle.
#include <algorithm>
#include <cstddef>

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
        if (e != hipSuccess) {
            if (h->dev) (void)hipFree(h->dev);
            if (h->dev_bwd) (void)hipFree(h->dev_bwd);
            if (h->stats) (void)hipFree(h->stats);
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
