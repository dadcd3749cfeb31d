// common.h -- error plumbing shared by the translation units of libvanerf_hip.so.
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/vanerf_hip.h"
#include "layer_spec.h"

namespace vanerf {

struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};

[[noreturn]] inline void throw_error(const char* fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    throw Error(VANERF_EINVAL, buf);
}

inline void hip_check(hipError_t e, const char* what)
{
    if (e != hipSuccess) {
        char buf[512];
        snprintf(buf, sizeof buf, "%s: %s", what, hipGetErrorString(e));
        throw Error(VANERF_EHIP, buf);
    }
}
#define HIP_CHECK(x) ::vanerf::hip_check((x), #x)

void set_last_error(const char* msg);

// Runs `fn`, converts C++ exceptions into the C ABI's negative return codes (never throws across the ABI).
template <class F>
int guarded(F&& fn) noexcept
{
    try {
        fn();
        return VANERF_OK;
    } catch (const Error& e) {
        set_last_error(e.what());
        return e.code;
    } catch (const std::bad_alloc&) {
        set_last_error("out of host memory");
        return VANERF_ENOMEM;
    } catch (const std::exception& e) {
        set_last_error(e.what());
        return VANERF_EINVAL;
    } catch (...) {
        set_last_error("unknown error");
        return VANERF_EINVAL;
    }
}

void pack_weights_host(const VanerfWeightTable& w, std::vector<float>& out, LayerOffsets& offs, int mode = 0, std::vector<float>* bwd = nullptr);
// The packer's two stages (weights_pack.cpp): where a layer's parameters are and where its effective weights go in the flat array `eff` ...
struct LayerSrc {
    const float* w; // [nout][kin] (weight-normed layers: weight_v)
    const float* g; // weight_g or null
    const float* b; // bias or null
    int nout, kin;
    unsigned eff;   // offset of the layer in eff: [nout][kin], then the bias
};
void layer_sources(const VanerfWeightTable& w, LayerSrc out[NUM_LAYERS]);
// ... and the placement of eff in the streams: out[i] = id ? eff[id - 1] : 0.  fwd / bwd: the fp32 forward stream and the transposed stream of
// the fused backward; fwd_b: the bf16x3 stream, two ints per 32-bit word (low half's id | part << 30, high half's id; part 0 = high bf16 part).
struct PackTables {
    std::vector<int> fwd, bwd, fwd_b;
    LayerOffsets offs{};
    unsigned n_eff = 0;
};
const PackTables& pack_tables();
int layer_slots(int layer, int* k_of_slot, int cap);

void composite_with_handle(const VanerfWeights* w, const float* rgba, const float* z, const float* msdf, const float* rgba_b, const float* msdf_b,
                           const int32_t* src, int Sa, int Sb, int R, float* color, float* depth, float* alpha, float* sdf, float* contrib, void* stream);

} // namespace vanerf

struct VanerfWeights {
    float* dev = nullptr;   // packed fragment streams
    float* dev_bwd = nullptr; // fp32 handles only: the transposed streams of the fused backward pass (query_backward.hip)
    size_t n_floats_bwd = 0;
    size_t n_floats = 0;
    vanerf::LayerOffsets offs{};
    int mode = 0;
    float beta = 0.1f;      // sigmoid_beta as packed from the host (clamped); stale once vanerf_weights_update took it from the device ...
    float* dev_beta = nullptr; // ... so the passes' composites read this device copy (one float, clamped)
    float* dev_eff = nullptr;  // vanerf_weights_update's stage 1: the effective weights (allocated at the first update)
    int device = 0;
    unsigned long long* stats = nullptr; // [0]: running count of 32-sample groups that took query_kernel's all-invalid short path
};
