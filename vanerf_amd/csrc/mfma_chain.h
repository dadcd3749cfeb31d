// mfma_chain.h -- the fp32 MFMA layer runner shared by query_kernel.hip (forward, fp32 mode) and query_backward.hip (training):
// a chain of v_mfma_f32_32x32x2_f32 whose A fragments stream from L2 through a register ring (layer_spec.h describes the layout).
#pragma once
#include <utility>

#include "common.h"

namespace vanerf_chain {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// ---------------------------------------------------------------------------------------------
// weight fragment stream: register ring, prefetch distance ~1000 cycles
// ---------------------------------------------------------------------------------------------
template <int NB> struct WFrag { float v[NB]; };

// Address = wave-uniform base (SGPR pair) + per-lane element offset (one VGPR): `global_load ... v_off, s[base] offset:imm`.
// A per-lane 64-bit pointer would cost two VGPRs per 4 KB window of the unrolled stream.
// A fragments are fetched with buffer loads: `buffer_load_dword{,x2,x3,x4} v, v_off, s[rsrc:rsrc+3], s_off offen`.  The 128-bit
// resource descriptor and the per-step byte offset `so` (a compile-time constant -> one s_mov) are scalar, the lane's byte
// offset `vb` (lane * NB * 4) is the only VGPR of address for the whole 640 KB stream: no VALU address arithmetic at all
// (64-bit global addressing cost two v_add_co per 4 KB window here; VALU cycles add to fp32-MFMA cycles on gfx950).
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x3 __attribute__((ext_vector_type(3)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
using WRsrc = __amdgpu_buffer_rsrc_t;

template <int NB> __device__ __forceinline__ WFrag<NB> wload(WRsrc rs, unsigned so, unsigned vb)
{
    WFrag<NB> r;
    if constexpr (NB == 1) {
        r.v[0] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, vb, so, 0));
    } else if constexpr (NB == 2) {
        const u32x2 t = __builtin_amdgcn_raw_buffer_load_b64(rs, vb, so, 0);
        r.v[0] = __uint_as_float(t.x); r.v[1] = __uint_as_float(t.y);
    } else if constexpr (NB == 3) {
        const u32x3 t = __builtin_amdgcn_raw_buffer_load_b96(rs, vb, so, 0);
        r.v[0] = __uint_as_float(t.x); r.v[1] = __uint_as_float(t.y); r.v[2] = __uint_as_float(t.z);
    } else {
        const u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(rs, vb, so, 0);
        r.v[0] = __uint_as_float(t.x); r.v[1] = __uint_as_float(t.y); r.v[2] = __uint_as_float(t.z); r.v[3] = __uint_as_float(t.w);
    }
    return r;
}

template <class F, int... I> __device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>)
{
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F> __device__ __forceinline__ void static_for(F&& f)
{
    static_for_impl(static_cast<F&&>(f), std::make_integer_sequence<int, N>{});
}

// ring depth per block count: D * NB = 15..16 VGPRs, D * NB MFMAs (x 64 cycles) between a load and its use
template <int NB> struct RingDepth { static constexpr int value = NB == 1 ? 16 : NB == 2 ? 8 : NB == 3 ? 5 : 4; };

template <int NB> __device__ __forceinline__ void mfma_step(f32x16 (&acc)[NB], const WFrag<NB>& a, float b)
{
#pragma unroll
    for (int ob = 0; ob < NB; ++ob) acc[ob] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.v[ob], b, acc[ob], 0, 0, 0);
#ifdef VANERF_PIN_KSTEPS // experiment knob: forbid scheduling across k-steps
    __builtin_amdgcn_sched_barrier(0);
#endif
}

// The register ring of one layer's A fragments.  ring_start() issues the first D loads; it is called well before the layer
// runs (ahead of the previous layer's activation epilogue) so that the L2 latency of the ring fill is hidden.
template <int NB> struct Ring { WFrag<NB> f[RingDepth<NB>::value]; };

template <int NB, int T> __device__ __forceinline__ Ring<NB> ring_start(WRsrc rs, unsigned sbase, unsigned voff)
{
    constexpr int D = RingDepth<NB>::value;
    Ring<NB> r;
    static_for<D>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        if constexpr (i < T) r.f[i] = wload<NB>(rs, (sbase + i * 64 * NB) * 4u, voff);
    });
    return r;
}

// Runs the T k-steps of one layer, fully unrolled.  operand(integral_constant<int, t>) returns the B operand (this
// lane's activation) of step t; step t's A fragment lives in ring slot t % D and is re-loaded with step t + D as soon
// as it has been consumed.  `sbase` is the float offset of step 0 in the stream, `voff` = lane * NB * 4 (bytes).
template <int NB, int T, class Op>
__device__ __forceinline__ void run_layer(f32x16 (&acc)[NB], Ring<NB>& ring, WRsrc rs, unsigned sbase, unsigned voff, Op&& operand)
{
    constexpr int D = RingDepth<NB>::value;
    static_for<T>([&](auto tc) {
        constexpr int t = decltype(tc)::value;
        const float b = operand(tc);
        const WFrag<NB> a = ring.f[t % D];
        if constexpr (t + D < T) ring.f[t % D] = wload<NB>(rs, (sbase + (t + D) * 64 * NB) * 4u, voff);
        mfma_step<NB>(acc, a, b);
    });
}


template <int NB> __device__ __forceinline__ void zero(f32x16 (&acc)[NB])
{
#pragma unroll
    for (int ob = 0; ob < NB; ++ob)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[ob][r] = 0.0f;
}

} // namespace vanerf_chain
