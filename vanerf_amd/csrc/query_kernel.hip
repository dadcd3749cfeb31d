// query_kernel.hip -- the per-sample hot kernel: VANeRF.query + eval_func for N samples
// (reference src/model.py:748-957, 1140-1160; networks src/networks.py:27-33, 75-106, 281-293;
// src/utils.py:136-151, 633-649, 744-779, 822-880; src/spatial.py:59-117).
//
// One wave = 32 samples.  Lane l works on sample j = l & 31; the two lanes j and j + 32 that share
// a sample split every K dimension between them (half h = l >> 5), see layer_spec.h.  All 20 dense
// layers run as MFMA chains -- query_kernel<0>: v_mfma_f32_32x32x2_f32 on fp32 operands; query_kernel<1>:
// v_mfma_f32_32x32x16_bf16 on split (hi + lo) operands, fp32 accumulate -- whose accumulators stay in registers
// from the bilinear gathers to the final (alpha, sdf, rgb) store: activations never touch LDS or HBM.
// No LDS at all: key points come through the scalar cache (fp32 kernel) or live in registers (bf16 kernel); the 1-NN
// vertex index arrives from vanerf_mesh_query_accel.  Weights stream from L2 as pre-permuted MFMA A-fragments (weights_pack.cpp).
//
// Built with -ffp-contract=off: the integer-valued outputs (1-NN index) depend on fp32 compare
// results and must match oracle/mesh_oracle.c bit for bit; fused multiply-adds are spelled fmaf().
#include <cstdlib>
#include <utility>

#include "common.h"
#include "mfma_chain.h"

using namespace vanerf;
using namespace vanerf_chain;


namespace {

// Waves per SIMD of the split-bf16 kernel: TWO (round 3), as one 8-wave block per CU that owns the CU's LDS.  One in-order wave per SIMD
// issues at most one instruction every ~4 cycles, and a full 32-sample group is ~9 700 instructions around 990 MFMAs: the wave was
// issue-bound at ~54 k cycles per group with NO fragment traffic at all (timing experiment VANERF_EXP_FRAG_SRC=2), against 31.7 k cycles of
// matrix-pipe time.  A second wave fills the issue slots; it needs the kernel in 256 registers WITHOUT scratch (build.py fails a build
// that spills): key points in LDS instead of 63 VGPRs, gathers issued where they are used, fragment rings one k-step deep and kept per
// output block, LDS addresses formed from three window bases (see LAddr).  Round 1 had tried two waves with the 476-register kernel cut
// to 256 by the compiler (155 VGPRs in scratch): results were not reproducible run to run; the scratch-free
// build is bit-identical to the one-wave build and reproducible (tools/check_mode1_determinism.py).  VANERF_WAVES_PER_SIMD_B=1
// VANERF_WPB_B=4 builds the round-2 configuration (one 4-wave block per CU, 512 registers per wave).
#ifndef VANERF_WAVES_PER_SIMD_B
#define VANERF_WAVES_PER_SIMD_B 2
#endif
#ifndef VANERF_WAVES_PER_SIMD
#define VANERF_WAVES_PER_SIMD 2
#endif
#ifndef VANERF_WPB_B
#define VANERF_WPB_B (VANERF_WAVES_PER_SIMD_B == 2 ? 8 : 4)
#endif
// waves per block: the fp32 kernel runs two 4-wave blocks per CU, the split-bf16 kernel ONE block per CU (it owns the CU's LDS)
template <int MODE> constexpr int WPB = MODE == 1 ? VANERF_WPB_B : 4;

// Diagnostic build only (-DVANERF_STAMPS): per-phase s_memtime deltas summed per wave into QueryParams::stamps.
// No stamp executes in the product build; stamp values never reach an output element.

#ifdef VANERF_STAMPS
#define STAMP(k)                                                                                  \
    do {                                                                                          \
        unsigned long long t_;                                                                    \
        __builtin_amdgcn_sched_barrier(0);                                                        \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");               \
        __builtin_amdgcn_sched_barrier(0);                                                        \
        phase_cycles[k] += t_ - t_prev;                                                           \
        t_prev = t_;                                                                              \
    } while (0)
constexpr int N_PHASES = 12;
#else
#define STAMP(k) do { } while (0)
#endif

struct QueryParams {
    VanerfFrame f;
    const float* w;
    unsigned wbytes;
    const float* pts;
    const float* qsdf;
    const uint8_t* qvis;
    const float* noise;
    const int32_t* knn_in;
    const int32_t* order;  // optional: slot k of the launch works on sample order[k] (validity partition), else on sample k
    int raw; // 1: write [sdf_pred, rad, r, g, b] (VANeRF.query), 0: eval_func applied ([alpha, sdf, r, g, b])
    long long n;
    float* out;
    uint8_t* valid;
    unsigned long long* stamps; // [waves][N_PHASES] (diagnostic build), else unused
    unsigned* queue;            // work-queue head: next unclaimed 32-sample group (zero before the launch)
    unsigned long long* short_groups; // optional: += number of groups that took the all-invalid short path
    float wm1[4], hm1[4];             // (float)(W - 1), (float)(H - 1) of the image / tex / geo0 / geo1 maps (kept scalar: no per-lane copies)
    float *xs, *aux;                  // spill mode (training): X and auxiliary spills, [rows][npad] floats (layer_spec.h)
    long long npad;
};

// ---------------------------------------------------------------------------------------------
// split-bf16 ("bf16x3") variant of the layer runner: W = W_hi + W_lo, X = X_hi + X_lo (bf16 each),
// acc += W_hi X_hi + W_hi X_lo + W_lo X_hi on v_mfma_f32_32x32x16_bf16 (fp32 accumulate).  One bf16 k-step covers 8 consecutive
// k-pairs of the fp32 ordering, so the accumulator -> operand chaining of layer_spec.h is unchanged.  Measured against the fp32
// oracle the dropped W_lo X_lo term and the 16-bit operands cost 1.2e-5 absolute on the outputs (the fp32-MFMA path: 9e-6).
// ---------------------------------------------------------------------------------------------
// ring depths (steps in flight) for layers with 4/3, 1 and 2 output blocks.  With the pinned software pipeline of run_layer_b, measured in
// one session: (2,4,2) 8.16-8.23 ms, (2,3,2) 8.22, (2,2,2) 8.24, (1,4,2) 8.25, (3,6,3) 8.32, (1,2,1) 8.33, (4,8,4) +4 %, (5,8,4) +8 %:
// two k-steps ahead cover the L2 latency, deeper rings only add loads in flight
// experiment knobs: bit L set = layer L runs with one product (W_hi X_hi) / two products (W_hi X_hi + W_lo X_hi: activations rounded to bf16)
#ifndef VANERF_P1_MASK
#define VANERF_P1_MASK 0
#endif
#ifndef VANERF_P2_MASK
#define VANERF_P2_MASK 0
#endif
#ifndef VANERF_CHUNKS
#define VANERF_CHUNKS 4
#endif
// ring depth in BLOCK fragments (one (hi, lo) pair = 8 registers, 2 KB of stream) for layers with 4, 3, 2 and 1 output blocks.
// One wave per SIMD, fragments from L2: two k-steps ahead (4 for the one-block layers) cover the latency (measured in steps, one session:
// (2,4,2) 8.16-8.23 ms, (2,2,2) 8.24, (1,4,2) 8.25, (3,6,3) 8.32, (4,8,4) +4 %).  Two waves per SIMD, fragments from the LDS ring: two blocks
// ahead (one for the one-block layers) -- 4 / 3 / 2 / 2 blocks measured the same 5.42 ms with 15 more registers, 6 / 4 spill.
#ifndef VANERF_DB4
#define VANERF_DB4 (VANERF_WAVES_PER_SIMD_B == 2 ? 2 : 8)
#endif
#ifndef VANERF_DB3
#define VANERF_DB3 (VANERF_WAVES_PER_SIMD_B == 2 ? 2 : 6)
#endif
#ifndef VANERF_DB2
#define VANERF_DB2 (VANERF_WAVES_PER_SIMD_B == 2 ? 2 : 4)
#endif
#ifndef VANERF_DB1
#define VANERF_DB1 (VANERF_WAVES_PER_SIMD_B == 2 ? 1 : 4)
#endif
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

struct FragB { u32x4 hi, lo; }; // A fragments (hi and lo parts) of one 32-row output block of one bf16 k-step

// Fragments of the LAST layers of the stream stay resident in LDS for the whole launch (split-bf16 kernel: one block per CU, persistent):
// every group needs them -- the all-invalid groups need nothing else -- and the CU's vector-memory path (L1: 64 B per clock) is what the
// fragment stream loads most (72 % of its cycles, profiles/r02_a_*): ds_read_b128 has four times that width and a fifth of the latency.
// Layers VANERF_LDS_FIRST .. NUM_LAYERS-1 are copied once per block: head1, head2, ibr_compress and the four TexVisFusion layers = 156 KB.
// Two-wave build: the layers in front of the resident ones are SHARED through an LDS ring (below), so only ibr_compress and the four
// TexVisFusion layers (126 KB: what the all-invalid groups need) stay resident and 32 KB go to the ring.
#ifndef VANERF_RING
#define VANERF_RING (VANERF_WAVES_PER_SIMD_B == 2 ? 1 : 0)
#endif
#ifndef VANERF_LDS_FIRST
#define VANERF_LDS_FIRST (VANERF_RING ? 15 /* L_IBR */ : 13 /* L_HEAD1 */)
#endif
constexpr int LDS_FIRST = VANERF_LDS_FIRST;
constexpr unsigned RES_BASE_DW = layer_offset_b(LDS_FIRST), RES_DW = layer_offset_b(NUM_LAYERS) - layer_offset_b(LDS_FIRST);
// Two waves per SIMD (VANERF_WAVES_PER_SIMD_B == 2: 8-wave blocks, 256 registers per wave): nothing long-lived may sit in registers -- the
// lane half's key points live in LDS behind the resident fragments (one ds_read_b128 per key point and group), gathers are issued right
// before their use (the partner wave hides their latency), rings are one step deep.
constexpr bool W2 = VANERF_WAVES_PER_SIMD_B == 2;
// LDS map of the split-bf16 kernel (all of it in the dynamic region, which then starts at LDS address 0: no static __shared__ in this kernel):
//   [0, 672)      the 42 key points as float4 (two-wave build)            [672, 768)  control words: s_base[2], s_valid[2][waves per block]
//   [1024, 1024 + 32 KB)     fragment ring (two-wave build)               [LDS_RES, LDS_RES + RES_DW * 4)  resident fragments
constexpr bool RING = VANERF_RING != 0;
constexpr unsigned RING_PHASE_PIECES = 16u, RING_BYTES = RING ? 2u * RING_PHASE_PIECES * 1024u : 0u; // two halves of one 16 KB phase each
constexpr unsigned LDS_KPT = 0u, LDS_CTRL = 2u * PE_KPT_PER_HALF * 16u, LDS_LAT0 = 768u, LDS_RING = 1024u, LDS_RES = RING ? LDS_RING + RING_BYTES : 1024u;
// [768, 896): ibr_compress of an all-zero pooled latent (its bias through the same MFMA chain), [lane half][16 registers]: what every sample of an
// all-invalid group gets from that layer -- computed once per block, read by the short path instead of 27 MFMAs per group
constexpr unsigned DYN_LDS_BYTES = LDS_RES + RES_DW * 4u;
// The streamed part of the fragment stream (layers 0 .. LDS_FIRST-1) as 1 KB pieces (one (k-step, output block, hi | lo) fragment each = one
// LDS-DMA wave instruction), in the order the layers consume them; a PHASE is 16 consecutive pieces.
constexpr unsigned RING_PIECES = RES_BASE_DW / 256u, RING_PHASES = ((RING_PIECES + RING_PHASE_PIECES - 1u) / RING_PHASE_PIECES + 1u) / 2u * 2u;
static_assert(!RING || VANERF_WPB_B == 8, "the ring's DMA schedule deals 2 pieces of a phase to each of 8 waves");
static_assert(!RING || RING_PHASES * RING_PHASE_PIECES * 256u <= layer_offset_b(NUM_LAYERS), "the last phase's DMA must stay inside the stream");
static_assert(LDS_CTRL + 8u + 2u * 4u * VANERF_WPB_B <= LDS_LAT0 && LDS_LAT0 + 128u <= LDS_RING, "control words / constant latent overlap their neighbours");
static_assert(DYN_LDS_BYTES <= 160u * 1024u, "key points + control words + resident fragments must fit the CU's 160 KB");
extern __shared__ __attribute__((aligned(16))) u32x4 s_dyn[];
typedef __attribute__((address_space(3))) u32x4 lds_u32x4_t;
typedef __attribute__((address_space(3))) unsigned lds_u32_t;
// A ds instruction addresses base VGPR + 16-bit immediate.  Left to itself the compiler forms one base register (lane * 16 + constant) per
// access beyond the first 64 KB, hoists all of them out of the sample loop and spills them; so the three 64 KB windows get one base each,
// re-made opaque at the top of every round (LAddr), and every access names its window and its offset inside it.
struct LAddr {
    unsigned w[3];
    unsigned dma_lds, dma_src;  // (ring build) the wave's LDS-DMA destination / stream offset inside a phase, wave-uniform
};
__device__ __forceinline__ LAddr make_laddr(int lane, unsigned wv_uniform = 0u)
{
    LAddr a;
    a.dma_lds = LDS_RING + wv_uniform * 2048u; a.dma_src = wv_uniform * 2048u;
    a.w[0] = (unsigned)lane * 16u;
    asm volatile("" : "+v"(a.w[0]));
    a.w[1] = a.w[0] + 0x10000u; a.w[2] = a.w[0] + 0x20000u;
    asm volatile("" : "+v"(a.w[1]), "+v"(a.w[2]));
    return a;
}
template <unsigned BYTE> __device__ __forceinline__ u32x4 lds_frag(const LAddr& a) // 16 bytes at LDS address BYTE + lane * 16
{
    return *reinterpret_cast<const lds_u32x4_t*>((size_t)(a.w[BYTE >> 16] + (BYTE & 0xffffu)));
}
__device__ __forceinline__ lds_u32_t* lds_ctrl() { return reinterpret_cast<lds_u32_t*>((size_t)LDS_CTRL); }
template <int NB> struct RingDepthB { static constexpr int value = NB == 1 ? VANERF_DB1 : NB == 2 ? VANERF_DB2 : NB == 3 ? VANERF_DB3 : VANERF_DB4; };
template <int NB> struct RingB { FragB b[RingDepthB<NB>::value]; };

// ---- the fragment ring (two-wave build) ---------------------------------------------------------------------------------------------
// The eight waves of the block walk the same fragment stream one round at a time.  Fetched by every wave for itself, the stream is 8 x 516 KB
// per round through the CU's 64 B / clock vector-memory path: 65 % busy, the largest share of what the waves wait for (profiles/r03_a_*).
// Instead phase P (16 KB) is brought into one half of a 32 KB LDS ring ONCE -- every wave issues two `buffer_load_dwordx4 ... lds` (LDS-DMA:
// 1 KB of memory to M0 + 16 * lane, no registers) -- while the waves read phase P - 1 from the other half with ds_read_b128.  Protocol, per
// wave, at the first read of phase P (phase_begin<P>; everything is unrolled, so P is a compile-time constant at every read):
//     s_waitcnt vmcnt(0) lgkmcnt(0)   my two pieces of phase P have landed (they were issued a phase ago); my reads of phase P - 1 are back
//     s_barrier                        ... and so have / are everybody's: phase P may be read, the half of phase P - 1 may be overwritten
//     issue the DMA of phase P + 1     into the half phase P - 1 occupied
// The last phase issues phase 0 of the NEXT full round, the top-of-round barrier (every wave has passed a vmcnt(0) before its stores) stands in
// for phase_begin<0>, and rounds that take the all-invalid path touch neither half: half 0 holds phase 0 whenever a round starts.
// vmcnt(0) is exact here: in the two-wave build gathers are issued where they are consumed, nothing else is in flight inside the layer stack.
template <unsigned P> __device__ __forceinline__ void ring_dma(WRsrc rs, const LAddr& la) // this wave's two pieces of phase P -> half P % 2
{
    constexpr unsigned lds_off = (P % 2u) * RING_PHASE_PIECES * 1024u, src_off = P * RING_PHASE_PIECES * 1024u;
#if defined(VANERF_EXP_RING) && (VANERF_EXP_RING == 5 || VANERF_EXP_RING == 7)
    unsigned soff5;
    asm volatile("s_add_u32 m0, %1, %3\n\ts_add_u32 %0, %2, %4\n\ts_nop 0\n\t"
                 "buffer_load_dwordx4 %5, %6, %0 offen lds\n\tbuffer_load_dwordx4 %5, %6, %0 offen offset:1024 lds"
                 : "=&s"(soff5)
                 : "s"(la.dma_lds), "s"(la.dma_src), "i"(lds_off), "i"(src_off), "v"(la.w[0]), "s"(rs)
                 : "memory");
    return;
#endif
    unsigned keep, soff;
    // M0 = LDS destination (written and used in ONE statement: hipcc owns M0 everywhere else); one wait state between its write and the DMA
    asm volatile("s_mov_b32 %0, m0\n\ts_add_u32 m0, %2, %4\n\ts_add_u32 %1, %3, %5\n\ts_nop 0\n\t"
                 "buffer_load_dwordx4 %6, %7, %1 offen lds\n\tbuffer_load_dwordx4 %6, %7, %1 offen offset:1024 lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep), "=&s"(soff)
                 : "s"(la.dma_lds), "s"(la.dma_src), "i"(lds_off), "i"(src_off), "v"(la.w[0]), "s"(rs)
                 : "memory");
}
template <unsigned P> __device__ __forceinline__ void phase_begin(WRsrc rs, const LAddr& la)
{
#if defined(VANERF_EXP_RING) && VANERF_EXP_RING == 1 // timing experiments (results may be wrong): 1 no lgkmcnt wait, 2 no barrier, 3 no DMA, 4 nothing
    if constexpr (P > 0) asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    ring_dma<(P + 1u) % RING_PHASES>(rs, la);
#elif defined(VANERF_EXP_RING) && VANERF_EXP_RING == 2
    if constexpr (P > 0) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    ring_dma<(P + 1u) % RING_PHASES>(rs, la);
#elif defined(VANERF_EXP_RING) && VANERF_EXP_RING == 3
    if constexpr (P > 0) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
#elif defined(VANERF_EXP_RING) && VANERF_EXP_RING == 4
#elif defined(VANERF_EXP_RING) && VANERF_EXP_RING == 7
    if constexpr (P > 0) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    ring_dma<(P + 1u) % RING_PHASES>(rs, la);
#else
    if constexpr (P > 0) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    ring_dma<(P + 1u) % RING_PHASES>(rs, la);
#endif
}

// DW: dword offset of the block's hi part in the stream (the lo part follows 1 KB later); the lane's 16 bytes sit at lane * 16
template <bool RES, unsigned DW> __device__ __forceinline__ FragB wload_blk(WRsrc rs, const LAddr& la)
{
    FragB r;
    if constexpr (RING && !RES) { // streamed layer, through the ring
        constexpr unsigned q = DW / 256u; // piece index of the hi part
        static_assert(q % 2u == 0 && q + 1u < RING_PHASES * RING_PHASE_PIECES, "block fragments are two consecutive pieces inside the streamed part");
        if constexpr (q % RING_PHASE_PIECES == 0) phase_begin<q / RING_PHASE_PIECES>(rs, la);
        r.hi = lds_frag<LDS_RING + (q % (2u * RING_PHASE_PIECES)) * 1024u>(la);
        r.lo = lds_frag<LDS_RING + ((q + 1u) % (2u * RING_PHASE_PIECES)) * 1024u>(la);
        return r;
    }
    if constexpr (RES) { // resident layer: the same fragments from LDS
        r.hi = lds_frag<LDS_RES + (DW - RES_BASE_DW) * 4u>(la);
        r.lo = lds_frag<LDS_RES + (DW - RES_BASE_DW + 256u) * 4u>(la);
        return r;
    }
#if defined(VANERF_EXP_FRAG_SRC) && VANERF_EXP_FRAG_SRC == 1 // timing experiment, wrong results: every fragment from LDS (wrapped into the resident region)
    r.hi = lds_frag<LDS_RES + (DW % RES_DW) * 4u>(la);
    r.lo = lds_frag<LDS_RES + ((DW + 256u) % RES_DW) * 4u>(la);
    return r;
#elif defined(VANERF_EXP_FRAG_SRC) && VANERF_EXP_FRAG_SRC == 2 // timing experiment, wrong results: no fragment traffic at all (stale registers)
    {
        const unsigned vb = la.w[0];
        u32x4 a = {vb, vb + 1u, vb + 2u, vb + 3u}, b = {vb + 4u, vb + 5u, vb + 6u, vb + 7u};
        asm volatile("" : "+v"(a), "+v"(b));
        r.hi = a; r.lo = b;
        return r;
    }
#endif
    r.hi = __builtin_amdgcn_raw_buffer_load_b128(rs, la.w[0], DW * 4u, 0);
    r.lo = __builtin_amdgcn_raw_buffer_load_b128(rs, la.w[0], (DW + 256u) * 4u, 0);
    return r;
}

// Block fragment i of a layer = (k-step i / NB, output block i % NB) lies i * 512 dwords into the layer's stream.
template <int NB, int T, bool RES, unsigned SBASE_DW> __device__ __forceinline__ RingB<NB> ring_start_b(WRsrc rs, const LAddr& la)
{
    constexpr int D = RingDepthB<NB>::value, S = (T + 7) / 8;
    RingB<NB> r;
    static_for<D>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        if constexpr (i < S * NB) r.b[i] = wload_blk<RES, SBASE_DW + i * 512u>(rs, la);
    });
    __builtin_amdgcn_sched_barrier(0x000F); // keep the ring fill where it is written: ahead of the previous layer's epilogue
    return r;
}

__device__ __forceinline__ unsigned pack_bf16(float a, float b)
{
    const bf16x2 v = {(__bf16)a, (__bf16)b}; // v_cvt_pk_bf16_f32 (round to nearest even)
    return __builtin_bit_cast(unsigned, v);
}

struct NoPre { template <class C> __device__ __forceinline__ void operator()(C) const {} };

// pre(integral_constant<int, s>) runs before the operands of bf16 step s are gathered (lazy producers, e.g. the positional encoding).
//
// Software pipeline, written out and pinned.  One wave per SIMD issues in order: a bf16 32x32x16 MFMA costs the wave ~16 issue cycles
// and then runs 34 cycles beside whatever the wave issues next, so up to ~4 VALU instructions per MFMA are free -- IF they stand
// between the MFMAs (tools/probe_bf16_valu.hip).  Left to itself the machine scheduler emits the 3*NB MFMAs of a k-step back to
// back (the wave sits in MFMA issue) and then the operand split of the next step (the matrix pipe idles), and it sinks every ring
// re-load to just above its use (no prefetch).  An ablation priced that at 5 ms of MFMA time added to 4.4 ms of everything else.
// So: step s's MFMAs are cut into four chunks, after each chunk comes one pair-split of step s+1's operands (5 VALU), the re-load of
// the consumed ring slot is issued before the first chunk, and a sched_barrier(0) after every chunk keeps that order.
// (sched_group_barrier could request the same interleave, but its solver did not finish on this 10 k-instruction block in 15 min.)
template <int NB, int T, bool RES, unsigned SBASE_DW, int PRODS = 3, class Op, class Pre = NoPre>
__device__ __forceinline__ void run_layer_b(f32x16 (&acc)[NB], RingB<NB>& ring, WRsrc rs, const LAddr& la, Op&& operand,
                                            Pre&& pre = Pre{})
{
    constexpr int D = RingDepthB<NB>::value, S = (T + 7) / 8;
    constexpr int NCH = VANERF_CHUNKS; // chunks a step's MFMAs are cut into (4 / NCH operand pairs of the next step are split after each)
    // hi = bf16(x) (round to nearest even), lo = bf16(x - hi) for operand pair i of step s: 6 VALU -- v_cvt_pk, v_lshlrev, v_and, two
    // v_sub_f32, v_cvt_pk (no packed f32 math: a v_pk_add_f32 beside MFMAs costs more than the two scalar ops it replaces, and the file
    // is built with -fno-slp-vectorize for the same reason).  The empty asm keeps the packed hi opaque: without it the compiler
    // re-converts each element on its own to feed the subtraction (7 per pair).
    auto split_pair = [&](auto sc, auto ic, u32x4& bh, u32x4& bl) {
        constexpr int s = decltype(sc)::value, i = decltype(ic)::value, t0 = 8 * s + 2 * i;
        float x0 = 0.0f, x1 = 0.0f;
        if constexpr (t0 < T) x0 = operand(std::integral_constant<int, t0>{});
        if constexpr (t0 + 1 < T) x1 = operand(std::integral_constant<int, t0 + 1>{});
        unsigned hpk = pack_bf16(x0, x1);
        asm("" : "+v"(hpk));
        bh[i] = hpk;
        if constexpr (PRODS == 3) bl[i] = pack_bf16(x0 - __uint_as_float(hpk << 16), x1 - __uint_as_float(hpk & 0xffff0000u));
    };
    u32x4 bh, bl;
    pre(std::integral_constant<int, 0>{});
    static_for<4>([&](auto ic) { split_pair(std::integral_constant<int, 0>{}, ic, bh, bl); });
    __builtin_amdgcn_sched_barrier(0);
    static_for<S>([&](auto sc) {
        constexpr int s = decltype(sc)::value;
        FragB a[NB];
        const bf16x8 xh = __builtin_bit_cast(bf16x8, bh), xl = __builtin_bit_cast(bf16x8, bl);
        u32x4 nh = {}, nl = {};
        // MFMA m of the step (m = 3*ob + product) belongs to chunk m * 4 / (3*NB); products of one block stay in order hh, hl, lh.
        // A block's fragment leaves the ring at its first product and its slot is re-loaded at once with the fragment D blocks ahead.
        static_for<NCH>([&](auto cc) {
            constexpr int c = decltype(cc)::value;
            static_for<3 * NB>([&](auto mc) {
                constexpr int m = decltype(mc)::value, ob = m / 3, pr = m % 3, idx = s * NB + ob;
                if constexpr (m * NCH / (3 * NB) == c) {
                    if constexpr (pr == 0) {
                        a[ob] = ring.b[idx % D];
                        if constexpr (idx + D < S * NB) ring.b[idx % D] = wload_blk<RES, SBASE_DW + (idx + D) * 512u>(rs, la);
                    }
                    if constexpr (pr == 0 || (pr == 1 && PRODS == 3) || (pr == 2 && PRODS >= 2)) {
                        const bf16x8 wh = __builtin_bit_cast(bf16x8, a[ob].hi), wl = __builtin_bit_cast(bf16x8, a[ob].lo);
                        acc[ob] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pr == 2 ? wl : wh, pr == 1 ? xl : xh, acc[ob], 0, 0, 0);
                    }
                }
            });
            if constexpr (s + 1 < S) {
                if constexpr (c == 0) pre(std::integral_constant<int, s + 1>{});
                static_for<4 / NCH>([&](auto pc) { split_pair(std::integral_constant<int, s + 1>{}, std::integral_constant<int, c * (4 / NCH) + decltype(pc)::value>{}, nh, nl); });
            }
            __builtin_amdgcn_sched_barrier(0);
        });
        bh = nh; bl = nl;
    });
}

// precision-mode dispatch: MODE 0 = fp32 MFMA (exact), MODE 1 = split-bf16 x3
template <int MODE, int NB> struct RingSel { using type = Ring<NB>; };
template <int NB> struct RingSel<1, NB> { using type = RingB<NB>; };

// `la`: the lane's place in a fragment -- its index (fp32 kernel) or its LDS / buffer byte offsets (split-bf16 kernel, LAddr)
// fp32 kernel: the lane index, plus (training, spill mode: vanerf_query_forward_spill) where this lane's column of the X / auxiliary
// spills starts -- null pointers in the inference kernel, where every store below folds away
struct LLane {
    int lane;
    // spill mode: buffer stores, offset = voff (this lane's column: 4 (col + h npad)) + a wave-uniform row offset (2 r x 4 npad) -- one s_mul and
    // one store per value (global stores with 64-bit addresses took ~9 instructions each: 19.6 k instructions per group against 9.8 k without
    // the spills).  A lane that must not spill holds voff = SPILL_OFF: beyond both buffers' sizes, so the hardware drops the store.
    WRsrc xs, aux;       // Xs[X_ROWS][npad], Aux[AUX_ROWS][npad]
    unsigned voff, row4; // row4 = 4 npad (bytes per row)
    bool on;             // the kernel's SPILL parameter (a constant: the stores fold away in the other builds)
};
constexpr unsigned SPILL_OFF = 0xfffffff0u;
template <int L, int t> __device__ __forceinline__ void spill_x(const LLane& la, float v)
{
    if (la.on) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), la.xs, la.voff, (unsigned)(x_row_base(L) + 2 * t) * la.row4, 0);
}
template <int K> __device__ __forceinline__ void spill_aux(const LLane& la, float v)
{
    if (la.on) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), la.aux, la.voff, (unsigned)(2 * K) * la.row4, 0);
}
#ifndef VANERF_WAVES_PER_SIMD_SPILL
#define VANERF_WAVES_PER_SIMD_SPILL 1
#endif
template <int MODE> struct LaneSel { using type = LLane; };
template <> struct LaneSel<1> { using type = LAddr; };
__device__ __forceinline__ unsigned lane_of(const LLane& la) { return (unsigned)la.lane; }
template <int MODE, int NB, int T, int L> __device__ __forceinline__ typename RingSel<MODE, NB>::type ring_start_m(WRsrc rs, const typename LaneSel<MODE>::type& la)
{
    if constexpr (MODE == 0) return ring_start<NB, T>(rs, layer_offset(L), lane_of(la) * NB * 4u);
    else return ring_start_b<NB, T, (L >= LDS_FIRST), layer_offset_b(L)>(rs, la);
}

template <int MODE, int NB, int T, int L, class Op>
__device__ __forceinline__ void run_layer_m(f32x16 (&acc)[NB], typename RingSel<MODE, NB>::type& ring, WRsrc rs, const typename LaneSel<MODE>::type& la, Op&& operand)
{
    if constexpr (MODE == 0)
        run_layer<NB, T>(acc, ring, rs, layer_offset(L), lane_of(la) * NB * 4u, [&](auto tc) -> float {
            const float b = operand(tc);
            spill_x<L, decltype(tc)::value>(la, b); // (spill mode only)
            return b;
        });
    else run_layer_b<NB, T, (L >= LDS_FIRST), layer_offset_b(L), (((VANERF_P1_MASK >> L) & 1) ? 1 : ((VANERF_P2_MASK >> L) & 1) ? 2 : 3)>(acc, ring, rs, la, static_cast<Op&&>(operand));
}

// lane id from v_mbcnt, not from a register that would have to stay live (or be spilled) across the whole sample loop
__device__ __forceinline__ int lane_id_fresh()
{
    int l;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
    return l;
}


// Barriers of the block's four waves.  block_barrier_lds: LDS writes before it are visible after it (s_waitcnt lgkmcnt(0) + s_barrier; NOT
// __syncthreads(), whose workgroup-scope release also drains vmcnt -- the fragment prefetch and the stores of the previous group are in flight here).
__device__ __forceinline__ void block_barrier_lds() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// (More barriers inside a round were measured and lost: four rendezvous points in the layer stack cost 3.4 % -- 6.62 -> 6.84 ms.)

// ---------------------------------------------------------------------------------------------
// elementwise helpers
// ---------------------------------------------------------------------------------------------
// 1 / (1 + exp(-x)) on v_exp_f32 / v_rcp_f32 (1 ulp each; the gates and boundary weights they feed are compared at 1e-4)
__device__ __forceinline__ float sigmoid_f(float x) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * -1.44269504f)); }

// max(x, 0) as ONE v_med3_f32: fmaxf() costs two instructions (the backend first canonicalises the MFMA result)
__device__ __forceinline__ float relu_f(float x) { return __builtin_amdgcn_fmed3f(x, 0.0f, 3.0e38f); }

// torch.nn.Softplus(beta=100, threshold=20) (src/utils.py:656): x if 100x > 20 else log1p(exp(100x))/100, evaluated as
// max(x,0) + ln2/100 * log2(1 + exp2(-|x| * 100 log2(e))) on the raw v_exp_f32 / v_log_f32 (1 + e is in [1,2]: no denormal or
// range handling).  Above the threshold the correction is < 2.1e-11 and x + it rounds to x (x > 0.2), so no select is
// needed.  |error| < 1e-8 against fp64 over x in [-0.5, 0.5] (tools/probe_trig.hip).  6 VALU instructions, 2 transcendental.
__device__ __forceinline__ float softplus100(float x)
{
    const float e = __builtin_amdgcn_exp2f(fabsf(x) * -144.269504f);
    return fmaf(__builtin_amdgcn_logf(1.0f + e), 6.93147181e-3f, relu_f(x));
}

template <int NB> __device__ __forceinline__ void relu(f32x16 (&a)[NB])
{
#pragma unroll
    for (int ob = 0; ob < NB; ++ob)
#pragma unroll
        for (int r = 0; r < 16; ++r) a[ob][r] = relu_f(a[ob][r]);
}

template <int NB> __device__ __forceinline__ void softplus(f32x16 (&a)[NB])
{
#pragma unroll
    for (int ob = 0; ob < NB; ++ob)
#pragma unroll
        for (int r = 0; r < 16; ++r) a[ob][r] = softplus100(a[ob][r]);
}

// Activation at the point of use.  An activation output is consumed exactly once, as a B operand of the next layer.  The fp32 kernel
// applies the activation to a whole accumulator block right after its layer (VALU time adds to fp32-MFMA time wherever it stands); the
// split-bf16 kernel applies it when the operand is gathered, inside the software pipeline of run_layer_b, where VALU work runs beside
// the MFMAs of the previous k-step.  Same function, same values, same results.
enum Act { ACT_RELU = 1, ACT_SOFTPLUS = 2 };
template <int MODE, int ACT> __device__ __forceinline__ float lazy_act(float v)
{
    if constexpr (MODE == 0) return v; // already applied in bulk
    else if constexpr (ACT == ACT_RELU) return relu_f(v);
    else return softplus100(v);
}

// ---------------------------------------------------------------------------------------------
// grid_sample(bilinear, border, align_corners=True) (src/utils.py:136-151)
// ---------------------------------------------------------------------------------------------
struct Bilin {
    int o00, o01, o10, o11; // pixel offsets y*W + x of the nw, ne, sw, se taps
    float w00, w01, w10, w11;
};

__device__ __forceinline__ Bilin bilin_setup(float x, float y, int H, int W, float wm1, float hm1)
{
    float ix = ((x + 1.0f) / 2.0f) * wm1;
    float iy = ((y + 1.0f) / 2.0f) * hm1;
    ix = fminf(wm1, fmaxf(ix, 0.0f));
    iy = fminf(hm1, fmaxf(iy, 0.0f));
    float fx = floorf(ix), fy = floorf(iy);
    float wx1 = ix - fx, wx0 = (fx + 1.0f) - ix;
    float wy1 = iy - fy, wy0 = (fy + 1.0f) - iy;
    int x0 = (int)fx, y0 = (int)fy;
    int x1 = min(x0 + 1, W - 1), y1 = min(y0 + 1, H - 1); // out-of-range taps carry weight 0
    Bilin b;
    b.o00 = y0 * W + x0; b.o01 = y0 * W + x1; b.o10 = y1 * W + x0; b.o11 = y1 * W + x1;
    b.w00 = wx0 * wy0; b.w01 = wx1 * wy0; b.w10 = wx0 * wy1; b.w11 = wx1 * wy1;
    return b;
}

__device__ __forceinline__ float bilin_mix(const Bilin& b, float v00, float v01, float v10, float v11)
{
    return ((v00 * b.w00 + v01 * b.w01) + v10 * b.w10) + v11 * b.w11;
}

// Load through a wave-uniform base pointer plus a 32-bit BYTE offset: the form `global_load v, v_off, s[base]` needs the
// zero-extended 32-bit offset to be in bytes already (an element offset would have to be widened before the shift: two 64-bit
// VALU adds per load).
template <class T> __device__ __forceinline__ T ld_off(const void* __restrict__ base, unsigned byte_off)
{
    return *reinterpret_cast<const T*>(reinterpret_cast<const char*>(base) + byte_off);
}

// C4 = number of float4 groups to gather from a channel-last map with `C` channels, starting at channel coff.
template <int C4> __device__ __forceinline__ void gather(const float* __restrict__ map, const Bilin& b, int C, int coff, float (&dst)[C4 * 4])
{
    const unsigned o00 = (unsigned)(b.o00 * C + coff) * 4u, o01 = (unsigned)(b.o01 * C + coff) * 4u, o10 = (unsigned)(b.o10 * C + coff) * 4u,
                   o11 = (unsigned)(b.o11 * C + coff) * 4u;
#pragma unroll
    for (int i = 0; i < C4; ++i) {
        const float4 a = ld_off<float4>(map, o00 + 16 * i), c = ld_off<float4>(map, o01 + 16 * i), d = ld_off<float4>(map, o10 + 16 * i),
                     e = ld_off<float4>(map, o11 + 16 * i);
        dst[4 * i + 0] = bilin_mix(b, a.x, c.x, d.x, e.x);
        dst[4 * i + 1] = bilin_mix(b, a.y, c.y, d.y, e.y);
        dst[4 * i + 2] = bilin_mix(b, a.z, c.z, d.z, e.z);
        dst[4 * i + 3] = bilin_mix(b, a.w, c.w, d.w, e.w);
    }
}

template <int C4> __device__ __forceinline__ void load_row(const float* __restrict__ table, unsigned elem_off, float (&dst)[C4 * 4])
{
#pragma unroll
    for (int i = 0; i < C4; ++i) {
        const float4 a = ld_off<float4>(table, elem_off * 4u + 16 * i);
        dst[4 * i] = a.x; dst[4 * i + 1] = a.y; dst[4 * i + 2] = a.z; dst[4 * i + 3] = a.w;
    }
}

// Projection of a sample into the source view and its validity mask (src/model.py:780-803).  ONE definition, used by query_kernel
// and by the validity partition below, so that both always take the same decision.
struct Projected { float x, y, zn, mask; };
__device__ __forceinline__ Projected project_and_mask(const VanerfFrame& F, float wm1, float hm1, float px, float py, float pz)
{
    const float vx = fmaf(pz, F.KRT[2], fmaf(py, F.KRT[1], px * F.KRT[0])) + F.KRT[3];
    const float vy = fmaf(pz, F.KRT[6], fmaf(py, F.KRT[5], px * F.KRT[4])) + F.KRT[7];
    const float vz = fmaf(pz, F.KRT[10], fmaf(py, F.KRT[9], px * F.KRT[8])) + F.KRT[11];
    Projected r;
    r.x = 2.0f * ((vx / vz) / (F.width - 1.0f)) - 1.0f;
    r.y = 2.0f * ((vy / vz) / (F.height - 1.0f)) - 1.0f;
    r.zn = 2.0f * (vz - F.znear) / (F.zfar - F.znear) - 1.0f;
    const float eps = 1e-2f;
    const bool in_img = (r.x >= -1.0f - eps) && (r.x <= 1.0f + eps) && (r.y >= -1.0f - eps) && (r.y <= 1.0f + eps) && (r.zn >= -1.0f);
    const Bilin bi = bilin_setup(r.x, r.y, F.hi, F.wi, wm1, hm1);
    const float fg = bilin_mix(bi, ld_off<float>(F.mask, 4u * bi.o00), ld_off<float>(F.mask, 4u * bi.o01), ld_off<float>(F.mask, 4u * bi.o10),
                               ld_off<float>(F.mask, 4u * bi.o11));
    r.mask = (in_img && fg > 0.1f) ? 1.0f : 0.0f;
    return r;
}

// one GeoVisFusion scale (src/networks.py:83-94 / 96-104): gates, gated 2-layer MLP.
//   HC = channels per lane half (32 for the 64-channel map, 4 for the 8-channel map), NBO = output blocks,
//   NREG_MID = registers of the last hidden block that carry real channels
template <int MODE, int HC, int NBO, int NREG_MID, int l_at_a>
__device__ __forceinline__ void geo_scale(WRsrc W, int lane, const typename LaneSel<MODE>::type& la, typename RingSel<MODE, 1>::type& ring_at,
                                          float (&pix)[HC], float (&nn)[HC], float (&tw)[HC], float s0, float s1,
                                          f32x16 (&outacc)[NBO])
{
    constexpr int TIN = 3 * HC + 2;
    constexpr int TOUT = (NBO - 1) * 16 + NREG_MID;
    auto input = [&](auto tc) -> float {
        constexpr int t = decltype(tc)::value;
        if constexpr (t < HC) return pix[t];
        else if constexpr (t < 2 * HC) return nn[t - HC];
        else if constexpr (t < 3 * HC) return tw[t - 2 * HC];
        else if constexpr (t == 3 * HC) return s0;
        else return s1;
    };
    f32x16 at[1];
    zero<1>(at);
    run_layer_m<MODE, 1, TIN, l_at_a>(at, ring_at, W, la, input);
    auto r_gate = ring_start_m<MODE, 1, 6, l_at_a + 1>(W, la);
    auto r_mid = ring_start_m<MODE, NBO, TIN, l_at_a + 2>(W, la);
    if constexpr (MODE == 0) relu<1>(at);
    f32x16 gate[1];
    zero<1>(gate);
    run_layer_m<MODE, 1, 6, l_at_a + 1>(gate, r_gate, W, la, [&](auto tc) -> float { return lazy_act<MODE, ACT_RELU>(at[0][decltype(tc)::value]); });
    // gates live in rows 0..2 = registers 0..2 of the h = 0 lanes
    const float a0 = __shfl(sigmoid_f(gate[0][0]), lane & 31);
    const float a1 = __shfl(sigmoid_f(gate[0][1]), lane & 31);
    const float a2 = __shfl(sigmoid_f(gate[0][2]), lane & 31);
    if constexpr (MODE == 0) { constexpr int A = HC == 32 ? AUX_G0 : AUX_G1; spill_aux<A>(la, a0); spill_aux<A + 1>(la, a1); spill_aux<A + 2>(la, a2); }
#pragma unroll
    for (int t = 0; t < HC; ++t) { pix[t] *= a0; nn[t] *= a1; tw[t] *= a2; }
    f32x16 mid[NBO];
    zero<NBO>(mid);
    run_layer_m<MODE, NBO, TIN, l_at_a + 2>(mid, r_mid, W, la, input);
    auto r_out = ring_start_m<MODE, NBO, TOUT, l_at_a + 3>(W, la);
    if constexpr (MODE == 0) relu<NBO>(mid);
    zero<NBO>(outacc);
    run_layer_m<MODE, NBO, TOUT, l_at_a + 3>(outacc, r_out, W, la,
                         [&](auto tc) -> float { constexpr int t = decltype(tc)::value; return lazy_act<MODE, ACT_RELU>(mid[t / 16][t % 16]); });
}

template <int MODE, bool SPILL = false>
__global__ __launch_bounds__(64 * WPB<MODE>, MODE == 1 ? VANERF_WAVES_PER_SIMD_B : (SPILL ? VANERF_WAVES_PER_SIMD_SPILL : VANERF_WAVES_PER_SIMD)) void query_kernel(const QueryParams P)
{
    static_assert(!SPILL || MODE == 0, "the training spill runs on the fp32 kernel");
    constexpr int WAVES_PER_BLOCK = WPB<MODE>, BLOCK = 64 * WAVES_PER_BLOCK;

    const int lane_k = threadIdx.x & 63;
    [[maybe_unused]] const int lane = lane_k, j = lane & 31, h = lane >> 5; // (shadowed inside the sample loop)
    [[maybe_unused]] const long long wave = (long long)blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6); // STAMPS builds
    const long long ngroups = (P.n + 31) / 32;
    const VanerfFrame& F = P.f;
#ifdef VANERF_EXP_SHARE_BOUND // timing experiment, wrong results: only wave 0 of a block fetches fragments (zero-record descriptor: the others' loads return 0 without traffic)
    const unsigned wbytes_eff = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) == 0 ? P.wbytes : 0u;
#else
    const unsigned wbytes_eff = P.wbytes;
#endif
    const WRsrc W = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(P.w), 0, wbytes_eff, 0x00020000); // kernarg-derived: wave-uniform
    [[maybe_unused]] WRsrc xs_rs = W, aux_rs = W; // spill mode: the two spills as buffers of their exact sizes (the host checks they stay below 4 GB)
    if constexpr (SPILL) {
        xs_rs = __builtin_amdgcn_make_buffer_rsrc(P.xs, 0, (unsigned)(X_ROWS * 4ll * P.npad), 0x00020000);
        aux_rs = __builtin_amdgcn_make_buffer_rsrc(P.aux, 0, (unsigned)(AUX_ROWS * 4ll * P.npad), 0x00020000);
    }

#ifdef VANERF_STAMPS
    unsigned long long phase_cycles[N_PHASES] = {};
    unsigned long long t_prev, rt0;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(rt0)::"memory");
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_prev)::"memory");
#endif
    // split-bf16 kernel (one wave per SIMD, 512 registers): the lane half's 21 key points stay in registers for the whole launch
    // (63 VGPRs) -- per-group scalar loads of them exposed their latency 21 times per group with no second wave to hide it
    [[maybe_unused]] float kpx[PE_KPT_PER_HALF], kpy[PE_KPT_PER_HALF], kpz[PE_KPT_PER_HALF];
    if constexpr (MODE == 1 && W2) {
        if (threadIdx.x < 2 * PE_KPT_PER_HALF) s_dyn[LDS_KPT / 16 + threadIdx.x] = reinterpret_cast<const u32x4*>(F.kpt_cam)[threadIdx.x];
    } else if constexpr (MODE == 1) {
        const float4* __restrict__ kpl = reinterpret_cast<const float4*>(F.kpt_cam) + (h ? PE_KPT_PER_HALF : 0);
#pragma unroll
        for (int i = 0; i < PE_KPT_PER_HALF; ++i) { const float4 k = kpl[i]; kpx[i] = k.x; kpy[i] = k.y; kpz[i] = k.z; }
    }
    unsigned short_groups = 0;
    // The six per-sample inputs of a group are fetched one group ahead (its index is known from the early claim): their HBM latency
    // is otherwise the first thing a group waits for, and a single wave per SIMD (bf16 kernel) has nobody to hide it.
    struct SampleIn { float px, py, pz, sdf; int knn; unsigned char vis; };
    auto fetch = [&](unsigned grp, int j) {
        const long long sr = (long long)grp * 32 + j;
        long long sc = sr < P.n ? sr : P.n - 1; // also covers a claim beyond the last group (never used)
        if (P.order) sc = P.order[sc];
        SampleIn in;
        in.px = P.pts[3 * sc]; in.py = P.pts[3 * sc + 1]; in.pz = P.pts[3 * sc + 2];
        in.sdf = P.qsdf[sc]; in.knn = P.knn_in[sc]; in.vis = P.qvis[sc];
        return in;
    };
    // Work distribution.  A block claims FOUR consecutive 32-sample groups per round from a device-scope counter (wave w takes group
    // base + w) and its waves meet at one barrier per round: they walk the weight stream together, so three of the four fetches of every
    // fragment hit the CU's L1 instead of L2 (launch 7.6 -> 6.8 ms; with per-wave claims the waves drift apart and every wave streams the
    // 660 KB from L2 on its own).  The claim for round r + 2 is issued at the start of round r (thread 0) and published through LDS at the
    // start of round r + 1: no wave waits for the atomic.  The same barrier makes the "no valid sample" decision block-uniform, which is what
    // allows barriers inside the layer stack.  Every block leaves the loop once the counter passes ngroups: the grid always drains.
    // control words in LDS: fp32 kernel static, split-bf16 kernel inside its dynamic region (see the LDS map above)
    __shared__ unsigned s_ctrl0[MODE == 0 ? 2 + 2 * WAVES_PER_BLOCK : 1];
    auto s_base = [&](unsigned i) -> auto& { if constexpr (MODE == 0) return s_ctrl0[i]; else return lds_ctrl()[i]; };
    auto s_valid = [&](unsigned pp, unsigned k) -> auto& { if constexpr (MODE == 0) return s_ctrl0[2 + pp * WAVES_PER_BLOCK + k]; else return lds_ctrl()[2 + pp * WAVES_PER_BLOCK + k]; };
    const unsigned wv = threadIdx.x >> 6;
    [[maybe_unused]] const unsigned wv_u = (unsigned)__builtin_amdgcn_readfirstlane((int)wv); // the same in a scalar register
    unsigned pending = 0, par = 1;
    if (threadIdx.x == 0) s_base(0) = atomicAdd(P.queue, (unsigned)WAVES_PER_BLOCK);
    if constexpr (MODE == 1) { // resident fragments: one copy per block (the launch is persistent: 256 blocks x 156 KB from L2, once)
        const u32x4* __restrict__ src = reinterpret_cast<const u32x4*>(P.w) + RES_BASE_DW / 4;
        for (unsigned i = threadIdx.x; i < RES_DW / 4; i += BLOCK) s_dyn[LDS_RES / 16 + i] = src[i];
        if constexpr (RING) { // phase 0 of the ring, once: every full round re-issues it for its successor (see phase_begin)
            const LAddr la0 = make_laddr(lane, wv_u);
            ring_dma<0>(W, la0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    }
    __syncthreads();
    if constexpr (MODE == 1) { // the short path's constant latent (wave 0; the layer's fragments are resident by now)
        if (wv_u == 0u) {
            const LAddr la0 = make_laddr(lane, wv_u);
            auto r0 = ring_start_m<MODE, 1, 65, L_IBR>(W, la0);
            f32x16 lat0[1];
            zero<1>(lat0);
            const float one0 = (lane >> 5) ? 0.0f : 1.0f;
            run_layer_m<MODE, 1, 65, L_IBR>(lat0, r0, W, la0, [&](auto tc) -> float { return decltype(tc)::value < 64 ? 0.0f : one0; });
            if ((lane & 31) == 0)
                static_for<16>([&](auto rc) { constexpr int r = decltype(rc)::value; reinterpret_cast<lds_u32_t*>((size_t)LDS_LAT0)[(lane >> 5) * 16 + r] = __float_as_uint(lat0[0][r]); });
        }
        __syncthreads();
    }
    unsigned base_cur = s_base(0);
    if (threadIdx.x == 0) pending = atomicAdd(P.queue, (unsigned)WAVES_PER_BLOCK);
    SampleIn in_next = fetch(base_cur + wv, j);
    while (base_cur < (unsigned)ngroups) {
        const long long g = (long long)base_cur + wv; // a wave whose group lies beyond the last one runs with live == false: the block's barriers stay matched
        const SampleIn in = in_next;
        // split-bf16 kernel: the lane id and what follows from it are re-derived in every round (v_mbcnt, volatile), so that none of it is carried
        // around the loop in registers the 256-register budget does not have
        const int lane = MODE == 1 ? lane_id_fresh() : lane_k, j = lane & 31, h = lane >> 5;
        const float one_h0 = h ? 0.0f : 1.0f; // B operand of the bias k-step
        // the lane's fragment offsets, opaque per round (nothing derived from them is hoisted out of the loop and parked in registers)
        typename LaneSel<MODE>::type la;
        [[maybe_unused]] unsigned kp_lds = 0;
        if constexpr (MODE == 1) { la = make_laddr(lane, wv_u); kp_lds = LDS_KPT + ((la.w[0] >> 9) & 1u) * (PE_KPT_PER_HALF * 16u); }
        else {
            la.lane = lane; la.voff = SPILL_OFF; la.row4 = 0; la.on = SPILL;
            if constexpr (SPILL) {
                la.xs = xs_rs; la.aux = aux_rs;
                // the row stride, opaque per round: hoisted out of the work loop the ~1 100 row offsets are parked in registers.  (Set on every
                // path: a value that depends on the group index is divergent to the compiler, and a divergent store offset is a waterfall loop.)
                unsigned r4 = (unsigned)P.npad * 4u;
                asm volatile("" : "+s"(r4));
                la.row4 = r4;
                // column = the slot's own place g * 32 + j (also for lanes beyond n: their output gradient is zero); no column beyond the last group
                if (g < ngroups) la.voff = 4u * ((unsigned)g * 32u + (unsigned)j) + (unsigned)h * r4;
            }
        }
        const long long s_raw = g * 32 + j;
        const bool live = s_raw < P.n;
        long long s = live ? s_raw : P.n - 1;
        if (P.order) s = P.order[s];
        const float px = in.px, py = in.py, pz = in.pz;
        const float q_sdf = in.sdf;
        const float q_vis = in.vis ? 1.0f : 0.0f;

        // ---- projection into the source view, validity mask, boundary weight (src/model.py:780-821) ----
        const Projected pr = project_and_mask(F, P.wm1[0], P.hm1[0], px, py, pz);
        const float x = pr.x, y = pr.y, zn = pr.zn, mask = pr.mask;
        const bool wave_valid = __builtin_amdgcn_ballot_w64(mask > 0.0f) != 0ull;
        if (lane == 0) s_valid(par, wv) = wave_valid ? 1u : 0u;
        if (threadIdx.x == 0) s_base(par) = pending;
        block_barrier_lds();
        const unsigned base_next = (unsigned)__builtin_amdgcn_readfirstlane((int)s_base(par));
        unsigned anyv = 0;
#pragma unroll
        for (int k = 0; k < WAVES_PER_BLOCK; ++k) anyv |= s_valid(par, k);
        const bool any_valid = SPILL || __builtin_amdgcn_readfirstlane((int)anyv) != 0; // (spill mode: every group writes every row)
        par ^= 1u;
        if (threadIdx.x == 0) pending = atomicAdd(P.queue, (unsigned)WAVES_PER_BLOCK);
        in_next = fetch(base_next + wv, j);
        base_cur = base_next;
        __builtin_amdgcn_sched_barrier(0x000F); // keep these loads here (the scheduler sinks loads to their first use)
        const Bilin bi = bilin_setup(x, y, F.hi, F.wi, P.wm1[0], P.hm1[0]); // source-image taps (the same the mask used)
        float pw;
        {
            float ux = 0.5f * x + 0.5f, uy = 0.5f * y + 0.5f, uz = 0.5f * zn + 0.5f;
            float dx = fminf(ux, 1.0f - ux), dy = fminf(uy, 1.0f - uy), dz = fminf(uz, 1.0f - uz);
            float wx = sigmoid_f(5.0f * (dx * 10.0f - 1.0f)), wy = sigmoid_f(5.0f * (dy * 10.0f - 1.0f)),
                  wz = sigmoid_f(5.0f * (dz * 10.0f - 1.0f));
            float p = (wx * wy) * wz * mask;
            pw = p / (p + 1e-6f);
        }
        if constexpr (MODE == 0) spill_aux<AUX_PW>(la, pw);

        STAMP(0); // front end
        // ---- 1-NN vertex (src/networks.py:27-33): found by vanerf_mesh_query_accel (same pass as the SDF) ----------
        const int nn_idx = in.knn;
        const int tw_idx = nn_idx >= VANERF_NV_HAND ? nn_idx - VANERF_NV_HAND : nn_idx + VANERF_NV_HAND;
        const float vis_nn = ld_off<float>(F.vert_vis, 4u * nn_idx), vis_tw = ld_off<float>(F.vert_vis, 4u * tw_idx);
        const float sc0 = h ? q_vis : q_sdf;   // k-pair (sdf | qvis)
        const float sc1 = h ? vis_tw : vis_nn; // k-pair (vis_nn | vis_twin)

        STAMP(1); // 1-NN
        // ---- GeoVisFusion (src/networks.py:75-106) ---------------------------------------------------------
        // Every layer's fragment ring is started ahead of the previous layer's epilogue (see ring_start); the gathers of the
        // second scale and of the texture branch are issued here too, so their L2 latency hides behind the first layers.
        [[maybe_unused]] const unsigned v4 = (unsigned)lane * 16u; // byte offset of the lane in a 4-block fp32 fragment
        // chain operand: register r of block b of a previous accumulator array, then the bias step
        auto chain = [&](auto& src, auto tc, auto nsteps) -> float {
            constexpr int t = decltype(tc)::value;
            if constexpr (t < decltype(nsteps)::value) return src[t / 16][t % 16];
            else return one_h0;
        };
        auto chain_sp = [&](auto& src, auto tc, auto nsteps) -> float { // the same through Softplus (lazy_act)
            constexpr int t = decltype(tc)::value;
            if constexpr (t < decltype(nsteps)::value) return lazy_act<MODE, ACT_SOFTPLUS>(src[t / 16][t % 16]);
            else return one_h0;
        };
        float row[32]; // h = 0: nearest vertex [img3|tex8|gf18], h = 1: twin vertex
        float qi[4], qt[8];
        auto tex_gathers = [&]() { // issued ~2 layers before the texture branch needs them
            load_row<8>(F.vfeat_tex, (unsigned)((h ? tw_idx : nn_idx) * 32), row);
            gather<1>(F.img, bi, 4, 0, qi);
            const Bilin bt = bilin_setup(x, y, F.ht, F.wt, P.wm1[1], P.hm1[1]);
            gather<2>(F.tex, bt, 8, 0, qt);
        };
        f32x16 pool[4]; // [mean64 | var64]
        f32x16 head[1];
        // A sample whose projection misses the source view / foreground mask has pixel weight 0: its pooled latent is exactly
        // zero (0 * x), the density head is masked out by eval_func and only the colour branch (fed by the bias of
        // ibr_compress) survives.  When ALL 32 samples of the wave are such samples, GeoVisFusion, mlp_geo.layers1 and the head
        // (82 % of the MFMAs) are skipped -- same bits, wave-uniform branch.  With real foreground masks most samples are.
        if (any_valid) {
            auto r_at0 = ring_start_m<MODE, 1, 98, L_GEO_AT0_A>(W, la);
            f32x16 g64[2], g8[1];
            float pix8[4], nn8[4], tw8[4];
            {
                float pix[32], nn[32], tw[32];
                const Bilin b0 = bilin_setup(x, y, F.h0, F.w0, P.wm1[2], P.hm1[2]);
                gather<8>(F.geo0, b0, 64, 32 * h, pix);
                load_row<8>(F.vfeat0, (unsigned)(nn_idx * 64 + 32 * h), nn);
                load_row<8>(F.vfeat0, (unsigned)(tw_idx * 64 + 32 * h), tw);
                auto geo1_gathers = [&]() {
                    const Bilin b1 = bilin_setup(x, y, F.h1, F.w1, P.wm1[3], P.hm1[3]);
                    gather<1>(F.geo1, b1, 8, 4 * h, pix8);
                    load_row<1>(F.vfeat1, (unsigned)(nn_idx * 8 + 4 * h), nn8);
                    load_row<1>(F.vfeat1, (unsigned)(tw_idx * 8 + 4 * h), tw8);
                };
                if constexpr (!(MODE == 1 && W2)) geo1_gathers();
                STAMP(2); // geo gathers
                geo_scale<MODE, 32, 2, 16, L_GEO_AT0_A>(W, lane, la, r_at0, pix, nn, tw, sc0, sc1, g64);
                if constexpr (MODE == 1 && W2) geo1_gathers();
                STAMP(3); // geo0 layers
            }
            // mlp0's ring (7 x dwordx4) starts before the small second scale runs
            constexpr unsigned base0 = layer_offset(L_MLP0);
            constexpr int D0 = PE_FEATS;
            WFrag<4> ring0[D0];
            RingB<4> ring0b;
            {
                auto r_at1 = ring_start_m<MODE, 1, 14, L_GEO_AT1_A>(W, la);
                if constexpr (MODE == 0) {
                    static_for<D0>([&](auto fc) { constexpr int f = decltype(fc)::value; ring0[f] = wload<4>(W, (base0 + f * 256) * 4u, v4); });
                } else if constexpr (!W2) {
                    if constexpr (MODE == 1) ring0b = ring_start_b<4, 180, false, layer_offset_b(L_MLP0)>(W, la);
                }
                geo_scale<MODE, 4, 1, 4, L_GEO_AT1_A>(W, lane, la, r_at1, pix8, nn8, tw8, sc0, sc1, g8);
                if constexpr (MODE == 1 && W2) ring0b = ring_start_b<4, 180, false, layer_offset_b(L_MLP0)>(W, la);
                STAMP(4); // geo1
            }


            // ---- mlp_geo.layers1 (src/utils.py:822-852): [PE294 | geo64] -> 128 -> 128 -> [. | geo8] -> 120 -> 64 ----
            f32x16 xv[2];
            {
                f32x16 a0[4];
                zero<4>(a0);
                {
                    // SpatialEncoder 'rel_z_decay' (src/spatial.py:71-72, 109-117) in source-camera coordinates; the 7
                    // features of key point i (one per lane half) are k-steps 7i..7i+6, their fragments ring slots 0..6
                    float cx = fmaf(pz, F.extrin[2], fmaf(py, F.extrin[1], px * F.extrin[0])) + F.extrin[3];
                    float cy = fmaf(pz, F.extrin[6], fmaf(py, F.extrin[5], px * F.extrin[4])) + F.extrin[7];
                    float cz = fmaf(pz, F.extrin[10], fmaf(py, F.extrin[9], px * F.extrin[8])) + F.extrin[11];
                    const float4* __restrict__ kpg = reinterpret_cast<const float4*>(F.kpt_cam);
                    // the 7 positional-encoding features of key point i: dz, sin/cos(pi 2^l dz) for l = 0..2, all times the Gaussian decay.
                    // v_sin_f32 / v_cos_f32 take revolutions, so the arguments are dz/2, dz, 2dz exactly (|error| < 1.6e-7 absolute over
                    // the hand's extent, tools/probe_trig.hip)
                    auto pe_features = [&](int i, float (&feat)[PE_FEATS]) {
                        // fp32 kernel: key points through the scalar cache (wave-uniform addresses), selected per lane half
                        float kx, ky, kz;
                        if constexpr (MODE == 1 && W2) {
                            const u32x4 k = *reinterpret_cast<const lds_u32x4_t*>((size_t)(kp_lds + 16u * i));
                            kx = __uint_as_float(k.x); ky = __uint_as_float(k.y); kz = __uint_as_float(k.z);
                        } else if constexpr (MODE == 1) { kx = kpx[i]; ky = kpy[i]; kz = kpz[i]; }
                        else {
                            const float4 k0 = kpg[i], k1 = kpg[PE_KPT_PER_HALF + i];
                            kx = h ? k1.x : k0.x; ky = h ? k1.y : k0.y; kz = h ? k1.z : k0.z;
                        }
                        const float ddx = cx - kx, ddy = cy - ky, ddz = cz - kz;
                        const float d2 = (ddx * ddx + ddy * ddy) + ddz * ddz;
                        const float wk = __expf(-d2 * F.pe_inv_2sigma2);
                        const float dz = F.pe_scale * ddz;
                        const float s0 = __builtin_amdgcn_sinf(0.5f * dz), c0 = __builtin_amdgcn_cosf(0.5f * dz);
                        const float s1 = __builtin_amdgcn_sinf(dz), c1 = __builtin_amdgcn_cosf(dz);
                        const float s2 = __builtin_amdgcn_sinf(2.0f * dz), c2 = __builtin_amdgcn_cosf(2.0f * dz);
                        feat[0] = dz * wk; feat[1] = s0 * wk; feat[2] = c0 * wk; feat[3] = s1 * wk; feat[4] = c1 * wk;
                        feat[5] = s2 * wk; feat[6] = c2 * wk;
                    };
                    if constexpr (MODE == 0) {
                        // fully unrolled: as a run-time loop hipcc hoists the seven prefetch loads of an iteration above its first use and
                        // waits vmcnt(0) (no prefetch left), and copies the 64 accumulator registers around the back edge
                        static_for<PE_KPT_PER_HALF>([&](auto ic) {
                            constexpr int i = decltype(ic)::value;
                            float feat[PE_FEATS];
                            pe_features(i, feat);
                            static_for<D0>([&](auto fc) {
                                constexpr int f = decltype(fc)::value;
                                const WFrag<4> a = ring0[f];
                                // step 7(i+1)+f; after the last key point these are the first steps of the chain part below
                                ring0[f] = wload<4>(W, (base0 + ((i + 1) * D0 + f) * 256) * 4u, v4);
                                spill_x<L_MLP0, i * D0 + f>(la, feat[f]);
                                mfma_step<4>(a0, a, feat[f]);
                            });
                        });
                        constexpr unsigned nextp = base0 + (PE_KPT_PER_HALF + 1) * D0 * 256;
                        // remaining 32 + 1 steps: geo64 (two blocks) and the bias; ring slot = step % 7 (147 = 21 * 7)
                        constexpr int TREM = 33;
                        static_for<TREM>([&](auto tc) {
                            constexpr int t = decltype(tc)::value;
                            const float b = chain(g64, tc, std::integral_constant<int, 32>{});
                            const WFrag<4> a = ring0[t % D0];
                            if constexpr (t + D0 < TREM) ring0[t % D0] = wload<4>(W, (nextp + t * 256) * 4u, v4);
                            spill_x<L_MLP0, PE_KPT_PER_HALF * D0 + t>(la, b);
                            mfma_step<4>(a0, a, b);
                        });
                    } else {
                        // split-bf16: k-pairs 0..146 are the features (pair 7i+f = feature f of key point i), 147..178 the geo64 chain,
                        // 179 the bias.  A bf16 step takes 8 consecutive pairs, so key point i is computed right before the first step
                        // that needs it (at most two key points are live at a time).
                        float feat[PE_KPT_PER_HALF][PE_FEATS];
                        run_layer_b<4, 180, false, layer_offset_b(L_MLP0), (((VANERF_P1_MASK >> L_MLP0) & 1) ? 1 : ((VANERF_P2_MASK >> L_MLP0) & 1) ? 2 : 3)>(a0, ring0b, W, la,
                            [&](auto tc) -> float {
                                constexpr int t = decltype(tc)::value;
                                if constexpr (t < 147) return feat[t / 7][t % 7];
                                else return chain(g64, std::integral_constant<int, t - 147>{}, std::integral_constant<int, 32>{});
                            },
                            [&](auto sc) {
                                constexpr int s_ = decltype(sc)::value;
                                constexpr int lo_k = s_ == 0 ? 0 : (8 * s_ - 1) / 7 + 1, hi_k = (8 * s_ + 7) / 7; // key points first needed by this step
                                static_for<(hi_k < PE_KPT_PER_HALF ? hi_k : PE_KPT_PER_HALF - 1) - lo_k + 1 < 0 ? 0 : (hi_k < PE_KPT_PER_HALF ? hi_k : PE_KPT_PER_HALF - 1) - lo_k + 1>([&](auto dc) {
                                    constexpr int i = lo_k + decltype(dc)::value;
                                    pe_features(i, feat[i]);
                                });
                            });
                    }
                }
                STAMP(5); // mlp0 (PE + geo64)
                auto r1 = ring_start_m<MODE, 4, 65, L_MLP1>(W, la);
                if constexpr (MODE == 0) softplus<4>(a0);
                f32x16 a1[4];
                zero<4>(a1);
                run_layer_m<MODE, 4, 65, L_MLP1>(a1, r1, W, la, [&](auto tc) -> float { return chain_sp(a0, tc, std::integral_constant<int, 64>{}); });
                auto r2 = ring_start_m<MODE, 4, 69, L_MLP2>(W, la);
                if constexpr (MODE == 0) softplus<4>(a1);
                zero<4>(a0);
                run_layer_m<MODE, 4, 69, L_MLP2>(a0, r2, W, la, [&](auto tc) -> float {
                    constexpr int t = decltype(tc)::value;
                    if constexpr (t < 64) return lazy_act<MODE, ACT_SOFTPLUS>(a1[t / 16][t % 16]);
                    else if constexpr (t < 68) return g8[0][t - 64];
                    else return one_h0;
                });
                auto r3 = ring_start_m<MODE, 2, 61, L_MLP3>(W, la);
                if constexpr (MODE == 0) softplus<4>(a0);
                zero<2>(xv);
                run_layer_m<MODE, 2, 61, L_MLP3>(xv, r3, W, la, [&](auto tc) -> float { return chain_sp(a0, tc, std::integral_constant<int, 60>{}); });
            }
            STAMP(6); // softplus x3 + mlp1..3
            if constexpr (!(MODE == 1 && W2)) tex_gathers();
            // ---- PoolModule mean/var over V = 1 views (src/utils.py:744-779, 854-880) --------------------------
            auto rh0 = ring_start_m<MODE, 2, 65, L_HEAD0>(W, la);
            if constexpr (MODE == 0)
                static_for<32>([&](auto rc) { constexpr int r = decltype(rc)::value; spill_aux<AUX_XV + r>(la, xv[r / 16][r % 16]); });
    #pragma unroll
            for (int b = 0; b < 2; ++b)
    #pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float m = pw * xv[b][r];
                    float d = xv[b][r] - m;
                    pool[b][r] = m;
                    pool[2 + b][r] = pw * (d * d);
                }
            // ---- mlp_geo.layers2: 128 -> 64 -> 64 -> 2 (src/utils.py:709-719) ----------------------------------
            {
                f32x16 m0[2], m1[2];
                zero<2>(m0);
                run_layer_m<MODE, 2, 65, L_HEAD0>(m0, rh0, W, la, [&](auto tc) -> float { return chain(pool, tc, std::integral_constant<int, 64>{}); });
                auto rh1 = ring_start_m<MODE, 2, 33, L_HEAD1>(W, la);
                if constexpr (MODE == 0) softplus<2>(m0);
                zero<2>(m1);
                run_layer_m<MODE, 2, 33, L_HEAD1>(m1, rh1, W, la, [&](auto tc) -> float { return chain_sp(m0, tc, std::integral_constant<int, 32>{}); });
                auto rh2 = ring_start_m<MODE, 1, 33, L_HEAD2>(W, la);
                if constexpr (MODE == 0) softplus<2>(m1);
                zero<1>(head);
                run_layer_m<MODE, 1, 33, L_HEAD2>(head, rh2, W, la, [&](auto tc) -> float { return chain_sp(m1, tc, std::integral_constant<int, 32>{}); });
            }
        } else {
            if constexpr (!(MODE == 1 && W2)) tex_gathers();
            zero<4>(pool);
            zero<1>(head);
        }
        if (!wave_valid && lane == 0 && g < ngroups) ++short_groups; // counted by the wave's own samples (what bench.py prices), not by the block's path
        if constexpr (MODE == 1 && W2) tex_gathers();
        STAMP(7); // pool + head
        // ---- ibr_compress_gfeat 128 -> 24 (src/model.py:921) ------------------------------------------------
        f32x16 lat[1];
        typename RingSel<MODE, 3>::type r_ta;
        if (MODE == 1 && !any_valid) {
            // all-invalid group: the pooled latent is exactly zero, the layer returns its bias -- the block's constant (same chain, same bits)
            r_ta = ring_start_m<MODE, 3, 49, L_TEX_AT_A>(W, la);
            if constexpr (MODE == 1)
                static_for<16>([&](auto rc) { constexpr int r = decltype(rc)::value; lat[0][r] = __uint_as_float(reinterpret_cast<const lds_u32_t*>((size_t)LDS_LAT0)[h * 16 + r]); });
        } else {
            auto r_ibr = ring_start_m<MODE, 1, 65, L_IBR>(W, la);
            r_ta = ring_start_m<MODE, 3, 49, L_TEX_AT_A>(W, la);
            zero<1>(lat);
            run_layer_m<MODE, 1, 65, L_IBR>(lat, r_ibr, W, la, [&](auto tc) -> float { return chain(pool, tc, std::integral_constant<int, 64>{}); });
        }
        STAMP(8); // ibr
        // ---- TexVisFusion per-sample part (src/networks.py:281-293) -----------------------------------------
        f32x16 rgb[1];
        {
            float q[6];
            q[0] = h ? qt[3] : qi[0]; q[1] = h ? qt[4] : qi[1]; q[2] = h ? qt[5] : qi[2];
            q[3] = h ? qt[6] : qt[0]; q[4] = h ? qt[7] : qt[1]; q[5] = h ? 0.0f : qt[2];
            const float t0 = h ? vis_nn : q_vis; // k-pair (qvis | vis_nn)
            const float t1 = h ? 0.0f : vis_tw;  // k-pair (vis_twin | -)
            f32x16 latg = lat[0];
            auto tex_in = [&](auto tc) -> float {
                constexpr int t = decltype(tc)::value;
                if constexpr (t < 29) return row[t];
                else if constexpr (t < 35) return q[t - 29];
                else if constexpr (t < 47) return latg[t - 35];
                else if constexpr (t == 47) return t0;
                else return t1;
            };
            auto from_ta = [&](auto& ta_) { return [&](auto tc) -> float { constexpr int t = decltype(tc)::value; return lazy_act<MODE, ACT_RELU>(ta_[t / 16][t % 16]); }; };
            f32x16 ta[3];
            zero<3>(ta);
            run_layer_m<MODE, 3, 49, L_TEX_AT_A>(ta, r_ta, W, la, tex_in);
            auto r_tg = ring_start_m<MODE, 1, 48, L_TEX_AT_B>(W, la);
            if constexpr (MODE == 0) relu<3>(ta);
            f32x16 tg[1];
            zero<1>(tg);
            run_layer_m<MODE, 1, 48, L_TEX_AT_B>(tg, r_tg, W, la, from_ta(ta));
            auto r_tb = ring_start_m<MODE, 3, 49, L_TEX_A>(W, la);
            // six gates: rows 0..3 -> h = 0 lanes regs 0..3, rows 4,5 -> h = 1 lanes regs 0,1
            float m0 = sigmoid_f(tg[0][0]), m1 = sigmoid_f(tg[0][1]), m2 = sigmoid_f(tg[0][2]), m3 = sigmoid_f(tg[0][3]);
            float o0 = __shfl_xor(m0, 32), o1 = __shfl_xor(m1, 32), o2 = __shfl_xor(m2, 32);
            const float gq = h ? o0 : m0;     // gate 0: query feature
            const float g11 = h ? o2 : m1;    // gate 1 (nearest) / gate 2 (twin): [img3|tex8]
            const float ggf = h ? m0 : m3;    // gate 3 (nearest) / gate 4 (twin): global feature
            const float glat = h ? m1 : o1;   // gate 5: compressed latent
            if constexpr (MODE == 0) {
                spill_aux<AUX_TEX>(la, gq); spill_aux<AUX_TEX + 1>(la, g11); spill_aux<AUX_TEX + 2>(la, ggf); spill_aux<AUX_TEX + 3>(la, glat);
                static_for<12>([&](auto rc) { constexpr int r = decltype(rc)::value; spill_aux<AUX_LAT + r>(la, lat[0][r]); });
            }
#pragma unroll
            for (int t = 0; t < 11; ++t) row[t] *= g11;
#pragma unroll
            for (int t = 11; t < 29; ++t) row[t] *= ggf;
#pragma unroll
            for (int t = 0; t < 6; ++t) q[t] *= gq;
#pragma unroll
            for (int r = 0; r < 12; ++r) latg[r] = lat[0][r] * glat;
            zero<3>(ta);
            run_layer_m<MODE, 3, 49, L_TEX_A>(ta, r_tb, W, la, tex_in);
            auto r_rgb = ring_start_m<MODE, 1, 48, L_TEX_B>(W, la);
            if constexpr (MODE == 0) relu<3>(ta);
            zero<1>(rgb);
            run_layer_m<MODE, 1, 48, L_TEX_B>(rgb, r_rgb, W, la, from_ta(ta));
        }
        STAMP(9); // tex
        // ---- eval_func (src/model.py:1140-1160): rows 0,1 of the head / 0..2 of the colour live in the h = 0 lanes ----
        // ring build: the DMA this wave issued in the round's last phase (phase 0 of the next full round) has landed before the wave reaches
        // the next top-of-round barrier -- waited for here, ahead of the stores, where nothing younger is in flight
        if constexpr (MODE == 1 && RING) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (live && h == 0) {
            float rad = head[0][1];
            if (P.noise) rad += P.noise[s];
            float* o = P.out + 5 * s;
            o[0] = P.raw ? head[0][0] : mask * fmaxf(rad, 0.0f);
            o[1] = P.raw ? head[0][1] : mask * head[0][0] + (1.0f - mask) * F.invalid_sdf;
            o[2] = rgb[0][0];
            o[3] = rgb[0][1];
            o[4] = rgb[0][2];
            if (P.valid) P.valid[s] = mask > 0.0f;
        }
        STAMP(10); // store
    }
    if (P.short_groups && (MODE == 1 ? lane_id_fresh() : lane) == 0 && short_groups) atomicAdd(P.short_groups, (unsigned long long)short_groups);
#ifdef VANERF_STAMPS
    unsigned long long rt1;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(rt1)::"memory");
    phase_cycles[N_PHASES - 1] = rt1 - rt0; // 100 MHz reference clock over the same span: clock = sum(phases) / this * 100 MHz
    if (P.stamps && lane == 0)
        for (int k = 0; k < N_PHASES; ++k) P.stamps[wave * N_PHASES + k] = phase_cycles[k];
#endif
}

} // namespace

extern "C" int vanerf_query_samples(const VanerfWeights* w, const VanerfFrame* frame, const float* pts, const float* query_sdf,
                                    const uint8_t* query_vis, const int32_t* knn_idx, const float* noise, const int32_t* order, int raw, int64_t n,
                                    float* out, uint8_t* valid, void* queue_word, void* stream)
{
    return guarded([&] {
        if (n < 0) throw_error("vanerf_query_samples: n = %lld < 0", (long long)n);
        if (n == 0) return; // an empty batch is valid (and has null data pointers)
        if (!w || !w->dev || !frame || !pts || !query_sdf || !query_vis || !knn_idx || !out) throw_error("vanerf_query_samples: null argument");
        if (!queue_word || (reinterpret_cast<uintptr_t>(queue_word) & 7u)) throw_error("vanerf_query_samples: queue_word must be 8 bytes of device memory, 8-byte aligned");
        if ((n + 31) / 32 >= 0xffffff00LL) throw_error("vanerf_query_samples: n = %lld too large for one launch", (long long)n);
        const VanerfFrame& f = *frame;
        if (!f.geo0 || !f.geo1 || !f.tex || !f.img || !f.mask || !f.verts || !f.vfeat0 || !f.vfeat1 || !f.vfeat_tex || !f.vert_vis || !f.kpt_cam)
            throw_error("vanerf_query_samples: frame has a null pointer");
        if (f.h0 < 1 || f.w0 < 1 || f.h1 < 1 || f.w1 < 1 || f.ht < 1 || f.wt < 1 || f.hi < 1 || f.wi < 1)
            throw_error("vanerf_query_samples: feature-map sizes must be positive");
        QueryParams P;
        P.f = f; P.w = w->dev; P.wbytes = (unsigned)(w->n_floats * sizeof(float)); P.pts = pts; P.qsdf = query_sdf; P.qvis = query_vis; P.noise = noise; P.knn_in = knn_idx; P.order = order; P.raw = raw;
        P.n = n; P.out = out; P.valid = valid; P.stamps = nullptr; P.short_groups = w->stats;
        P.xs = nullptr; P.aux = nullptr; P.npad = 0;
        { const int ws[4] = {P.f.wi, P.f.wt, P.f.w0, P.f.w1}, hs[4] = {P.f.hi, P.f.ht, P.f.h0, P.f.h1};
          for (int i = 0; i < 4; ++i) { P.wm1[i] = (float)(ws[i] - 1); P.hm1[i] = (float)(hs[i] - 1); } }
       
        long long ngroups = (n + 31) / 32;
        const int wpb = w->mode == 1 ? WPB<1> : WPB<0>;
        long long blocks = (ngroups + wpb - 1) / wpb;
        int dev = 0, cus = 256;
        HIP_CHECK(hipGetDevice(&dev));
        HIP_CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
        // Two 4-wave blocks per CU = two waves per SIMD (256 VGPRs each).  fp32 MFMA and fp32 VALU do not overlap on gfx950
        // (tools/probe_mfma_valu.hip: every VALU instruction adds its issue cycles to the MFMA time), so the second wave does not
        // hide VALU work behind MFMAs; what it hides is load latency and the in-order issue gaps of its partner.
        int per_cu = w->mode == 1 ? VANERF_WAVES_PER_SIMD_B * 4 / WPB<1> : 2;
        if (const char* e = getenv("VANERF_BLOCKS_PER_CU")) per_cu = atoi(e) > 0 ? atoi(e) : 2; // experiment knob
        long long cap = (long long)cus * per_cu;
        if (blocks > cap) blocks = cap;
        P.queue = static_cast<unsigned*>(queue_word); // the caller's word: no launch shares a queue head with another (any number in flight, any streams)
        HIP_CHECK(hipMemsetAsync(P.queue, 0, 8, (hipStream_t)stream));
        if (w->mode == 1) {
            // the opt-in above 64 KB of dynamic LDS is per device: set on every call (cheap), as vanerf_mesh_query_accel does
            HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(query_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)DYN_LDS_BYTES));
            hipLaunchKernelGGL(query_kernel<1>, dim3((unsigned)blocks), dim3(64 * WPB<1>), DYN_LDS_BYTES, (hipStream_t)stream, P);
        } else hipLaunchKernelGGL(query_kernel<0>, dim3((unsigned)blocks), dim3(64 * WPB<0>), 0, (hipStream_t)stream, P);
        HIP_CHECK(hipGetLastError());
    });
}

// Training: the fp32 kernel in spill mode -- the same forward pass of N samples (raw outputs [sdf_pred, rad, r, g, b]),
// which also writes every layer's operands and a few auxiliary values for the fused backward chain (query_backward.hip); layouts in layer_spec.h.
extern "C" int vanerf_query_forward_spill(const VanerfWeights* w, const VanerfFrame* frame, const float* pts, const float* query_sdf,
                                          const uint8_t* query_vis, const int32_t* knn_idx, int64_t n, int64_t npad, float* out_raw, uint8_t* valid,
                                          float* xs, float* aux, void* queue_word, void* stream)
{
    return guarded([&] {
        if (n <= 0) throw_error("vanerf_query_forward_spill: n = %lld", (long long)n);
        if (!w || !w->dev || w->mode != 0) throw_error("vanerf_query_forward_spill: needs an fp32 weight handle (vanerf_weights_pack mode 0)");
        if (!frame || !pts || !query_sdf || !query_vis || !knn_idx || !out_raw || !xs || !aux) throw_error("vanerf_query_forward_spill: null argument");
        if (!queue_word || (reinterpret_cast<uintptr_t>(queue_word) & 7u)) throw_error("vanerf_query_forward_spill: queue_word must be 8 bytes of device memory, 8-byte aligned");
        if (npad < n || npad % 32 != 0) throw_error("vanerf_query_forward_spill: npad = %lld must be a multiple of 32 and >= n = %lld", (long long)npad, (long long)n);
        if ((long long)X_ROWS * 4 * npad >= (long long)SPILL_OFF)
            throw_error("vanerf_query_forward_spill: a block of %lld samples spills more than 4 GB (the kernels address the spills with 32-bit offsets): at most %lld",
                        (long long)npad, (long long)SPILL_OFF / (4 * X_ROWS) / 32 * 32);
        const VanerfFrame& f = *frame;
        if (!f.geo0 || !f.geo1 || !f.tex || !f.img || !f.mask || !f.verts || !f.vfeat0 || !f.vfeat1 || !f.vfeat_tex || !f.vert_vis || !f.kpt_cam)
            throw_error("vanerf_query_forward_spill: frame has a null pointer");
        QueryParams P;
        P.f = f; P.w = w->dev; P.wbytes = (unsigned)(w->n_floats * sizeof(float)); P.pts = pts; P.qsdf = query_sdf; P.qvis = query_vis; P.noise = nullptr; P.knn_in = knn_idx; P.order = nullptr; P.raw = 1;
        P.n = n; P.out = out_raw; P.valid = valid; P.stamps = nullptr; P.short_groups = nullptr;
        P.xs = xs; P.aux = aux; P.npad = npad;
        { const int ws[4] = {P.f.wi, P.f.wt, P.f.w0, P.f.w1}, hs[4] = {P.f.hi, P.f.ht, P.f.h0, P.f.h1};
          for (int i = 0; i < 4; ++i) { P.wm1[i] = (float)(ws[i] - 1); P.hm1[i] = (float)(hs[i] - 1); } }
        const long long ngroups = (n + 31) / 32;
        long long blocks = (ngroups + WPB<0> - 1) / WPB<0>;
        int dev = 0, cus = 256;
        HIP_CHECK(hipGetDevice(&dev));
        HIP_CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
        if (blocks > (long long)cus * VANERF_WAVES_PER_SIMD_SPILL) blocks = (long long)cus * VANERF_WAVES_PER_SIMD_SPILL; // one 4-wave block per CU (the spill build takes the whole register file: at 256 registers it goes to scratch)
        P.queue = static_cast<unsigned*>(queue_word);
        HIP_CHECK(hipMemsetAsync(P.queue, 0, 8, (hipStream_t)stream));
        hipLaunchKernelGGL((query_kernel<0, true>), dim3((unsigned)blocks), dim3(64 * WPB<0>), 0, (hipStream_t)stream, P);
        HIP_CHECK(hipGetLastError());
    });
}

// ---------------------------------------------------------------------------------------------
// Validity partition.  A 32-sample group whose samples ALL miss the source view takes the short path of query_kernel; groups are
// consecutive samples of a ray, and on the benchmark view 12.5 % of them are mixed: their invalid samples (7 % of all samples) ride
// through the long path.  vanerf_query_order computes the validity of every sample with the kernel's own project_and_mask and writes the
// stable partition [valid samples in order | invalid samples in order]; query_kernel then works on slot k = sample order[k], so only one
// group per launch is mixed.  Results are the same bits (the networks are per-sample functions; outputs are stored at the sample's place).
// Three small kernels: flags + per-block counts, scan of the block counts, scatter.
// ---------------------------------------------------------------------------------------------
constexpr int CP_BLOCK = 1024;

__global__ __launch_bounds__(CP_BLOCK) void order_count_kernel(const VanerfFrame F, float wm1, float hm1, const float* __restrict__ pts, long long n,
                                                               uint8_t* __restrict__ flags, unsigned* __restrict__ block_valid)
{
    const long long s = (long long)blockIdx.x * CP_BLOCK + threadIdx.x;
    int valid = 0;
    if (s < n) {
        valid = project_and_mask(F, wm1, hm1, pts[3 * s], pts[3 * s + 1], pts[3 * s + 2]).mask > 0.0f;
        flags[s] = (uint8_t)valid;
    }
    const int cnt = __syncthreads_count(valid);
    if (threadIdx.x == 0) block_valid[blockIdx.x] = (unsigned)cnt;
}

// exclusive scan of block_valid[nblocks] in place; total -> block_valid[nblocks]
__global__ __launch_bounds__(CP_BLOCK) void order_scan_kernel(unsigned* __restrict__ block_valid, int nblocks)
{
    __shared__ unsigned s_sum[CP_BLOCK];
    const int per = (nblocks + CP_BLOCK - 1) / CP_BLOCK;
    const int b0 = threadIdx.x * per, b1 = min(b0 + per, nblocks);
    unsigned loc = 0;
    for (int b = b0; b < b1; ++b) loc += block_valid[b];
    s_sum[threadIdx.x] = loc;
    __syncthreads();
    for (int d = 1; d < CP_BLOCK; d <<= 1) { // Hillis-Steele inclusive scan
        const unsigned v = threadIdx.x >= (unsigned)d ? s_sum[threadIdx.x - d] : 0u;
        __syncthreads();
        s_sum[threadIdx.x] += v;
        __syncthreads();
    }
    unsigned run = s_sum[threadIdx.x] - loc; // exclusive prefix of this thread's chunk
    for (int b = b0; b < b1; ++b) { const unsigned c = block_valid[b]; block_valid[b] = run; run += c; }
    if (threadIdx.x == CP_BLOCK - 1) block_valid[nblocks] = s_sum[CP_BLOCK - 1];
}

__global__ __launch_bounds__(CP_BLOCK) void order_scatter_kernel(const uint8_t* __restrict__ flags, long long n, const unsigned* __restrict__ block_off,
                                                                 int nblocks, int32_t* __restrict__ order)
{
    __shared__ unsigned s_wave[CP_BLOCK / 64];
    const long long s = (long long)blockIdx.x * CP_BLOCK + threadIdx.x;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const bool live = s < n;
    const bool valid = live && flags[s] != 0;
    const unsigned long long m = __ballot(valid);
    const unsigned below = (unsigned)__builtin_popcountll(m & ((1ull << lane) - 1ull));
    if (lane == 0) s_wave[wv] = (unsigned)__builtin_popcountll(m);
    __syncthreads();
    unsigned wbase = 0;
    for (int k = 0; k < wv; ++k) wbase += s_wave[k];
    if (!live) return;
    const unsigned rank_valid = wbase + below;                       // valid samples of this block before this one
    const unsigned off_valid = block_off[blockIdx.x], total_valid = block_off[nblocks];
    const unsigned long long before = (unsigned long long)blockIdx.x * CP_BLOCK; // samples before this block
    const unsigned long long pos = valid ? (unsigned long long)off_valid + rank_valid
                                         : (unsigned long long)total_valid + (before - off_valid) + (threadIdx.x - rank_valid);
    order[pos] = (int32_t)s;
}

extern "C" int vanerf_query_order(const VanerfFrame* frame, const float* pts, int64_t n, int32_t* order, void* scratch, int64_t scratch_bytes,
                                  void* stream)
{
    return guarded([&] {
        if (n < 0) throw_error("vanerf_query_order: n = %lld < 0", (long long)n);
        if (n == 0) return;
        if (!frame || !pts || !order || !scratch) throw_error("vanerf_query_order: null argument");
        if (n >= 0x7fffffffLL) throw_error("vanerf_query_order: n = %lld does not fit a 32-bit index", (long long)n);
        const VanerfFrame& f = *frame;
        if (!f.mask || f.hi < 1 || f.wi < 1) throw_error("vanerf_query_order: frame has no mask");
        const int nblocks = (int)((n + CP_BLOCK - 1) / CP_BLOCK);
        const int64_t need = n + 4 * ((int64_t)nblocks + 1) + 16;
        if (scratch_bytes < need) throw_error("vanerf_query_order: scratch of %lld bytes, %lld needed (vanerf_query_order_scratch)", (long long)scratch_bytes, (long long)need);
        uint8_t* flags = static_cast<uint8_t*>(scratch);
        unsigned* block_off = reinterpret_cast<unsigned*>(flags + ((n + 15) / 16) * 16);
        const hipStream_t st = (hipStream_t)stream;
        hipLaunchKernelGGL(order_count_kernel, dim3(nblocks), dim3(CP_BLOCK), 0, st, f, (float)(f.wi - 1), (float)(f.hi - 1), pts, (long long)n, flags, block_off);
        hipLaunchKernelGGL(order_scan_kernel, dim3(1), dim3(CP_BLOCK), 0, st, block_off, nblocks);
        hipLaunchKernelGGL(order_scatter_kernel, dim3(nblocks), dim3(CP_BLOCK), 0, st, flags, (long long)n, block_off, nblocks, order);
        HIP_CHECK(hipGetLastError());
    });
}

extern "C" int64_t vanerf_query_order_scratch(int64_t n) { return n + 4 * ((n + CP_BLOCK - 1) / CP_BLOCK + 1) + 32; }

#ifdef VANERF_STAMPS
// Diagnostic build: same launch with a per-wave phase-cycle table [waves][12] (device pointer, zero-initialised by the caller).
extern "C" int vanerf_debug_query_stamps(const VanerfWeights* w, const VanerfFrame* frame, const float* pts, const float* query_sdf,
                                         const uint8_t* query_vis, const int32_t* knn_idx, int64_t n, float* out, unsigned long long* stamps, int* n_waves, void* queue_word, void* stream)
{
    return guarded([&] {
        QueryParams P;
        P.f = *frame; P.w = w->dev; P.wbytes = (unsigned)(w->n_floats * sizeof(float)); P.pts = pts; P.qsdf = query_sdf; P.qvis = query_vis; P.noise = nullptr; P.knn_in = knn_idx; P.order = nullptr; P.raw = 0;
        P.n = n; P.out = out; P.valid = nullptr; P.stamps = stamps; P.short_groups = nullptr;
        { const int ws[4] = {P.f.wi, P.f.wt, P.f.w0, P.f.w1}, hs[4] = {P.f.hi, P.f.ht, P.f.h0, P.f.h1};
          for (int i = 0; i < 4; ++i) { P.wm1[i] = (float)(ws[i] - 1); P.hm1[i] = (float)(hs[i] - 1); } }
        long long ngroups = (n + 31) / 32;
        const int wpb = w->mode == 1 ? WPB<1> : WPB<0>;
        long long blocks = (ngroups + wpb - 1) / wpb;
        long long capd = w->mode == 1 ? 256LL * VANERF_WAVES_PER_SIMD_B * 4 / WPB<1> : 512;
        if (const char* e = getenv("VANERF_BLOCKS_PER_CU")) capd = 256LL * (atoi(e) > 0 ? atoi(e) : 2);
        if (blocks > capd) blocks = capd;
        *n_waves = (int)blocks * wpb;
        P.queue = static_cast<unsigned*>(queue_word);
        HIP_CHECK(hipMemsetAsync(P.queue, 0, 8, (hipStream_t)stream));
        if (stamps && w->mode == 1) HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(query_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)DYN_LDS_BYTES));
        if (stamps && w->mode == 1) hipLaunchKernelGGL(query_kernel<1>, dim3((unsigned)blocks), dim3(64 * WPB<1>), DYN_LDS_BYTES, (hipStream_t)stream, P);
        else if (stamps) hipLaunchKernelGGL(query_kernel<0>, dim3((unsigned)blocks), dim3(64 * WPB<0>), 0, (hipStream_t)stream, P);
        HIP_CHECK(hipGetLastError());
    });
}
#endif
