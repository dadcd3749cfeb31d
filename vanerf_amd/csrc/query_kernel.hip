// query_kernel.hip -- the per-sample hot kernel: VANeRF.query + eval_func for N samples
// (reference src/model.py:748-957, 1140-1160; networks src/networks.py:27-33, 75-106, 281-293;
// src/utils.py:136-151, 633-649, 744-779, 822-880; src/spatial.py:59-117).
//
// One wave = 32 samples.  Lane l works on sample j = l & 31; the two lanes j and j + 32 that share
// a sample split every K dimension between them (half h = l >> 5), see layer_spec.h.  All 20 dense
// layers run as v_mfma_f32_32x32x2_f32 chains whose accumulators stay in registers from the
// bilinear gathers to the final (alpha, sdf, rgb) store: activations never touch LDS or HBM.
// LDS holds only the mesh vertices (1-NN search) and the key points (positional encoding).
// Weights stream from L2 as pre-permuted MFMA A-fragments (weights_pack.cpp).
//
// Built with -ffp-contract=off: the integer-valued outputs (1-NN index) depend on fp32 compare
// results and must match oracle/mesh_oracle.c bit for bit; fused multiply-adds are spelled fmaf().
#include "common.h"

using namespace vanerf;

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int WAVES_PER_BLOCK = 4;
constexpr int BLOCK = 64 * WAVES_PER_BLOCK;

struct QueryParams {
    VanerfFrame f;
    const float* w;
    LayerOffsets offs;
    const float* pts;
    const float* qsdf;
    const uint8_t* qvis;
    const float* noise;
    long long n;
    float* out;
    uint8_t* valid;
    int32_t* knn_idx;
};

// ---------------------------------------------------------------------------------------------
// weight fragment stream with a two-deep register prefetch
// ---------------------------------------------------------------------------------------------
template <int NB> struct WFrag { float v[NB]; };

template <int NB> __device__ __forceinline__ WFrag<NB> wload(const float* p);
template <> __device__ __forceinline__ WFrag<1> wload<1>(const float* p) { return {{p[0]}}; }
template <> __device__ __forceinline__ WFrag<2> wload<2>(const float* p)
{
    float2 t = *reinterpret_cast<const float2*>(p);
    return {{t.x, t.y}};
}
template <> __device__ __forceinline__ WFrag<3> wload<3>(const float* p) { return {{p[0], p[1], p[2]}}; }
template <> __device__ __forceinline__ WFrag<4> wload<4>(const float* p)
{
    float4 t = *reinterpret_cast<const float4*>(p);
    return {{t.x, t.y, t.z, t.w}};
}

// Address = uniform base (SGPR pair, advanced with scalar adds) + per-lane element offset (one VGPR):
// `global_load ... v_off, s[base] offset:imm`.  A per-lane 64-bit pointer would cost two VGPRs per 4 KB window.
template <int NB> struct WStream {
    const float* sb; // wave-uniform
    unsigned voff;   // lane * NB
    WFrag<NB> f0, f1;
    __device__ __forceinline__ WStream(const float* base, int lane)
    {
        sb = base;
        voff = (unsigned)lane * NB;
        f0 = wload<NB>(sb + voff);
        f1 = wload<NB>(sb + 64 * NB + voff);
        sb += 2 * 64 * NB;
    }
    __device__ __forceinline__ WFrag<NB> next()
    {
        WFrag<NB> r = f0;
        f0 = f1;
        f1 = wload<NB>(sb + voff);
        sb += 64 * NB;
        return r;
    }
};

template <int NB> __device__ __forceinline__ void mma(f32x16 (&acc)[NB], WStream<NB>& s, float b)
{
    WFrag<NB> a = s.next();
#pragma unroll
    for (int ob = 0; ob < NB; ++ob) acc[ob] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.v[ob], b, acc[ob], 0, 0, 0);
#ifdef VANERF_PIN_KSTEPS // experiment knob: forbid scheduling across k-steps
    __builtin_amdgcn_sched_barrier(0);
#endif
}

// consume registers 0..NREGS-1 of a previous accumulator block as k-pairs
template <int NB, int NREGS> __device__ __forceinline__ void mma_regs(f32x16 (&acc)[NB], WStream<NB>& s, const f32x16& src)
{
#pragma unroll
    for (int r = 0; r < NREGS; ++r) mma<NB>(acc, s, src[r]);
}

template <int NB> __device__ __forceinline__ void zero(f32x16 (&acc)[NB])
{
#pragma unroll
    for (int ob = 0; ob < NB; ++ob)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[ob][r] = 0.0f;
}

// ---------------------------------------------------------------------------------------------
// elementwise helpers
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float sigmoid_f(float x) { return 1.0f / (1.0f + __expf(-x)); }

// torch.nn.Softplus(beta=100, threshold=20) (src/utils.py:656): x if 100x > 20 else log1p(exp(100x))/100,
// evaluated as max(x,0) + log(1 + exp(-|100x|))/100 (identical value, |error| < 2e-9).
__device__ __forceinline__ float softplus100(float x)
{
    float t = x * 100.0f;
    float e = __expf(-fabsf(t));
    float r = fmaxf(x, 0.0f) + __logf(1.0f + e) * 0.01f;
    return t > 20.0f ? x : r;
}

template <int NB> __device__ __forceinline__ void relu(f32x16 (&a)[NB])
{
#pragma unroll
    for (int ob = 0; ob < NB; ++ob)
#pragma unroll
        for (int r = 0; r < 16; ++r) a[ob][r] = fmaxf(a[ob][r], 0.0f);
}

template <int NB> __device__ __forceinline__ void softplus(f32x16 (&a)[NB])
{
#pragma unroll
    for (int ob = 0; ob < NB; ++ob)
#pragma unroll
        for (int r = 0; r < 16; ++r) a[ob][r] = softplus100(a[ob][r]);
}

// ---------------------------------------------------------------------------------------------
// grid_sample(bilinear, border, align_corners=True) (src/utils.py:136-151)
// ---------------------------------------------------------------------------------------------
struct Bilin {
    int o00, o01, o10, o11; // pixel offsets y*W + x of the nw, ne, sw, se taps
    float w00, w01, w10, w11;
};

__device__ __forceinline__ Bilin bilin_setup(float x, float y, int H, int W)
{
    float ix = ((x + 1.0f) / 2.0f) * (float)(W - 1);
    float iy = ((y + 1.0f) / 2.0f) * (float)(H - 1);
    ix = fminf((float)(W - 1), fmaxf(ix, 0.0f));
    iy = fminf((float)(H - 1), fmaxf(iy, 0.0f));
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

// C4 = number of float4 groups to gather from a channel-last map with `C` channels, starting at channel coff
template <int C4> __device__ __forceinline__ void gather(const float* map, const Bilin& b, int C, int coff, float (&dst)[C4 * 4])
{
    const float4* p00 = reinterpret_cast<const float4*>(map + (size_t)b.o00 * C + coff);
    const float4* p01 = reinterpret_cast<const float4*>(map + (size_t)b.o01 * C + coff);
    const float4* p10 = reinterpret_cast<const float4*>(map + (size_t)b.o10 * C + coff);
    const float4* p11 = reinterpret_cast<const float4*>(map + (size_t)b.o11 * C + coff);
#pragma unroll
    for (int i = 0; i < C4; ++i) {
        float4 a = p00[i], c = p01[i], d = p10[i], e = p11[i];
        dst[4 * i + 0] = bilin_mix(b, a.x, c.x, d.x, e.x);
        dst[4 * i + 1] = bilin_mix(b, a.y, c.y, d.y, e.y);
        dst[4 * i + 2] = bilin_mix(b, a.z, c.z, d.z, e.z);
        dst[4 * i + 3] = bilin_mix(b, a.w, c.w, d.w, e.w);
    }
}

template <int C4> __device__ __forceinline__ void load_row(const float* row, float (&dst)[C4 * 4])
{
    const float4* p = reinterpret_cast<const float4*>(row);
#pragma unroll
    for (int i = 0; i < C4; ++i) {
        float4 a = p[i];
        dst[4 * i] = a.x; dst[4 * i + 1] = a.y; dst[4 * i + 2] = a.z; dst[4 * i + 3] = a.w;
    }
}

// one GeoVisFusion scale (src/networks.py:83-94 / 96-104): gates, gated 2-layer MLP.
//   HC = channels per lane half (32 for the 64-channel map, 4 for the 8-channel map), NBO = output blocks
template <int HC, int NBO, int NREG_MID>
__device__ __forceinline__ void geo_scale(const float* w, const LayerOffsets& offs, int l_at_a, int lane, int h,
                                          float (&pix)[HC], float (&nn)[HC], float (&tw)[HC], float s0, float s1,
                                          f32x16 (&outacc)[NBO])
{
    f32x16 at[1];
    zero<1>(at);
    {
        WStream<1> s(w + offs.off[l_at_a], lane);
#pragma unroll
        for (int t = 0; t < HC; ++t) mma<1>(at, s, pix[t]);
#pragma unroll
        for (int t = 0; t < HC; ++t) mma<1>(at, s, nn[t]);
#pragma unroll
        for (int t = 0; t < HC; ++t) mma<1>(at, s, tw[t]);
        mma<1>(at, s, s0);
        mma<1>(at, s, s1);
    }
    relu<1>(at);
    f32x16 gate[1];
    zero<1>(gate);
    {
        WStream<1> s(w + offs.off[l_at_a + 1], lane);
        mma_regs<1, 6>(gate, s, at[0]);
    }
    // gates live in rows 0..2 = registers 0..2 of the h = 0 lanes
    float a0 = __shfl(sigmoid_f(gate[0][0]), lane & 31);
    float a1 = __shfl(sigmoid_f(gate[0][1]), lane & 31);
    float a2 = __shfl(sigmoid_f(gate[0][2]), lane & 31);
#pragma unroll
    for (int t = 0; t < HC; ++t) { pix[t] *= a0; nn[t] *= a1; tw[t] *= a2; }
    f32x16 mid[NBO];
    zero<NBO>(mid);
    {
        WStream<NBO> s(w + offs.off[l_at_a + 2], lane);
#pragma unroll
        for (int t = 0; t < HC; ++t) mma<NBO>(mid, s, pix[t]);
#pragma unroll
        for (int t = 0; t < HC; ++t) mma<NBO>(mid, s, nn[t]);
#pragma unroll
        for (int t = 0; t < HC; ++t) mma<NBO>(mid, s, tw[t]);
        mma<NBO>(mid, s, s0);
        mma<NBO>(mid, s, s1);
    }
    relu<NBO>(mid);
    zero<NBO>(outacc);
    {
        WStream<NBO> s(w + offs.off[l_at_a + 3], lane);
#pragma unroll
        for (int ob = 0; ob < NBO; ++ob) mma_regs<NBO, NREG_MID>(outacc, s, mid[ob]);
    }
    (void)h;
}

__global__ __launch_bounds__(BLOCK, 2) void query_kernel(const QueryParams P)
{
    __shared__ float4 s_vert[VANERF_NV];
    __shared__ float4 s_kpt[VANERF_NKPT];
    for (int i = threadIdx.x; i < VANERF_NV; i += BLOCK) s_vert[i] = reinterpret_cast<const float4*>(P.f.verts)[i];
    for (int i = threadIdx.x; i < VANERF_NKPT; i += BLOCK) s_kpt[i] = reinterpret_cast<const float4*>(P.f.kpt_cam)[i];
    __syncthreads();

    const int lane = threadIdx.x & 63, j = lane & 31, h = lane >> 5;
    const long long wave = (long long)blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6);
    const long long nwaves = (long long)gridDim.x * WAVES_PER_BLOCK;
    const long long ngroups = (P.n + 31) / 32;
    const VanerfFrame& F = P.f;
    const float one_h0 = h ? 0.0f : 1.0f; // B operand of the bias k-step

    for (long long g = wave; g < ngroups; g += nwaves) {
        // Opaque per-iteration copy of the (uniform) weight base: keeps LICM from hoisting the ~160 4-KB-window
        // base addresses of the unrolled fragment stream out of the loop (they would all be live at once).
        const float* W = P.w;
        asm volatile("" : "+s"(W));
        const long long s_raw = g * 32 + j;
        const bool live = s_raw < P.n;
        const long long s = live ? s_raw : P.n - 1;
        const float px = P.pts[3 * s], py = P.pts[3 * s + 1], pz = P.pts[3 * s + 2];
        const float q_sdf = P.qsdf[s];
        const float q_vis = P.qvis[s] ? 1.0f : 0.0f;

        // ---- projection into the source view, validity mask, boundary weight (src/model.py:780-821) ----
        float vx = fmaf(pz, F.KRT[2], fmaf(py, F.KRT[1], px * F.KRT[0])) + F.KRT[3];
        float vy = fmaf(pz, F.KRT[6], fmaf(py, F.KRT[5], px * F.KRT[4])) + F.KRT[7];
        float vz = fmaf(pz, F.KRT[10], fmaf(py, F.KRT[9], px * F.KRT[8])) + F.KRT[11];
        float x = 2.0f * ((vx / vz) / (F.width - 1.0f)) - 1.0f;
        float y = 2.0f * ((vy / vz) / (F.height - 1.0f)) - 1.0f;
        float zn = 2.0f * (vz - F.znear) / (F.zfar - F.znear) - 1.0f;
        const float eps = 1e-2f;
        bool in_img = (x >= -1.0f - eps) && (x <= 1.0f + eps) && (y >= -1.0f - eps) && (y <= 1.0f + eps) && (zn >= -1.0f);
        const Bilin bi = bilin_setup(x, y, F.hi, F.wi);
        float fg = bilin_mix(bi, F.mask[bi.o00], F.mask[bi.o01], F.mask[bi.o10], F.mask[bi.o11]);
        const float mask = (in_img && fg > 0.1f) ? 1.0f : 0.0f;
        float pw;
        {
            float ux = 0.5f * x + 0.5f, uy = 0.5f * y + 0.5f, uz = 0.5f * zn + 0.5f;
            float dx = fminf(ux, 1.0f - ux), dy = fminf(uy, 1.0f - uy), dz = fminf(uz, 1.0f - uz);
            float wx = sigmoid_f(5.0f * (dx / 0.1f - 1.0f)), wy = sigmoid_f(5.0f * (dy / 0.1f - 1.0f)),
                  wz = sigmoid_f(5.0f * (dz / 0.1f - 1.0f));
            float p = (wx * wy) * wz * mask;
            pw = p / (p + 1e-6f);
        }

        // ---- 1-NN vertex (src/networks.py:27-33): lane half h scans hand h, first minimum wins ----------
        int nn_idx;
        {
            float best = INFINITY;
            int bi_ = 0;
            const float4* vv = s_vert + h * VANERF_NV_HAND;
#pragma unroll 4
            for (int i = 0; i < VANERF_NV_HAND; ++i) {
                float4 v = vv[i];
                float dx = px - v.x, dy = py - v.y, dz = pz - v.z;
                float d = (dx * dx + dy * dy) + dz * dz;
                if (d < best) { best = d; bi_ = i; }
            }
            bi_ += h * VANERF_NV_HAND;
            float od = __shfl_xor(best, 32);
            int oi = __shfl_xor(bi_, 32);
            nn_idx = (od < best || (od == best && oi < bi_)) ? oi : bi_;
        }
        const int tw_idx = nn_idx >= VANERF_NV_HAND ? nn_idx - VANERF_NV_HAND : nn_idx + VANERF_NV_HAND;
        const float vis_nn = F.vert_vis[nn_idx], vis_tw = F.vert_vis[tw_idx];
        const float sc0 = h ? q_vis : q_sdf;   // k-pair (sdf | qvis)
        const float sc1 = h ? vis_tw : vis_nn; // k-pair (vis_nn | vis_twin)

        // ---- GeoVisFusion (src/networks.py:75-106) ---------------------------------------------------------
        f32x16 g64[2], g8[1];
        {
            float pix[32], nn[32], tw[32];
            const Bilin b0 = bilin_setup(x, y, F.h0, F.w0);
            gather<8>(F.geo0, b0, 64, 32 * h, pix);
            load_row<8>(F.vfeat0 + (size_t)nn_idx * 64 + 32 * h, nn);
            load_row<8>(F.vfeat0 + (size_t)tw_idx * 64 + 32 * h, tw);
            geo_scale<32, 2, 16>(W, P.offs, L_GEO_AT0_A, lane, h, pix, nn, tw, sc0, sc1, g64);
        }
        {
            float pix[4], nn[4], tw[4];
            const Bilin b1 = bilin_setup(x, y, F.h1, F.w1);
            gather<1>(F.geo1, b1, 8, 4 * h, pix);
            load_row<1>(F.vfeat1 + (size_t)nn_idx * 8 + 4 * h, nn);
            load_row<1>(F.vfeat1 + (size_t)tw_idx * 8 + 4 * h, tw);
            geo_scale<4, 1, 4>(W, P.offs, L_GEO_AT1_A, lane, h, pix, nn, tw, sc0, sc1, g8);
        }

        // ---- mlp_geo.layers1 (src/utils.py:822-852): [PE294 | geo64] -> 128 -> 128 -> [. | geo8] -> 120 -> 64 ----
        f32x16 xv[2];
        {
            f32x16 a0[4];
            zero<4>(a0);
            {
                WStream<4> st(W + P.offs.off[L_MLP0], lane);
                // SpatialEncoder 'rel_z_decay' (src/spatial.py:71-72, 109-117): source-camera coordinates
                float cx = fmaf(pz, F.extrin[2], fmaf(py, F.extrin[1], px * F.extrin[0])) + F.extrin[3];
                float cy = fmaf(pz, F.extrin[6], fmaf(py, F.extrin[5], px * F.extrin[4])) + F.extrin[7];
                float cz = fmaf(pz, F.extrin[10], fmaf(py, F.extrin[9], px * F.extrin[8])) + F.extrin[11];
                const float4* kp = s_kpt + h * PE_KPT_PER_HALF;
                for (int i = 0; i < PE_KPT_PER_HALF; ++i) {
                    float4 k = kp[i];
                    float ddx = cx - k.x, ddy = cy - k.y, ddz = cz - k.z;
                    float d2 = (ddx * ddx + ddy * ddy) + ddz * ddz;
                    float wk = __expf(-d2 * F.pe_inv_2sigma2);
                    float dz = F.pe_scale * ddz;
                    float s0, c0;
                    sincosf(dz * 3.14159274f, &s0, &c0);
                    float s1 = 2.0f * s0 * c0, c1 = fmaf(-2.0f * s0, s0, 1.0f);
                    float s2 = 2.0f * s1 * c1, c2 = fmaf(-2.0f * s1, s1, 1.0f);
                    mma<4>(a0, st, dz * wk);
                    mma<4>(a0, st, s0 * wk);
                    mma<4>(a0, st, c0 * wk);
                    mma<4>(a0, st, s1 * wk);
                    mma<4>(a0, st, c1 * wk);
                    mma<4>(a0, st, s2 * wk);
                    mma<4>(a0, st, c2 * wk);
                }
                mma_regs<4, 16>(a0, st, g64[0]);
                mma_regs<4, 16>(a0, st, g64[1]);
                mma<4>(a0, st, one_h0);
            }
            softplus<4>(a0);
            f32x16 a1[4];
            zero<4>(a1);
            {
                WStream<4> st(W + P.offs.off[L_MLP1], lane);
#pragma unroll
                for (int b = 0; b < 4; ++b) mma_regs<4, 16>(a1, st, a0[b]);
                mma<4>(a1, st, one_h0);
            }
            softplus<4>(a1);
            zero<4>(a0);
            {
                WStream<4> st(W + P.offs.off[L_MLP2], lane);
#pragma unroll
                for (int b = 0; b < 4; ++b) mma_regs<4, 16>(a0, st, a1[b]);
                mma_regs<4, 4>(a0, st, g8[0]);
                mma<4>(a0, st, one_h0);
            }
            softplus<4>(a0);
            zero<2>(xv);
            {
                WStream<2> st(W + P.offs.off[L_MLP3], lane);
#pragma unroll
                for (int b = 0; b < 3; ++b) mma_regs<2, 16>(xv, st, a0[b]);
                mma_regs<2, 12>(xv, st, a0[3]);
                mma<2>(xv, st, one_h0);
            }
        }
        // ---- PoolModule mean/var over V = 1 views (src/utils.py:744-779, 854-880) --------------------------
        f32x16 mean[2], var[2];
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float m = pw * xv[b][r];
                float d = xv[b][r] - m;
                mean[b][r] = m;
                var[b][r] = pw * (d * d);
            }
        // ---- mlp_geo.layers2: 128 -> 64 -> 64 -> 2 (src/utils.py:709-719) ----------------------------------
        f32x16 head[1];
        {
            f32x16 m0[2], m1[2];
            zero<2>(m0);
            {
                WStream<2> st(W + P.offs.off[L_HEAD0], lane);
                mma_regs<2, 16>(m0, st, mean[0]); mma_regs<2, 16>(m0, st, mean[1]);
                mma_regs<2, 16>(m0, st, var[0]);  mma_regs<2, 16>(m0, st, var[1]);
                mma<2>(m0, st, one_h0);
            }
            softplus<2>(m0);
            zero<2>(m1);
            {
                WStream<2> st(W + P.offs.off[L_HEAD1], lane);
                mma_regs<2, 16>(m1, st, m0[0]); mma_regs<2, 16>(m1, st, m0[1]);
                mma<2>(m1, st, one_h0);
            }
            softplus<2>(m1);
            zero<1>(head);
            {
                WStream<1> st(W + P.offs.off[L_HEAD2], lane);
                mma_regs<1, 16>(head, st, m1[0]); mma_regs<1, 16>(head, st, m1[1]);
                mma<1>(head, st, one_h0);
            }
        }
        // ---- ibr_compress_gfeat 128 -> 24 (src/model.py:921) ------------------------------------------------
        f32x16 lat[1];
        zero<1>(lat);
        {
            WStream<1> st(W + P.offs.off[L_IBR], lane);
            mma_regs<1, 16>(lat, st, mean[0]); mma_regs<1, 16>(lat, st, mean[1]);
            mma_regs<1, 16>(lat, st, var[0]);  mma_regs<1, 16>(lat, st, var[1]);
            mma<1>(lat, st, one_h0);
        }
        // ---- TexVisFusion per-sample part (src/networks.py:281-293) -----------------------------------------
        f32x16 rgb[1];
        {
            float row[32]; // h = 0: nearest vertex [img3|tex8|gf18], h = 1: twin vertex
            load_row<8>(F.vfeat_tex + (size_t)(h ? tw_idx : nn_idx) * 32, row);
            float qi[4], qt[8];
            gather<1>(F.img, bi, 4, 0, qi);
            const Bilin bt = bilin_setup(x, y, F.ht, F.wt);
            gather<2>(F.tex, bt, 8, 0, qt);
            float q[6];
            q[0] = h ? qt[3] : qi[0]; q[1] = h ? qt[4] : qi[1]; q[2] = h ? qt[5] : qi[2];
            q[3] = h ? qt[6] : qt[0]; q[4] = h ? qt[7] : qt[1]; q[5] = h ? 0.0f : qt[2];
            const float t0 = h ? vis_nn : q_vis; // k-pair (qvis | vis_nn)
            const float t1 = h ? 0.0f : vis_tw;  // k-pair (vis_twin | -)
            f32x16 ta[3];
            zero<3>(ta);
            {
                WStream<3> st(W + P.offs.off[L_TEX_AT_A], lane);
#pragma unroll
                for (int t = 0; t < 29; ++t) mma<3>(ta, st, row[t]);
#pragma unroll
                for (int t = 0; t < 6; ++t) mma<3>(ta, st, q[t]);
                mma_regs<3, 12>(ta, st, lat[0]);
                mma<3>(ta, st, t0);
                mma<3>(ta, st, t1);
            }
            relu<3>(ta);
            f32x16 tg[1];
            zero<1>(tg);
            {
                WStream<1> st(W + P.offs.off[L_TEX_AT_B], lane);
#pragma unroll
                for (int b = 0; b < 3; ++b) mma_regs<1, 16>(tg, st, ta[b]);
            }
            // six gates: rows 0..3 -> h = 0 lanes regs 0..3, rows 4,5 -> h = 1 lanes regs 0,1
            float m0 = sigmoid_f(tg[0][0]), m1 = sigmoid_f(tg[0][1]), m2 = sigmoid_f(tg[0][2]), m3 = sigmoid_f(tg[0][3]);
            float o0 = __shfl_xor(m0, 32), o1 = __shfl_xor(m1, 32), o2 = __shfl_xor(m2, 32);
            const float gq = h ? o0 : m0;     // gate 0: query feature
            const float g11 = h ? o2 : m1;    // gate 1 (nearest) / gate 2 (twin): [img3|tex8]
            const float ggf = h ? m0 : m3;    // gate 3 (nearest) / gate 4 (twin): global feature
            const float glat = h ? m1 : o1;   // gate 5: compressed latent
#pragma unroll
            for (int t = 0; t < 11; ++t) row[t] *= g11;
#pragma unroll
            for (int t = 11; t < 29; ++t) row[t] *= ggf;
#pragma unroll
            for (int t = 0; t < 6; ++t) q[t] *= gq;
            f32x16 latg;
#pragma unroll
            for (int r = 0; r < 12; ++r) latg[r] = lat[0][r] * glat;
            zero<3>(ta);
            {
                WStream<3> st(W + P.offs.off[L_TEX_A], lane);
#pragma unroll
                for (int t = 0; t < 29; ++t) mma<3>(ta, st, row[t]);
#pragma unroll
                for (int t = 0; t < 6; ++t) mma<3>(ta, st, q[t]);
                mma_regs<3, 12>(ta, st, latg);
                mma<3>(ta, st, t0);
                mma<3>(ta, st, t1);
            }
            relu<3>(ta);
            zero<1>(rgb);
            {
                WStream<1> st(W + P.offs.off[L_TEX_B], lane);
#pragma unroll
                for (int b = 0; b < 3; ++b) mma_regs<1, 16>(rgb, st, ta[b]);
            }
        }
        // ---- eval_func (src/model.py:1140-1160): rows 0,1 of the head / 0..2 of the colour live in the h = 0 lanes ----
        if (live && h == 0) {
            float rad = head[0][1];
            if (P.noise) rad += P.noise[s];
            float* o = P.out + 5 * s;
            o[0] = mask * fmaxf(rad, 0.0f);
            o[1] = mask * head[0][0] + (1.0f - mask) * F.invalid_sdf;
            o[2] = rgb[0][0];
            o[3] = rgb[0][1];
            o[4] = rgb[0][2];
            if (P.valid) P.valid[s] = mask > 0.0f;
            if (P.knn_idx) P.knn_idx[s] = nn_idx;
        }
    }
}

} // namespace

extern "C" int vanerf_query_samples(const VanerfWeights* w, const VanerfFrame* frame, const float* pts, const float* query_sdf,
                                    const uint8_t* query_vis, const float* noise, int64_t n, float* out, uint8_t* valid,
                                    int32_t* knn_idx, void* stream)
{
    return guarded([&] {
        if (!w || !w->dev || !frame || !pts || !query_sdf || !query_vis || !out) throw_error("vanerf_query_samples: null argument");
        if (n < 0) throw_error("vanerf_query_samples: n = %lld < 0", (long long)n);
        if (n == 0) return;
        const VanerfFrame& f = *frame;
        if (!f.geo0 || !f.geo1 || !f.tex || !f.img || !f.mask || !f.verts || !f.vfeat0 || !f.vfeat1 || !f.vfeat_tex || !f.vert_vis || !f.kpt_cam)
            throw_error("vanerf_query_samples: frame has a null pointer");
        if (f.h0 < 1 || f.w0 < 1 || f.h1 < 1 || f.w1 < 1 || f.ht < 1 || f.wt < 1 || f.hi < 1 || f.wi < 1)
            throw_error("vanerf_query_samples: feature-map sizes must be positive");
        QueryParams P;
        P.f = f; P.w = w->dev; P.offs = w->offs; P.pts = pts; P.qsdf = query_sdf; P.qvis = query_vis; P.noise = noise;
        P.n = n; P.out = out; P.valid = valid; P.knn_idx = knn_idx;
        long long ngroups = (n + 31) / 32;
        long long blocks = (ngroups + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK;
        int dev = 0, cus = 256;
        HIP_CHECK(hipGetDevice(&dev));
        HIP_CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
        long long cap = (long long)cus * 2; // two 256-thread blocks per CU (launch bounds), persistent waves stride over groups
        if (blocks > cap) blocks = cap;
        hipLaunchKernelGGL(query_kernel, dim3((unsigned)blocks), dim3(BLOCK), 0, (hipStream_t)stream, P);
        HIP_CHECK(hipGetLastError());
    });
}
