// query_backward.hip -- training: the fused backward pass of the per-sample networks, the reverse of
// query_kernel.hip (reference: the autograd of VANeRF.query + query_color, src/model.py:748-957; networks src/networks.py:75-106, 281-293;
// src/utils.py:633-649, 709-719, 744-779, 822-880).
//
// Forward, D[out][sample] = W X keeps the sample on the lane and hands the accumulator registers of one layer to the next as its B
// operands.  Backward, dX = W^T dY is the same chain with the roles swapped: the K dimension runs over the layer's OUTPUT rows in
// D-register order, so the registers that hold dY (the gradient of a layer's accumulator) are the B operands as they stand, and the
// transposed fragment stream (weights_pack.cpp: emit_bwd) puts row (16 ib + reg', half h') of the result where input slot (t, h) of the
// layer sits -- which is where the PREVIOUS layer's accumulator register sits.  The gradient therefore walks the twenty layers in
// registers exactly as the activations did: v_mfma_f32_32x32x2_f32 chains, no LDS, no cross-lane traffic except the gate sums.
//
// What the chain needs from the forward pass comes from the spills the fp32 kernel writes in spill mode (vanerf_query_forward_spill):
// every layer's operands X (the value AFTER the previous activation: relu' = [X > 0], softplus' = 1 - exp(-100 X)), the gates, the pixel
// weight, the pre-pooling latent.  What it leaves behind for the host: every layer's dY (Ys), so that the weight gradients are ONE matrix
// product per layer over all samples, dW'[out][slot] = Ys_l Xs_l^T (vanerf_amd/hip_backward.py), and the gradients of the gathered inputs
// (pixel taps, nearest / twin vertex rows) for the scatter kernels (IGs).  Layouts: layer_spec.h.
#include "common.h"
#include "mfma_chain.h"

using namespace vanerf;
using namespace vanerf_chain;

namespace {

constexpr int BW_BLOCK = 256; // four waves, one per SIMD (the unrolled chain takes the 512-register budget)

struct BwdParams {
    const float* wb;        // backward fragment streams
    unsigned wbytes;
    // gradient with respect to eval_func's outputs [alpha, sdf, r, g, b] of every sample (src/model.py:1140-1160); d2 / noise2 (or NULL): the
    // second set of density-noise draws the coarse points carry inside the fine batch; raw / valid: what the forward spill pass returned
    const float *d, *d2, *noise, *noise2, *raw;
    const uint8_t* valid;
    long long n, npad;
    const float* xs;        // [X_ROWS][npad]
    const float* aux;       // [AUX_ROWS][npad]
    float* ys;              // [Y_ROWS][npad]
    float* ig;              // IG_ROWS floats per sample: row-major tensors one after the other (layer_spec.h)
};

constexpr int crow0(int reg) { return (reg & 3) + 8 * (reg >> 2); }

// pass P of layer L's backward: acc[blocks of the pass] = W^T dY over the layer's kBT k-pairs; dy(t'') = this lane's D register t''
template <int L, int P, class Op>
__device__ __forceinline__ void bwd_pass(f32x16 (&acc)[bwd_pass_blocks(L, P)], WRsrc rs, unsigned lane, Op&& dy)
{
    constexpr int NB = bwd_pass_blocks(L, P), T = kBT[L];
    zero<NB>(acc);
    auto ring = ring_start<NB, T>(rs, bwd_pass_offset(L, P), lane * NB * 4u);
    run_layer<NB, T>(acc, ring, rs, bwd_pass_offset(L, P), lane * NB * 4u, static_cast<Op&&>(dy));
}

// Spill addressing: buffer instructions, offset = the lane's column (VGPR) + a wave-uniform row offset (row x 4 npad, one s_mul) -- global
// accesses with 64-bit addresses took ~4 extra instructions each (13.1 k instructions per group).  A lane that must not store holds the
// column NO_COL: beyond the buffer's size, the hardware drops the store.
constexpr unsigned NO_COL = 0xfffffff0u;
struct Spills {
    WRsrc xs, aux, ys, ig;
    unsigned row4;            // bytes per row of the channel-major spills: 4 npad
    unsigned vx;              // xs / aux: 4 (col + h npad)
    unsigned vy, vy0;         // ys: 4 (col + 4 h npad); the same for the h = 0 lanes only
    unsigned col;             // the sample's column
};

// the layer's output gradient, row by row, into Ys (what the weight gradient's matrix product reads)
template <int L, class Op> __device__ __forceinline__ void store_dy(Op&& dy, const Spills& S)
{
    static_for<kBT[L]>([&](auto tc) {
        constexpr int t = decltype(tc)::value, row_a = 32 * (t / 16) + crow0(t % 16);
        if constexpr (row_a < kNOUT[L]) {
            const float v = dy(tc);
            // rows row_a (h = 0) and row_a + 4 (h = 1); the last rows of a layer exist in the h = 0 lanes only
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), S.ys, (row_a + 4 < kNOUT[L]) ? S.vy : S.vy0, (unsigned)(y_row_base(L) + row_a) * S.row4, 0);
        }
    });
}

__global__ __launch_bounds__(BW_BLOCK, 1) void query_backward_kernel(const BwdParams P)
{
    const int lane = threadIdx.x & 63, j = lane & 31, h = lane >> 5;
    const WRsrc W = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(P.wb), 0, P.wbytes, 0x00020000);
    const long long ngroups = P.npad / 32, nwaves = (long long)gridDim.x * (BW_BLOCK / 64);
    // the four spills as buffers of their exact sizes (the host checks that the largest stays below 4 GB)
    const WRsrc xs_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(P.xs), 0, (unsigned)(X_ROWS * 4ll * P.npad), 0x00020000);
    const WRsrc aux_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(P.aux), 0, (unsigned)(AUX_ROWS * 4ll * P.npad), 0x00020000);
    const WRsrc ys_rs = __builtin_amdgcn_make_buffer_rsrc(P.ys, 0, (unsigned)(Y_ROWS * 4ll * P.npad), 0x00020000);
    const WRsrc ig_rs = __builtin_amdgcn_make_buffer_rsrc(P.ig, 0, (unsigned)(IG_ROWS * 4ll * P.npad), 0x00020000);
    for (long long g = (long long)blockIdx.x * (BW_BLOCK / 64) + (threadIdx.x >> 6); g < ngroups; g += nwaves) {
        // the row stride, opaque per group: the ~3 000 row offsets (row x npad) of a group are loop invariants, and hoisted out of the loop they
        // cost 1 900-3 600 spilled SGPRs (v_writelane / v_readlane around every access; one build of this kernel restored wrong offsets in the
        // second and later groups of a wave) -- recomputed next to their use they are one s_mul each
        size_t npad = (size_t)P.npad;
        asm volatile("" : "+s"(npad));
        const size_t col = (size_t)g * 32 + (size_t)j;
        Spills S;
        S.xs = xs_rs; S.aux = aux_rs; S.ys = ys_rs; S.ig = ig_rs;
        S.row4 = (unsigned)npad * 4u;
        S.col = (unsigned)col;
        S.vx = 4u * (unsigned)col + (unsigned)h * S.row4;
        S.vy = 4u * (unsigned)col + (unsigned)(4 * h) * S.row4;
        S.vy0 = h ? NO_COL : S.vy;
        auto X = [&](auto lc, auto tc) -> float {
#ifdef VANERF_EXP_BWD_NOLOAD // timing experiment, wrong results: what the chain costs without its spill loads
            return 1.0f;
#endif
            return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(S.xs, S.vx, (unsigned)(x_row_base(decltype(lc)::value) + 2 * decltype(tc)::value) * S.row4, 0));
        };
        auto AUX = [&](auto kc) -> float {
#ifdef VANERF_EXP_BWD_NOLOAD
            return 0.5f;
#endif
            return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(S.aux, S.vx, (unsigned)(2 * decltype(kc)::value) * S.row4, 0));
        };
        // IG spill: tensor at `off` floats per sample (x npad: wave-uniform), rows of `stride` floats; this lane's sample, channel ch
        auto IG1 = [&](int off, int stride, int ch, float v) {
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), S.ig, 4u * (S.col * (unsigned)stride + (unsigned)ch), (unsigned)off * S.row4, 0);
        };
        auto IG4 = [&](int off, int stride, int ch, float a, float b, float c, float dd) {
            const u32x4 q = {__float_as_uint(a), __float_as_uint(b), __float_as_uint(c), __float_as_uint(dd)};
            // (the whole offset in the VGPR: a 16-byte buffer store with an SGPR offset reads its data late, and the compiler put a v_pk_mov_b32
            // into the data registers right behind one -- lanes 12..15 / 28..31 of one channel stored the NEXT value, differently from run to run)
            __builtin_amdgcn_raw_buffer_store_b128(q, S.ig, 4u * (S.col * (unsigned)stride + (unsigned)ch) + (unsigned)off * S.row4, 0, 0);
        };
#define LC(l) std::integral_constant<int, (l)>{}
        auto regs = [](auto& arr) { return [&arr](auto tc) -> float { constexpr int t = decltype(tc)::value; return arr[t / 16][t % 16]; }; };
        auto relu_g = [](float x, float v) { return x > 0.0f ? v : 0.0f; };
        auto sp_g = [](float x, float v) { return v * (1.0f - __expf(-100.0f * x)); }; // softplus(beta = 100)' from its output

        // gradient with respect to the raw outputs: rows 0, 1 of the head and 0..2 of the colour live in the h = 0 lanes
        // through eval_func: alpha = mask relu(rad + noise), sdf = mask sdf_pred + (1 - mask) const, colour untouched -> [sdf_pred, rad, r, g, b]
        float d_out[5] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
        if (h == 0 && (long long)col < P.n) {
            const float mask = P.valid[col] ? 1.0f : 0.0f, rad = P.raw[col * 5 + 1];
            float da = ((P.noise ? rad + P.noise[col] : rad) > 0.0f) ? P.d[col * 5] : 0.0f, ds = P.d[col * 5 + 1];
            d_out[2] = P.d[col * 5 + 2]; d_out[3] = P.d[col * 5 + 3]; d_out[4] = P.d[col * 5 + 4];
            if (P.d2) {
                if ((P.noise2 ? rad + P.noise2[col] : rad) > 0.0f) da += P.d2[col * 5];
                ds += P.d2[col * 5 + 1];
                d_out[2] += P.d2[col * 5 + 2]; d_out[3] += P.d2[col * 5 + 3]; d_out[4] += P.d2[col * 5 + 4];
            }
            d_out[0] = mask * ds; d_out[1] = mask * da;
        }

        // ================= colour branch: TexVisFusion (src/networks.py:281-293), ibr_compress (src/model.py:921) =================
        f32x16 d_in_tex[3]; // gradient of the (ungated) TexVisFusion input, slots t = 0..47
        {
            f32x16 dy19[1];
            zero<1>(dy19);
            dy19[0][0] = d_out[2]; dy19[0][1] = d_out[3]; dy19[0][2] = d_out[4];
            store_dy<L_TEX_B>(regs(dy19), S);
            f32x16 dx19[3];
            bwd_pass<L_TEX_B, 0>(dx19, W, (unsigned)lane, regs(dy19));
            f32x16 dy18[3];
            static_for<48>([&](auto tc) { constexpr int t = decltype(tc)::value; dy18[t / 16][t % 16] = relu_g(X(LC(L_TEX_B), tc), dx19[t / 16][t % 16]); });
            store_dy<L_TEX_A>(regs(dy18), S);
            f32x16 dx18[3]; // gradient of the GATED input
            bwd_pass<L_TEX_A, 0>(dx18, W, (unsigned)lane, regs(dy18));
            // gates (src/networks.py:287-290): products input x gate; the lane half's gates come from the auxiliary spill as it used them
            const float gq = AUX(LC(AUX_TEX)), g11 = AUX(LC(AUX_TEX + 1)), ggf = AUX(LC(AUX_TEX + 2)), glat = AUX(LC(AUX_TEX + 3));
            float s_q = 0.0f, s_11 = 0.0f, s_gf = 0.0f, s_lat = 0.0f; // sum of input x gradient per gate (this lane half's share)
            static_for<47>([&](auto tc) {
                constexpr int t = decltype(tc)::value;
                const float gv = dx18[t / 16][t % 16];
                float xin; // the ungated input
                if constexpr (t >= 35) xin = AUX(LC(AUX_LAT + (t - 35)));
                else xin = X(LC(L_TEX_AT_A), tc);
                const float gate = t < 11 ? g11 : t < 29 ? ggf : t < 35 ? gq : glat;
                (t < 11 ? s_11 : t < 29 ? s_gf : t < 35 ? s_q : s_lat) += xin * gv;
                d_in_tex[t / 16][t % 16] = gate * gv;
            });
            d_in_tex[2][15] = 0.0f; // slot 47: (qvis | vis_nn), no gradient
            s_q += __shfl_xor(s_q, 32);      // gate 0 and gate 5 are shared by the two lane halves
            s_lat += __shfl_xor(s_lat, 32);
            const float o_11 = __shfl_xor(s_11, 32);
            auto dsig = [](float g, float s) { return s * g * (1.0f - g); };
            // rows of fconv_at's output: 0 query, 1 nearest [img3|tex8], 2 twin, 3 nearest global, 4 twin global, 5 latent;
            // rows 0..3 sit in registers 0..3 of the h = 0 lanes, rows 4, 5 in registers 0, 1 of the h = 1 lanes
            f32x16 dy17[1];
            zero<1>(dy17);
            {
                const float g11_o = __shfl_xor(g11, 32);
                dy17[0][0] = h ? dsig(ggf, s_gf) : dsig(gq, s_q);
                dy17[0][1] = h ? dsig(glat, s_lat) : dsig(g11, s_11);
                dy17[0][2] = h ? 0.0f : dsig(g11_o, o_11);
                dy17[0][3] = h ? 0.0f : dsig(ggf, s_gf);
            }
            store_dy<L_TEX_AT_B>(regs(dy17), S);
            f32x16 dx17[3];
            bwd_pass<L_TEX_AT_B, 0>(dx17, W, (unsigned)lane, regs(dy17));
            f32x16 dy16[3];
            static_for<48>([&](auto tc) { constexpr int t = decltype(tc)::value; dy16[t / 16][t % 16] = relu_g(X(LC(L_TEX_AT_B), tc), dx17[t / 16][t % 16]); });
            store_dy<L_TEX_AT_A>(regs(dy16), S);
            f32x16 dx16[3];
            bwd_pass<L_TEX_AT_A, 0>(dx16, W, (unsigned)lane, regs(dy16));
            static_for<47>([&](auto tc) { constexpr int t = decltype(tc)::value; d_in_tex[t / 16][t % 16] += dx16[t / 16][t % 16]; });
            // vertex rows (t < 29) and query feature (29..34) go back to the host's scatters
            {
                // this lane half's vertex row (nearest: h = 0, twin: h = 1): channels = slots 0..28.  (The tensor's offset depends on h: it goes
                // into the lane part of the address, 32 npad floats further for the h = 1 lanes.)
                static_for<7>([&](auto qc) {
                    constexpr int t = 4 * decltype(qc)::value;
                    const u32x4 q = {__float_as_uint(d_in_tex[t / 16][t % 16]), __float_as_uint(d_in_tex[(t + 1) / 16][(t + 1) % 16]),
                                     __float_as_uint(d_in_tex[(t + 2) / 16][(t + 2) % 16]), __float_as_uint(d_in_tex[(t + 3) / 16][(t + 3) % 16])};
                    __builtin_amdgcn_raw_buffer_store_b128(q, S.ig, 4u * (S.col * 32u + (unsigned)t) + (unsigned)(IG_TEX + 32 * h) * S.row4, 0, 0);
                });
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(d_in_tex[1][12]), S.ig, 4u * (S.col * 32u + 28u) + (unsigned)h * 32u * S.row4, (unsigned)IG_TEX * S.row4, 0);
                // h = 0: slots 32..34 = tex 0..2 -> channels 0..2; h = 1: slots 29..33 = tex 3..7 -> channels 3..7.  Five stores from every lane, no
                // branch (the h = 0 lanes repeat their last one)
                static_for<5>([&](auto ic) {
                    constexpr int i = decltype(ic)::value, i0 = i < 2 ? i : 2;
                    const float v1 = d_in_tex[(29 + i) / 16][(29 + i) % 16], v0 = d_in_tex[2][i0];
                    IG1(IG_TEX_XY, 8, h ? 3 + i : i0, h ? v1 : v0);
                });
            }
        }
        f32x16 dpool[4]; // gradient of [mean64 | var64]
        {
            f32x16 dy15[1];
            zero<1>(dy15);
            static_for<12>([&](auto rc) { constexpr int r = decltype(rc)::value; dy15[0][r] = d_in_tex[(35 + r) / 16][(35 + r) % 16]; });
            store_dy<L_IBR>(regs(dy15), S);
            bwd_pass<L_IBR, 0>(dpool, W, (unsigned)lane, regs(dy15));
        }
        // ================= density head: mlp_geo.layers2 (src/utils.py:709-719) =================
        {
            f32x16 dy14[1];
            zero<1>(dy14);
            dy14[0][0] = d_out[0]; dy14[0][1] = d_out[1];
            store_dy<L_HEAD2>(regs(dy14), S);
            f32x16 dx14[2];
            bwd_pass<L_HEAD2, 0>(dx14, W, (unsigned)lane, regs(dy14));
            f32x16 dy13[2];
            static_for<32>([&](auto tc) { constexpr int t = decltype(tc)::value; dy13[t / 16][t % 16] = sp_g(X(LC(L_HEAD2), tc), dx14[t / 16][t % 16]); });
            store_dy<L_HEAD1>(regs(dy13), S);
            f32x16 dx13[2];
            bwd_pass<L_HEAD1, 0>(dx13, W, (unsigned)lane, regs(dy13));
            f32x16 dy12[2];
            static_for<32>([&](auto tc) { constexpr int t = decltype(tc)::value; dy12[t / 16][t % 16] = sp_g(X(LC(L_HEAD1), tc), dx13[t / 16][t % 16]); });
            store_dy<L_HEAD0>(regs(dy12), S);
            f32x16 dph[4];
            bwd_pass<L_HEAD0, 0>(dph, W, (unsigned)lane, regs(dy12));
#pragma unroll
            for (int b = 0; b < 4; ++b) dpool[b] += dph[b];
        }
        // ================= pooling over V = 1 views (src/utils.py:744-779): m = pw x, var = pw (x - m)^2 =================
        f32x16 dy11[2];
        {
            const float pw = AUX(LC(AUX_PW)), q1 = 1.0f - pw;
            static_for<32>([&](auto rc) {
                constexpr int r = decltype(rc)::value;
                const float xv = AUX(LC(AUX_XV + r));
                dy11[r / 16][r % 16] = pw * (dpool[r / 16][r % 16] + 2.0f * dpool[2 + r / 16][r % 16] * xv * q1 * q1);
            });
        }
        // ================= mlp_geo.layers1 (src/utils.py:822-852) =================
        f32x16 dg64[2];
        float dg8[4];
        {
            store_dy<L_MLP3>(regs(dy11), S);
            f32x16 dx11[4];
            bwd_pass<L_MLP3, 0>(dx11, W, (unsigned)lane, regs(dy11));
            f32x16 dy10[4];
            zero<4>(dy10);
            static_for<60>([&](auto tc) { constexpr int t = decltype(tc)::value; dy10[t / 16][t % 16] = sp_g(X(LC(L_MLP3), tc), dx11[t / 16][t % 16]); });
            store_dy<L_MLP2>(regs(dy10), S);
            f32x16 dx10a[4], dx10b[1];
            bwd_pass<L_MLP2, 0>(dx10a, W, (unsigned)lane, regs(dy10));
            bwd_pass<L_MLP2, 1>(dx10b, W, (unsigned)lane, regs(dy10));
#pragma unroll
            for (int r = 0; r < 4; ++r) dg8[r] = dx10b[0][r]; // slots 64..67: the 8-channel geometry feature, no activation in between
            f32x16 dy9[4];
            static_for<64>([&](auto tc) { constexpr int t = decltype(tc)::value; dy9[t / 16][t % 16] = sp_g(X(LC(L_MLP2), tc), dx10a[t / 16][t % 16]); });
            store_dy<L_MLP1>(regs(dy9), S);
            f32x16 dx9[4];
            bwd_pass<L_MLP1, 0>(dx9, W, (unsigned)lane, regs(dy9));
            f32x16 dy8[4];
            static_for<64>([&](auto tc) { constexpr int t = decltype(tc)::value; dy8[t / 16][t % 16] = sp_g(X(LC(L_MLP1), tc), dx9[t / 16][t % 16]); });
            store_dy<L_MLP0>(regs(dy8), S);
            f32x16 dx8[3]; // slot blocks 9..11 = k-pairs 144..191; the 64-channel geometry feature is k-pairs 147..178
            bwd_pass<L_MLP0, 0>(dx8, W, (unsigned)lane, regs(dy8));
            static_for<32>([&](auto uc) { constexpr int u = decltype(uc)::value, t = 147 + u - 144; dg64[u / 16][u % 16] = dx8[t / 16][t % 16]; });
        }
        // ================= GeoVisFusion (src/networks.py:75-106), both scales: gates, gated 2-layer MLP =================
        // HC channels per lane half and group; layers l_a (gate 1), l_a + 1 (gate 2), l_a + 2 (mid), l_a + 3 (out)
        auto geo_scale_bwd = [&](auto hc_c, auto la_c, auto aux_c, auto ig_c, auto& dy_out /* f32x16[NBO] */, auto nbo_c, auto nmid_c) {
            constexpr int HC = decltype(hc_c)::value, LA = decltype(la_c)::value, AG = decltype(aux_c)::value, IGB = decltype(ig_c)::value;
            constexpr int NBO = decltype(nbo_c)::value, NMID = decltype(nmid_c)::value; // blocks of the output / k-pairs of the hidden layer
            constexpr int NSB = kBNB[LA]; // slot blocks of the input
            store_dy<LA + 3>(regs(dy_out), S);
            f32x16 dx3[NBO];
            bwd_pass<LA + 3, 0>(dx3, W, (unsigned)lane, regs(dy_out));
            f32x16 dy2[NBO];
            zero<NBO>(dy2);
            static_for<NMID>([&](auto tc) { constexpr int t = decltype(tc)::value; dy2[t / 16][t % 16] = relu_g(X(LC(LA + 3), tc), dx3[t / 16][t % 16]); });
            store_dy<LA + 2>(regs(dy2), S);
            f32x16 dxg[NSB]; // gradient of the gated input
            {
                f32x16 pa[bwd_pass_blocks(LA + 2, 0)];
                bwd_pass<LA + 2, 0>(pa, W, (unsigned)lane, regs(dy2));
#pragma unroll
                for (int b = 0; b < bwd_pass_blocks(LA + 2, 0); ++b) dxg[b] = pa[b];
                if constexpr (NSB > 4) {
                    f32x16 pb[bwd_pass_blocks(LA + 2, 1)];
                    bwd_pass<LA + 2, 1>(pb, W, (unsigned)lane, regs(dy2));
#pragma unroll
                    for (int b = 0; b < bwd_pass_blocks(LA + 2, 1); ++b) dxg[4 + b] = pb[b];
                }
            }
            const float a0 = AUX(LC(AG)), a1 = AUX(LC(AG + 1)), a2 = AUX(LC(AG + 2));
            float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f;
            f32x16 d_in[NSB];
            zero<NSB>(d_in);
            static_for<3 * HC>([&](auto tc) {
                constexpr int t = decltype(tc)::value;
                const float gv = dxg[t / 16][t % 16], xin = X(LC(LA), tc);
                (t < HC ? s0 : t < 2 * HC ? s1 : s2) += xin * gv;
                d_in[t / 16][t % 16] = (t < HC ? a0 : t < 2 * HC ? a1 : a2) * gv;
            });
            s0 += __shfl_xor(s0, 32); s1 += __shfl_xor(s1, 32); s2 += __shfl_xor(s2, 32); // a gate multiplies the channels of both lane halves
            f32x16 dy1[1]; // fconv_at's second layer: rows 0..2 = registers 0..2 of the h = 0 lanes
            zero<1>(dy1);
            dy1[0][0] = h ? 0.0f : s0 * a0 * (1.0f - a0);
            dy1[0][1] = h ? 0.0f : s1 * a1 * (1.0f - a1);
            dy1[0][2] = h ? 0.0f : s2 * a2 * (1.0f - a2);
            store_dy<LA + 1>(regs(dy1), S);
            f32x16 dx1[1];
            bwd_pass<LA + 1, 0>(dx1, W, (unsigned)lane, regs(dy1));
            f32x16 dy0[1];
            zero<1>(dy0);
            static_for<6>([&](auto tc) { constexpr int t = decltype(tc)::value; dy0[0][t] = relu_g(X(LC(LA + 1), tc), dx1[0][t]); });
            store_dy<LA>(regs(dy0), S);
            {
                f32x16 pa[bwd_pass_blocks(LA, 0)];
                bwd_pass<LA, 0>(pa, W, (unsigned)lane, regs(dy0));
#pragma unroll
                for (int b = 0; b < bwd_pass_blocks(LA, 0); ++b) d_in[b] += pa[b];
                if constexpr (NSB > 4) {
                    f32x16 pb[bwd_pass_blocks(LA, 1)];
                    bwd_pass<LA, 1>(pb, W, (unsigned)lane, regs(dy0));
#pragma unroll
                    for (int b = 0; b < bwd_pass_blocks(LA, 1); ++b) d_in[4 + b] += pb[b];
                }
            }
            static_for<3 * HC / 4>([&](auto qc) { // group t / HC = pix | nearest | twin; channel HC h + t % HC: four consecutive channels per store
                constexpr int t = 4 * decltype(qc)::value;
                IG4(IGB + 2 * HC * (t / HC), 2 * HC, HC * h + t % HC, d_in[t / 16][t % 16], d_in[(t + 1) / 16][(t + 1) % 16], d_in[(t + 2) / 16][(t + 2) % 16],
                    d_in[(t + 3) / 16][(t + 3) % 16]);
            });
        };
        {
            f32x16 dy7[1];
            zero<1>(dy7);
#pragma unroll
            for (int r = 0; r < 4; ++r) dy7[0][r] = dg8[r];
            geo_scale_bwd(LC(4), LC(L_GEO_AT1_A), LC(AUX_G1), LC(IG_GEO1), dy7, LC(1), LC(4));
        }
        geo_scale_bwd(LC(32), LC(L_GEO_AT0_A), LC(AUX_G0), LC(IG_GEO0), dg64, LC(2), LC(32));
#undef LC
    }
}

} // namespace

// The backward chain for the n samples whose forward pass vanerf_query_forward_spill has just spilled (same stream): d[n][5] (+ d2, noise draws) in,
// Ys[Y_ROWS][npad] and the IG spill (IG_ROWS floats per sample, layer_spec.h) out (every column < npad is written).  Pure function of the spills and the weights.
extern "C" int vanerf_query_backward(const VanerfWeights* w, const float* d, const float* d2, const float* noise, const float* noise2,
                                     const float* raw, const uint8_t* valid, int64_t n, int64_t npad, const float* xs, const float* aux,
                                     float* ys, float* ig, void* stream)
{
    return guarded([&] {
        if (n <= 0) throw_error("vanerf_query_backward: n = %lld", (long long)n);
        if (!w || !w->dev_bwd || w->mode != 0) throw_error("vanerf_query_backward: needs an fp32 weight handle (vanerf_weights_pack mode 0)");
        if (!d || !raw || !valid || !xs || !aux || !ys || !ig) throw_error("vanerf_query_backward: null argument");
        if (npad < n || npad % 32 != 0) throw_error("vanerf_query_backward: npad = %lld must be a multiple of 32 and >= n = %lld", (long long)npad, (long long)n);
        if ((long long)X_ROWS * 4 * npad >= (long long)NO_COL) throw_error("vanerf_query_backward: a block of %lld samples spills more than 4 GB", (long long)npad);
        BwdParams P;
        P.wb = w->dev_bwd; P.wbytes = (unsigned)(w->n_floats_bwd * sizeof(float)); P.d = d; P.d2 = d2; P.noise = noise; P.noise2 = noise2; P.raw = raw; P.valid = valid; P.n = n; P.npad = npad;
        P.xs = xs; P.aux = aux; P.ys = ys; P.ig = ig;
        int dev = 0, cus = 256;
        HIP_CHECK(hipGetDevice(&dev));
        HIP_CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
        long long blocks = (npad / 32 + BW_BLOCK / 64 - 1) / (BW_BLOCK / 64);
        if (blocks > cus) blocks = cus; // one block per CU; the waves stride over the groups (uniform cost per group)
        hipLaunchKernelGGL(query_backward_kernel, dim3((unsigned)blocks), dim3(BW_BLOCK), 0, (hipStream_t)stream, P);
        HIP_CHECK(hipGetLastError());
    });
}

extern "C" int vanerf_ig_tensor(int which, int* offset, int* channels, int* stride)
{
    static const int tab[9][3] = {{IG_GEO0, 64, 64}, {IG_GEO0 + 64, 64, 64}, {IG_GEO0 + 128, 64, 64}, {IG_GEO1, 8, 8}, {IG_GEO1 + 8, 8, 8},
                                  {IG_GEO1 + 16, 8, 8}, {IG_TEX, 29, 32}, {IG_TEX + 32, 29, 32}, {IG_TEX_XY, 8, 8}};
    int rc = guarded([&] {
        if (which < 0 || which >= 9) throw_error("vanerf_ig_tensor: tensor %d of 9", which);
        if (offset) *offset = tab[which][0];
        if (channels) *channels = tab[which][1];
        if (stride) *stride = tab[which][2];
    });
    return rc < 0 ? rc : 9;
}

extern "C" int vanerf_spill_rows(int* x_rows, int* y_rows, int* aux_rows, int* ig_rows)
{
    return guarded([&] {
        if (x_rows) *x_rows = X_ROWS;
        if (y_rows) *y_rows = Y_ROWS;
        if (aux_rows) *aux_rows = AUX_ROWS;
        if (ig_rows) *ig_rows = IG_ROWS;
    });
}

// slot -> input channel of layer `layer` (2 T entries; >= 0 channel, -1 unused, -2 bias); k_of_slot may be NULL.  Returns 2 T (< 0: error).
extern "C" int vanerf_layer_slots(int layer, int32_t* k_of_slot, int cap)
{
    int n = 0;
    const int rc = guarded([&] { n = layer_slots(layer, k_of_slot, cap); });
    return rc == VANERF_OK ? n : rc;
}

extern "C" int vanerf_layer_rows(int layer, int* x_row, int* y_row, int* n_out)
{
    return guarded([&] {
        if (layer < 0 || layer >= NUM_LAYERS) throw_error("vanerf_layer_rows: layer %d", layer);
        if (x_row) *x_row = x_row_base(layer);
        if (y_row) *y_row = y_row_base(layer);
        if (n_out) *n_out = kNOUT[layer];
    });
}
