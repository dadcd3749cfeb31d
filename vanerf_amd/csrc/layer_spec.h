// layer_spec.h -- the one description of the per-sample network stack shared by the host-side
// weight packer (weights_pack.cpp) and the device kernel (query_kernel.hip).
//
// Every dense layer y = W x (+ b) of the reference's per-sample networks
//   GeoVisFusion   src/networks.py:43-106      MLPUNetFusion  src/utils.py:609-649, 781-880
//   ibr_compress   src/model.py:633-636, 921   TexVisFusion   src/networks.py:281-293
// runs as a chain of v_mfma_f32_32x32x2_f32 with the WEIGHTS as the A operand and the
// per-sample activations as the B operand: D[out][sample] = sum_k W[out][k] X[k][sample].
//   lane l of the wave:  sample j = l & 31,  half h = l >> 5
//   A operand (1 VGPR):  W[out = 32*ob + j][k = kpair[t][h]]
//   B operand (1 VGPR):  X[k = kpair[t][h]][sample j]
//   D (16 VGPRs):        out row = 32*ob + (reg & 3) + 8*(reg >> 2) + 4*h,  column = sample j
// A k-step t therefore contracts TWO input channels, one supplied by the h = 0 lanes and one by
// the h = 1 lanes.  Because D keeps the sample on the lane, the D registers of one layer ARE the
// B operands of the next one (after the activation function) with k-pair
//   (32*ob + row0(reg), 32*ob + row0(reg) + 4),  row0(reg) = (reg & 3) + 8*(reg >> 2)
// -- no LDS round trip, no cross-lane traffic.  The packer permutes the K columns accordingly.
//
// Packed stream of one layer with NB output blocks and T k-steps: float [T][64 lanes][NB].
#pragma once

namespace vanerf {

enum Layer {
    L_GEO_AT0_A = 0, L_GEO_AT0_B, L_GEO_ATED0_A, L_GEO_ATED0_B,
    L_GEO_AT1_A, L_GEO_AT1_B, L_GEO_ATED1_A, L_GEO_ATED1_B,
    L_MLP0, L_MLP1, L_MLP2, L_MLP3, L_HEAD0, L_HEAD1, L_HEAD2, L_IBR,
    L_TEX_AT_A, L_TEX_AT_B, L_TEX_A, L_TEX_B, NUM_LAYERS
};

// output blocks (32 rows each) and k-steps per layer
constexpr int kNB[NUM_LAYERS] = {1, 1, 2, 2, 1, 1, 1, 1, 4, 4, 4, 2, 2, 2, 1, 1, 3, 1, 3, 1};
constexpr int kT[NUM_LAYERS] = {
    98, 6, 98, 32,      // geo scale 0: [pix32|nn32|twin32|2 scalar pairs], 10 -> 3, same, 64 -> 64
    14, 6, 14, 4,       // geo scale 1: [pix4|nn4|twin4|2], 10 -> 3, same, 8 -> 8
    147 + 32 + 1,       // mlp0: 294 PE + 64 geo + bias
    64 + 1,             // mlp1
    64 + 4 + 1,         // mlp2: 128 + 8 geo + bias
    60 + 1,             // mlp3: 120 + bias
    64 + 1, 32 + 1, 32 + 1, // head: [mean64|var64] -> 64 -> 64 -> 2
    64 + 1,             // ibr_compress 128 -> 24
    49, 48, 49, 48      // tex: 96 (+2 pad) -> 96 -> 6 ; 96 -> 96 -> 3 (of 40)
};

constexpr int PE_KPT_PER_HALF = 21; // 42 key points split between the two lane halves
constexpr int PE_FEATS = 7;         // dz, sin/cos x 3 octaves (src/spatial.py:20-43)

struct LayerOffsets {
    unsigned off[NUM_LAYERS]; // float offset of each layer's fragment stream
};

constexpr unsigned layer_floats(int l) { return (unsigned)kT[l] * 64u * (unsigned)kNB[l]; }

// Layers are packed back to back in enum order, so every offset is a compile-time constant (no SGPRs spent on a table).
constexpr unsigned layer_offset(int l)
{
    unsigned o = 0;
    for (int i = 0; i < l; ++i) o += layer_floats(i);
    return o;
}

// ---- split-bf16 ("bf16x3") stream: W = W_hi + W_lo (two bf16), one k-step of v_mfma_f32_32x32x16_bf16 covers 8 consecutive
// k-pairs of the fp32 ordering (lane half h supplies its 8 channels), three products per step: W_hi X_hi + W_hi X_lo + W_lo X_hi.
// Layout of a layer: uint32 [S = ceil(T/8)][NB][2 (hi, lo)][64 lanes][4] -- each (step, block, part) is one coalesced 1 KB load.
constexpr int steps_b(int l) { return (kT[l] + 7) / 8; }
constexpr unsigned layer_dwords_b(int l) { return (unsigned)steps_b(l) * (unsigned)kNB[l] * 2u * 64u * 4u; }
constexpr unsigned layer_offset_b(int l)
{
    unsigned o = 0;
    for (int i = 0; i < l; ++i) o += layer_dwords_b(i);
    return o;
}

// ---- training: spills of the fused backward pass (query_backward.hip) ---------------------------------------
// The forward kernel in spill mode writes every layer's B operands X (the layer's inputs after the previous activation, slot (t, h) = k-pair t
// of lane half h) channel-major: Xs[x_row_base(l) + 2 t + h][sample]; the backward chain writes every layer's output gradient dY the same
// way: Ys[y_row_base(l) + out row][sample].  The weight gradient of layer l is then ONE matrix product over the samples,
// dW'[out][slot] = Ys_l Xs_l^T (a bias slot holds the operand 1, so its column is the bias gradient), mapped back to the reference's
// [out][in] layout through the slot -> input-channel table of weights_pack.cpp (vanerf_layer_slots).
constexpr int kNOUT[NUM_LAYERS] = {10, 3, 64, 64, 10, 3, 8, 8, 128, 128, 120, 64, 64, 64, 2, 24, 96, 6, 96, 3};
constexpr int x_row_base(int l) { int o = 0; for (int i = 0; i < l; ++i) o += 2 * kT[i]; return o; }
constexpr int y_row_base(int l) { int o = 0; for (int i = 0; i < l; ++i) o += kNOUT[i]; return o; }
constexpr int X_ROWS = x_row_base(NUM_LAYERS), Y_ROWS = y_row_base(NUM_LAYERS);
// k-pairs of a layer's OUTPUT rows in D-register order (block ob, register r <-> rows 32 ob + row0(r) and + 4): the K dimension of dX = W^T dY
constexpr int kBT[NUM_LAYERS] = {6, 3, 32, 32, 6, 3, 4, 4, 64, 64, 60, 32, 32, 32, 2, 12, 48, 4, 48, 3};
// input-slot blocks (16 k-pairs each) of a layer that need a gradient: first block, number of blocks
constexpr int kBLO[NUM_LAYERS] = {0, 0, 0, 0, 0, 0, 0, 0, 9, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
constexpr int kBNB[NUM_LAYERS] = {6, 1, 6, 2, 1, 1, 1, 1, 3, 4, 5, 4, 4, 2, 2, 4, 3, 3, 3, 3};
// Backward stream of layer l: passes of at most 4 slot blocks, pass p = float [kBT][64 lanes][nb_p] (A fragments of v_mfma_f32_32x32x2_f32:
// lane (jj, hh) holds W[out row of k-pair t'' half hh][channel of slot row 32 ib + jj])
constexpr int bwd_pass_blocks(int l, int p) { return kBNB[l] - 4 * p > 4 ? 4 : kBNB[l] - 4 * p; }
constexpr unsigned bwd_layer_floats(int l) { return (unsigned)kBT[l] * 64u * (unsigned)kBNB[l]; }
constexpr unsigned bwd_layer_offset(int l) { unsigned o = 0; for (int i = 0; i < l; ++i) o += bwd_layer_floats(i); return o; }
constexpr unsigned bwd_pass_offset(int l, int p) { return bwd_layer_offset(l) + (unsigned)kBT[l] * 64u * 4u * (unsigned)p; }
// auxiliary per-sample values of the forward pass the backward needs besides X: row 2 k + h holds lane half h's value of quantity k
enum AuxRow {
    AUX_G0 = 0,      // 3: gates of GeoVisFusion scale 0 (a0, a1, a2)
    AUX_G1 = 3,      // 3: scale 1
    AUX_TEX = 6,     // 4: TexVisFusion gates as the lane half uses them (gq, g11, ggf, glat)
    AUX_PW = 10,     // pixel weight
    AUX_XV = 11,     // 32: output of mlp_geo.layers1 before pooling, D registers [2 blocks][16]
    AUX_LAT = 43,    // 12: ibr_compress output (the latent before its gate)
    AUX_COUNT = 55
};
constexpr int AUX_ROWS = 2 * AUX_COUNT;
// input gradients handed back to the host (gathers' backward): rows 2 t + h of
// IG spill: the gradients of the gathered inputs as ROW-MAJOR tensors [npad][C], one after the other -- what the scatter kernels read as they
// stand (a lane holds 32 / 4 consecutive channels of its sample in consecutive registers: float4 stores).  Offsets in floats per sample
// (x npad = offset of the tensor in the spill):
constexpr int IG_GEO0 = 0;            // 3 x [npad][64]: d pix | d nearest vertex row | d twin vertex row of scale 0 (channel = 32 h + t)
constexpr int IG_GEO1 = 3 * 64;       // 3 x [npad][8]: the same of scale 1 (channel = 4 h + t)
constexpr int IG_TEX = IG_GEO1 + 24;  // 2 x [npad][32]: d vertex row [img3 | tex8 | global18 | 3 unused] of the nearest (h = 0) and the twin vertex (h = 1)
constexpr int IG_TEX_XY = IG_TEX + 64; // [npad][8]: d texture-map pixel feature (channels 0..2 from the h = 0 lanes' slots 32..34, 3..7 from h = 1's 29..33)
constexpr int IG_ROWS = IG_TEX_XY + 8; // floats per sample

} // namespace vanerf
