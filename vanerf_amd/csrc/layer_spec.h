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

} // namespace vanerf
