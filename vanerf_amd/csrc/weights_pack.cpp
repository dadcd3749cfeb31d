// weights_pack.cpp -- folds weight-norm and permutes the reference's [out][in] weight matrices into
// the MFMA fragment streams described in layer_spec.h.  Host side; one hipMemcpy per pack.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstring>
#include <utility>
#include <vector>

#include "common.h"
#include "layer_spec.h"

using namespace vanerf;

namespace {

constexpr int BIAS = -2; // k index meaning "this lane half carries the bias, B operand = 1"
constexpr int ZERO = -1;

using Pairs = std::vector<std::pair<int, int>>;

inline int row0(int reg) { return (reg & 3) + 8 * (reg >> 2); }

// k-pairs that read the D registers of a previous layer with `nb` blocks (last block: `nregs_last` registers)
void chain(Pairs& p, int nb, int nregs_last, int base, int kin_limit)
{
    for (int ob = 0; ob < nb; ++ob) {
        int nr = (ob == nb - 1) ? nregs_last : 16;
        for (int r = 0; r < nr; ++r) {
            int k0 = ob * 32 + row0(r), k1 = k0 + 4;
            p.emplace_back(k0 < kin_limit ? base + k0 : ZERO, k1 < kin_limit ? base + k1 : ZERO);
        }
    }
}

void bias_pair(Pairs& p) { p.emplace_back(BIAS, ZERO); }

struct Mat {
    std::vector<float> w; // effective [nout][kin]
    std::vector<float> b; // [nout] or empty
    int nout, kin;
};

Mat plain(const float* w, const float* b, int nout, int kin)
{
    Mat m{std::vector<float>(w, w + (size_t)nout * kin), {}, nout, kin};
    if (b) m.b.assign(b, b + nout);
    return m;
}

// torch.nn.utils.weight_norm, dim=0: W[o][:] = g[o] * v[o][:] / ||v[o][:]||_2 (src/utils.py:674-675)
Mat weight_normed(const float* v, const float* g, const float* b, int nout, int kin)
{
    Mat m{std::vector<float>((size_t)nout * kin), std::vector<float>(b, b + nout), nout, kin};
    for (int o = 0; o < nout; ++o) {
        // torch: v * (g / norm(v)); norm accumulates in fp32 (vectorised); double here, error << 1 ulp of W
        double s = 0.0;
        for (int k = 0; k < kin; ++k) s += (double)v[(size_t)o * kin + k] * (double)v[(size_t)o * kin + k];
        float scale = g[o] / (float)std::sqrt(s);
        for (int k = 0; k < kin; ++k) m.w[(size_t)o * kin + k] = v[(size_t)o * kin + k] * scale;
    }
    return m;
}

void emit(std::vector<float>& out, int layer, const Mat& m, const Pairs& pairs)
{
    const int nb = kNB[layer], T = kT[layer];
    if ((int)pairs.size() != T) throw_error("internal: layer %d has %d k-pairs, spec says %d", layer, (int)pairs.size(), T);
    if (m.nout > nb * 32) throw_error("internal: layer %d: %d outputs do not fit %d blocks", layer, m.nout, nb);
    size_t base = out.size();
    out.resize(base + (size_t)T * 64 * nb, 0.0f);
    for (int t = 0; t < T; ++t)
        for (int lane = 0; lane < 64; ++lane) {
            int k = (lane >> 5) ? pairs[t].second : pairs[t].first;
            if (k == ZERO) continue;
            if (k != BIAS && (k < 0 || k >= m.kin)) throw_error("internal: layer %d k-index %d out of range", layer, k);
            for (int ob = 0; ob < nb; ++ob) {
                int o = ob * 32 + (lane & 31);
                if (o >= m.nout) continue;
                float v = (k == BIAS) ? (m.b.empty() ? 0.0f : m.b[o]) : m.w[(size_t)o * m.kin + k];
                out[base + ((size_t)t * 64 + lane) * nb + ob] = v;
            }
        }
}

inline unsigned short bf16_rne(float f)
{
    unsigned u;
    std::memcpy(&u, &f, 4);
    return (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}
inline float bf16_to_f32(unsigned short b)
{
    unsigned u = (unsigned)b << 16;
    float f;
    std::memcpy(&f, &u, 4);
    return f;
}

// bf16x3 stream of one layer (layer_spec.h): element j of the 8-element fragment of lane (r, h) at step s is k-pair 8s + j
void emit_b(std::vector<float>& out, int layer, const Mat& m, const Pairs& pairs)
{
    const int nb = kNB[layer], T = kT[layer], S = steps_b(layer);
    if ((int)pairs.size() != T) throw_error("internal: layer %d has %d k-pairs, spec says %d", layer, (int)pairs.size(), T);
    size_t base = out.size();
    out.resize(base + layer_dwords_b(layer), 0.0f);
    unsigned* dst = reinterpret_cast<unsigned*>(out.data() + base);
    for (int s = 0; s < S; ++s)
        for (int ob = 0; ob < nb; ++ob)
            for (int lane = 0; lane < 64; ++lane) {
                const int o = ob * 32 + (lane & 31);
                unsigned short hi[8] = {}, lo[8] = {};
                for (int j = 0; j < 8; ++j) {
                    const int t = 8 * s + j;
                    if (t >= T || o >= m.nout) continue;
                    const int k = (lane >> 5) ? pairs[t].second : pairs[t].first;
                    if (k == ZERO) continue;
                    const float v = (k == BIAS) ? (m.b.empty() ? 0.0f : m.b[o]) : m.w[(size_t)o * m.kin + k];
                    hi[j] = bf16_rne(v);
                    lo[j] = bf16_rne(v - bf16_to_f32(hi[j]));
                }
                for (int part = 0; part < 2; ++part) {
                    const unsigned short* e = part ? lo : hi;
                    unsigned* q = dst + ((((size_t)s * nb + ob) * 2 + part) * 64 + lane) * 4;
                    for (int i = 0; i < 4; ++i) q[i] = (unsigned)e[2 * i] | ((unsigned)e[2 * i + 1] << 16);
                }
            }
}

// [pix | nn | twin] each `c` channels split in halves between the lane halves, then (sdf|qvis), (vis_nn|vis_twin)
Pairs geo_input_pairs(int c)
{
    Pairs p;
    int hc = c / 2;
    for (int grp = 0; grp < 3; ++grp)
        for (int t = 0; t < hc; ++t) p.emplace_back(grp * c + t, grp * c + hc + t);
    p.emplace_back(3 * c + 0, 3 * c + 1);
    p.emplace_back(3 * c + 2, 3 * c + 3);
    return p;
}

// TexVisFusion input (src/networks.py:284-286): [q11 | nn11 | tw11 | nn_gf18 | tw_gf18 | latent24 | qvis | vis_nn | vis_tw]
Pairs tex_input_pairs()
{
    Pairs p;
    for (int t = 0; t < 11; ++t) p.emplace_back(11 + t, 22 + t);  // h0: nearest vertex row, h1: twin vertex row
    for (int t = 0; t < 18; ++t) p.emplace_back(33 + t, 51 + t);
    for (int u = 0; u < 6; ++u) p.emplace_back(u, u < 5 ? 6 + u : ZERO); // query feature: h0 q[0..5], h1 q[6..10]
    chain(p, 1, 12, 69, 24);                                        // latent24 straight from the ibr accumulator
    p.emplace_back(93, 94);
    p.emplace_back(95, ZERO);
    return p;
}

} // namespace

namespace {

// k-pairs (input channel of lane half 0 / 1 per k-step) of every layer: structure only, no weights
Pairs layer_pairs(int l)
{
    Pairs p;
    switch (l) {
    case L_GEO_AT0_A: case L_GEO_ATED0_A: return geo_input_pairs(64);
    case L_GEO_AT1_A: case L_GEO_ATED1_A: return geo_input_pairs(8);
    case L_GEO_AT0_B: case L_GEO_AT1_B: chain(p, 1, 6, 0, 10); return p;
    case L_GEO_ATED0_B: chain(p, 2, 16, 0, 64); return p;
    case L_GEO_ATED1_B: chain(p, 1, 4, 0, 8); return p;
    case L_MLP0:
        for (int i = 0; i < PE_KPT_PER_HALF; ++i)
            for (int f = 0; f < PE_FEATS; ++f) p.emplace_back(f * 42 + i, f * 42 + PE_KPT_PER_HALF + i);
        chain(p, 2, 16, 294, 64);
        bias_pair(p);
        return p;
    case L_MLP1: chain(p, 4, 16, 0, 128); bias_pair(p); return p;
    case L_MLP2: chain(p, 4, 16, 0, 128); chain(p, 1, 4, 128, 8); bias_pair(p); return p;
    case L_MLP3: chain(p, 4, 12, 0, 120); bias_pair(p); return p;
    case L_HEAD0: case L_IBR: chain(p, 2, 16, 0, 64); chain(p, 2, 16, 64, 64); bias_pair(p); return p; // [mean | var]
    case L_HEAD1: case L_HEAD2: chain(p, 2, 16, 0, 64); bias_pair(p); return p;
    case L_TEX_AT_A: case L_TEX_A: return tex_input_pairs();
    case L_TEX_AT_B: case L_TEX_B: chain(p, 3, 16, 0, 96); return p;
    }
    throw_error("internal: no layer %d", l);
}

// effective [out][in] matrix (+ bias) of every layer (weight-norm folded)
Mat layer_matrix(const VanerfWeightTable& w, int l)
{
    switch (l) {
    case L_GEO_AT0_A: return plain(w.geo_at0_w1, nullptr, 10, 196);
    case L_GEO_AT0_B: return plain(w.geo_at0_w2, nullptr, 3, 10);
    case L_GEO_ATED0_A: return plain(w.geo_ated0_w1, nullptr, 64, 196);
    case L_GEO_ATED0_B: return plain(w.geo_ated0_w2, nullptr, 64, 64);
    case L_GEO_AT1_A: return plain(w.geo_at1_w1, nullptr, 10, 28);
    case L_GEO_AT1_B: return plain(w.geo_at1_w2, nullptr, 3, 10);
    case L_GEO_ATED1_A: return plain(w.geo_ated1_w1, nullptr, 8, 28);
    case L_GEO_ATED1_B: return plain(w.geo_ated1_w2, nullptr, 8, 8);
    case L_MLP0: return weight_normed(w.l1_v[0], w.l1_g[0], w.l1_b[0], 128, 358);
    case L_MLP1: return weight_normed(w.l1_v[1], w.l1_g[1], w.l1_b[1], 128, 128);
    case L_MLP2: return weight_normed(w.l1_v[2], w.l1_g[2], w.l1_b[2], 120, 136);
    case L_MLP3: return plain(w.l1_w3, w.l1_b3, 64, 120);
    case L_HEAD0: return weight_normed(w.l2_v[0], w.l2_g[0], w.l2_b[0], 64, 128);
    case L_HEAD1: return weight_normed(w.l2_v[1], w.l2_g[1], w.l2_b[1], 64, 64);
    case L_HEAD2: return plain(w.l2_w2, w.l2_b2, 2, 64);
    case L_IBR: return plain(w.ibr_w, w.ibr_b, 24, 128);
    case L_TEX_AT_A: return plain(w.tex_at_w1, nullptr, 96, 96);
    case L_TEX_AT_B: return plain(w.tex_at_w2, nullptr, 6, 96);
    case L_TEX_A: return plain(w.tex_w1, nullptr, 96, 96);
    // IBRRenderingHead at V = 1 returns rgb_feat[..., :3] exactly (src/model.py:1613, 1635): rows 0..2 of 40
    case L_TEX_B: return plain(w.tex_w2, nullptr, 3, 96);
    }
    throw_error("internal: no layer %d", l);
}

// Backward stream of one layer (layer_spec.h): dX = W^T dY as the same MFMA chain with the roles swapped.  K dimension: the layer's output
// rows in D-register order (k-pair t'' = block t''/16, register t''%16: rows 32 ob + row0(r) and + 4); output rows: the layer's input
// SLOTS -- slot row 32 ib + jj of the result is k-pair t = 16 (kBLO + ib) + reg', lane half h' with jj = row0(reg') + 4 h', i.e. exactly
// where the previous layer's D registers sit, so the gradient chains from layer to layer in registers as the activations do forwards.
void emit_bwd(std::vector<float>& out, int layer, const Mat& m, const Pairs& pairs)
{
    const int T = kT[layer], BT = kBT[layer], nbt = kBNB[layer], blo = kBLO[layer];
    if (m.nout != kNOUT[layer]) throw_error("internal: layer %d has %d outputs, spec says %d", layer, m.nout, kNOUT[layer]);
    Pairs kk; // output rows per k-pair, D-register order
    chain(kk, (m.nout + 31) / 32, 16, 0, m.nout);
    for (int t = BT; t < (int)kk.size(); ++t)
        if (kk[t].first != ZERO || kk[t].second != ZERO) throw_error("internal: layer %d: output row beyond its %d backward k-pairs", layer, BT);
    const size_t base = out.size();
    if (base != bwd_layer_offset(layer)) throw_error("internal: backward stream of layer %d starts at %zu, spec says %u", layer, base, bwd_layer_offset(layer));
    out.resize(base + bwd_layer_floats(layer), 0.0f);
    for (int p = 0; 4 * p < nbt; ++p) {
        const int nb = bwd_pass_blocks(layer, p);
        float* dst = out.data() + bwd_pass_offset(layer, p);
        for (int t2 = 0; t2 < BT; ++t2)
            for (int lane = 0; lane < 64; ++lane) {
                const int o = (lane >> 5) ? kk[t2].second : kk[t2].first;
                if (o == ZERO) continue;
                const int jj = lane & 31, hs = (jj >> 2) & 1, reg = (jj & 3) + 4 * (jj >> 3);
                for (int ib = 0; ib < nb; ++ib) {
                    const int t = 16 * (blo + 4 * p + ib) + reg;
                    if (t >= T) continue;
                    const int c = hs ? pairs[t].second : pairs[t].first;
                    if (c < 0) continue; // ZERO / BIAS slots carry no input gradient
                    dst[((size_t)t2 * 64 + lane) * nb + ib] = m.w[(size_t)o * m.kin + c];
                }
            }
    }
}

} // namespace

namespace vanerf {

void pack_weights_host(const VanerfWeightTable& w, std::vector<float>& out, LayerOffsets& offs, int mode, std::vector<float>* bwd)
{
    out.clear();
    if (bwd) bwd->clear();
    for (int l = 0; l < NUM_LAYERS; ++l) {
        offs.off[l] = (unsigned)out.size();
        const unsigned want = mode ? layer_offset_b(l) : layer_offset(l);
        if (offs.off[l] != want) throw_error("internal: layer %d starts at %u, layer_spec.h says %u", l, offs.off[l], want);
        const Mat m = layer_matrix(w, l);
        const Pairs p = layer_pairs(l);
        mode ? emit_b(out, l, m, p) : emit(out, l, m, p);
        if (bwd) emit_bwd(*bwd, l, m, p);
    }
    // slack behind the last layer (prefetch rings never read past a layer's own steps, this is belt and braces)
    out.resize(out.size() + 2 * 64 * 4, 0.0f);
    if (bwd) bwd->resize(bwd->size() + 2 * 64 * 4, 0.0f);
}

// slot -> input channel of layer l (2 T entries, slot 2 t + h): >= 0 channel of the reference's [out][in] matrix, -1 unused, -2 bias
int layer_slots(int l, int* k_of_slot, int cap)
{
    if (l < 0 || l >= NUM_LAYERS) throw_error("vanerf_layer_slots: layer %d", l);
    const Pairs p = layer_pairs(l);
    if ((int)p.size() != kT[l]) throw_error("internal: layer %d has %d k-pairs, spec says %d", l, (int)p.size(), kT[l]);
    if (k_of_slot) {
        if (cap < 2 * kT[l]) throw_error("vanerf_layer_slots: room for %d slots, layer %d has %d", cap, l, 2 * kT[l]);
        for (int t = 0; t < kT[l]; ++t) { k_of_slot[2 * t] = p[t].first; k_of_slot[2 * t + 1] = p[t].second; }
    }
    return 2 * kT[l];
}

} // namespace vanerf
