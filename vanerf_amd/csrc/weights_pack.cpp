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

namespace vanerf {

void pack_weights_host(const VanerfWeightTable& w, std::vector<float>& out, LayerOffsets& offs, int mode)
{
    out.clear();
    auto begin = [&](int l) {
        offs.off[l] = (unsigned)out.size();
        const unsigned want = mode ? layer_offset_b(l) : layer_offset(l);
        if (offs.off[l] != want) throw_error("internal: layer %d starts at %u, layer_spec.h says %u", l, offs.off[l], want);
    };
    auto put = [&](std::vector<float>& o, int l, const Mat& m, const Pairs& p) { mode ? emit_b(o, l, m, p) : emit(o, l, m, p); };

    // ---- GeoVisFusion (src/networks.py:75-106) ------------------------------------------------
    {
        Pairs in0 = geo_input_pairs(64), in1 = geo_input_pairs(8), b10, c2, c1;
        chain(b10, 1, 6, 0, 10);
        chain(c2, 2, 16, 0, 64);
        chain(c1, 1, 4, 0, 8);
        begin(L_GEO_AT0_A);   put(out, L_GEO_AT0_A, plain(w.geo_at0_w1, nullptr, 10, 196), in0);
        begin(L_GEO_AT0_B);   put(out, L_GEO_AT0_B, plain(w.geo_at0_w2, nullptr, 3, 10), b10);
        begin(L_GEO_ATED0_A); put(out, L_GEO_ATED0_A, plain(w.geo_ated0_w1, nullptr, 64, 196), in0);
        begin(L_GEO_ATED0_B); put(out, L_GEO_ATED0_B, plain(w.geo_ated0_w2, nullptr, 64, 64), c2);
        begin(L_GEO_AT1_A);   put(out, L_GEO_AT1_A, plain(w.geo_at1_w1, nullptr, 10, 28), in1);
        begin(L_GEO_AT1_B);   put(out, L_GEO_AT1_B, plain(w.geo_at1_w2, nullptr, 3, 10), b10);
        begin(L_GEO_ATED1_A); put(out, L_GEO_ATED1_A, plain(w.geo_ated1_w1, nullptr, 8, 28), in1);
        begin(L_GEO_ATED1_B); put(out, L_GEO_ATED1_B, plain(w.geo_ated1_w2, nullptr, 8, 8), c1);
    }
    // ---- MLPUNetFusion (src/utils.py:633-649, 822-852) ------------------------------------------
    {
        Pairs p0;
        for (int i = 0; i < PE_KPT_PER_HALF; ++i)
            for (int f = 0; f < PE_FEATS; ++f) p0.emplace_back(f * 42 + i, f * 42 + PE_KPT_PER_HALF + i);
        chain(p0, 2, 16, 294, 64);
        bias_pair(p0);
        begin(L_MLP0); put(out, L_MLP0, weight_normed(w.l1_v[0], w.l1_g[0], w.l1_b[0], 128, 358), p0);
        Pairs p1; chain(p1, 4, 16, 0, 128); bias_pair(p1);
        begin(L_MLP1); put(out, L_MLP1, weight_normed(w.l1_v[1], w.l1_g[1], w.l1_b[1], 128, 128), p1);
        Pairs p2; chain(p2, 4, 16, 0, 128); chain(p2, 1, 4, 128, 8); bias_pair(p2);
        begin(L_MLP2); put(out, L_MLP2, weight_normed(w.l1_v[2], w.l1_g[2], w.l1_b[2], 120, 136), p2);
        Pairs p3; chain(p3, 4, 12, 0, 120); bias_pair(p3);
        begin(L_MLP3); put(out, L_MLP3, plain(w.l1_w3, w.l1_b3, 64, 120), p3);
        Pairs pool; chain(pool, 2, 16, 0, 64); chain(pool, 2, 16, 64, 64); bias_pair(pool); // [mean | var]
        begin(L_HEAD0); put(out, L_HEAD0, weight_normed(w.l2_v[0], w.l2_g[0], w.l2_b[0], 64, 128), pool);
        Pairs h1; chain(h1, 2, 16, 0, 64); bias_pair(h1);
        begin(L_HEAD1); put(out, L_HEAD1, weight_normed(w.l2_v[1], w.l2_g[1], w.l2_b[1], 64, 64), h1);
        begin(L_HEAD2); put(out, L_HEAD2, plain(w.l2_w2, w.l2_b2, 2, 64), h1);
        begin(L_IBR);   put(out, L_IBR, plain(w.ibr_w, w.ibr_b, 24, 128), pool);
    }
    // ---- TexVisFusion per-sample part (src/networks.py:281-293) -----------------------------------
    {
        Pairs in = tex_input_pairs(), c3;
        chain(c3, 3, 16, 0, 96);
        begin(L_TEX_AT_A); put(out, L_TEX_AT_A, plain(w.tex_at_w1, nullptr, 96, 96), in);
        begin(L_TEX_AT_B); put(out, L_TEX_AT_B, plain(w.tex_at_w2, nullptr, 6, 96), c3);
        begin(L_TEX_A);    put(out, L_TEX_A, plain(w.tex_w1, nullptr, 96, 96), in);
        // IBRRenderingHead at V = 1 returns rgb_feat[..., :3] exactly (src/model.py:1613, 1635): rows 0..2 of 40
        begin(L_TEX_B);    put(out, L_TEX_B, plain(w.tex_w2, nullptr, 3, 96), c3);
    }
    // slack behind the last layer (prefetch rings never read past a layer's own steps, this is belt and braces)
    out.resize(out.size() + 2 * 64 * 4, 0.0f);
}

} // namespace vanerf
