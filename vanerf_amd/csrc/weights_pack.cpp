// weights_pack.cpp -- folds weight-norm and permutes the reference's [out][in] weight matrices into
// the MFMA fragment streams described in layer_spec.h.
// Two stages: (1) the EFFECTIVE weights, one flat fp32 array `eff` (every layer's [out][in] matrix with weight-norm folded, then its bias);
// (2) PLACEMENT, a pure gather out[i] = eff[id[i] - 1] (id 0 = the constant 0; the bf16x3 stream also splits the value into its two bf16
// parts).  The placement tables depend on layer_spec.h only, are built once per process and serve both the host packer
// (vanerf_weights_pack: one hipMemcpy per pack) and the device packer (weights_update.hip: vanerf_weights_update re-packs a handle in place
// from parameters that live on the device, two launches per stream, no synchronisation -- the training step's path).
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

// A layer's [out][in] matrix (+ bias) as the packer sees it.  The placement tables are built from matrices whose "values" are ids: 1 + the
// element's index in `eff` (exact in fp32: eff has ~150 k entries).
struct Mat {
    std::vector<float> w; // [nout][kin]
    std::vector<float> b; // [nout] or empty
    int nout, kin;
};

struct WordB { int id0, id1, part; }; // one 32-bit word of the bf16x3 stream: bf16 part `part` (0 high, 1 low) of eff[id0 - 1] | eff[id1 - 1] << 16

// shapes of the twenty layers: rows used, input channels, bias, weight-norm (the reference's modules: include/vanerf_hip.h, VanerfWeightTable)
struct Shape { int nout, kin, bias, wn; };
constexpr Shape kShape[NUM_LAYERS] = {
    {10, 196, 0, 0}, {3, 10, 0, 0}, {64, 196, 0, 0}, {64, 64, 0, 0}, {10, 28, 0, 0}, {3, 10, 0, 0}, {8, 28, 0, 0}, {8, 8, 0, 0},
    {128, 358, 1, 1}, {128, 128, 1, 1}, {120, 136, 1, 1}, {64, 120, 1, 0}, {64, 128, 1, 1}, {64, 64, 1, 1}, {2, 64, 1, 0}, {24, 128, 1, 0},
    {96, 96, 0, 0}, {6, 96, 0, 0}, {96, 96, 0, 0},
    {3, 96, 0, 0}, // IBRRenderingHead at V = 1 returns rgb_feat[..., :3] exactly (src/model.py:1613, 1635): rows 0..2 of 40
};
constexpr unsigned eff_offset(int l)
{
    unsigned o = 0;
    for (int i = 0; i < l; ++i) o += (unsigned)(kShape[i].nout * kShape[i].kin + (kShape[i].bias ? kShape[i].nout : 0));
    return o;
}
static_assert(eff_offset(NUM_LAYERS) < (1u << 24), "ids must be exact in fp32");

Mat id_matrix(int l)
{
    const Shape sh = kShape[l];
    if (sh.nout != kNOUT[l]) throw_error("internal: layer %d has %d outputs, spec says %d", l, sh.nout, kNOUT[l]);
    Mat m{std::vector<float>((size_t)sh.nout * sh.kin), {}, sh.nout, sh.kin};
    const unsigned base = eff_offset(l) + 1u;
    for (size_t i = 0; i < m.w.size(); ++i) m.w[i] = (float)(base + i);
    if (sh.bias) {
        m.b.resize(sh.nout);
        for (int o = 0; o < sh.nout; ++o) m.b[o] = (float)(base + m.w.size() + o);
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

// bf16x3 stream of one layer (layer_spec.h): element j of the 8-element fragment of lane (r, h) at step s is k-pair 8s + j.
// Placement only: per 32-bit word of the stream two ids (the elements in its low and high half) and which bf16 part of them it holds.
void emit_b_ids(std::vector<WordB>& out, int layer, const Mat& m, const Pairs& pairs)
{
    const int nb = kNB[layer], T = kT[layer], S = steps_b(layer);
    if ((int)pairs.size() != T) throw_error("internal: layer %d has %d k-pairs, spec says %d", layer, (int)pairs.size(), T);
    size_t base = out.size();
    out.resize(base + layer_dwords_b(layer), WordB{0, 0, 0});
    WordB* dst = out.data() + base;
    for (int s = 0; s < S; ++s)
        for (int ob = 0; ob < nb; ++ob)
            for (int lane = 0; lane < 64; ++lane) {
                const int o = ob * 32 + (lane & 31);
                int id[8] = {};
                for (int j = 0; j < 8; ++j) {
                    const int t = 8 * s + j;
                    if (t >= T || o >= m.nout) continue;
                    const int k = (lane >> 5) ? pairs[t].second : pairs[t].first;
                    if (k == ZERO) continue;
                    id[j] = (int)((k == BIAS) ? (m.b.empty() ? 0.0f : m.b[o]) : m.w[(size_t)o * m.kin + k]);
                }
                for (int part = 0; part < 2; ++part) {
                    WordB* q = dst + ((((size_t)s * nb + ob) * 2 + part) * 64 + lane) * 4;
                    for (int i = 0; i < 4; ++i) q[i] = WordB{id[2 * i], id[2 * i + 1], part};
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

// where every layer's parameters sit in a weight table (host or device pointers alike)
void sources(const VanerfWeightTable& w, vanerf::LayerSrc out[NUM_LAYERS])
{
    auto set = [&](int l, const float* ww, const float* g, const float* b) {
        const Shape sh = kShape[l];
        if (!ww || (sh.wn && !g) || (sh.bias && !b)) throw_error("weight table: a pointer of layer %d is null", l);
        out[l] = vanerf::LayerSrc{ww, sh.wn ? g : nullptr, sh.bias ? b : nullptr, sh.nout, sh.kin, eff_offset(l)};
    };
    set(L_GEO_AT0_A, w.geo_at0_w1, nullptr, nullptr);
    set(L_GEO_AT0_B, w.geo_at0_w2, nullptr, nullptr);
    set(L_GEO_ATED0_A, w.geo_ated0_w1, nullptr, nullptr);
    set(L_GEO_ATED0_B, w.geo_ated0_w2, nullptr, nullptr);
    set(L_GEO_AT1_A, w.geo_at1_w1, nullptr, nullptr);
    set(L_GEO_AT1_B, w.geo_at1_w2, nullptr, nullptr);
    set(L_GEO_ATED1_A, w.geo_ated1_w1, nullptr, nullptr);
    set(L_GEO_ATED1_B, w.geo_ated1_w2, nullptr, nullptr);
    set(L_MLP0, w.l1_v[0], w.l1_g[0], w.l1_b[0]);
    set(L_MLP1, w.l1_v[1], w.l1_g[1], w.l1_b[1]);
    set(L_MLP2, w.l1_v[2], w.l1_g[2], w.l1_b[2]);
    set(L_MLP3, w.l1_w3, nullptr, w.l1_b3);
    set(L_HEAD0, w.l2_v[0], w.l2_g[0], w.l2_b[0]);
    set(L_HEAD1, w.l2_v[1], w.l2_g[1], w.l2_b[1]);
    set(L_HEAD2, w.l2_w2, nullptr, w.l2_b2);
    set(L_IBR, w.ibr_w, nullptr, w.ibr_b);
    set(L_TEX_AT_A, w.tex_at_w1, nullptr, nullptr);
    set(L_TEX_AT_B, w.tex_at_w2, nullptr, nullptr);
    set(L_TEX_A, w.tex_w1, nullptr, nullptr);
    set(L_TEX_B, w.tex_w2, nullptr, nullptr);
}

// stage 1 on the host: eff from host pointers.  torch.nn.utils.weight_norm, dim=0: W[o][:] = g[o] * v[o][:] / ||v[o][:]||_2
// (src/utils.py:674-675); torch: v * (g / norm(v)), its norm accumulates in fp32 (vectorised); double here, error << 1 ulp of W.
// weights_update.hip's fold_kernel does the same arithmetic in the same order on the device.
void fold_host(const VanerfWeightTable& w, std::vector<float>& eff)
{
    vanerf::LayerSrc src[NUM_LAYERS];
    sources(w, src);
    eff.assign(eff_offset(NUM_LAYERS), 0.0f);
    for (int l = 0; l < NUM_LAYERS; ++l) {
        const vanerf::LayerSrc& s = src[l];
        float* e = eff.data() + s.eff;
        for (int o = 0; o < s.nout; ++o) {
            const float* v = s.w + (size_t)o * s.kin;
            if (s.g) {
                double sum = 0.0;
                for (int k = 0; k < s.kin; ++k) sum += (double)v[k] * (double)v[k];
                const float scale = s.g[o] / (float)std::sqrt(sum);
                for (int k = 0; k < s.kin; ++k) e[(size_t)o * s.kin + k] = v[k] * scale;
            } else {
                for (int k = 0; k < s.kin; ++k) e[(size_t)o * s.kin + k] = v[k];
            }
            if (s.b) e[(size_t)s.nout * s.kin + o] = s.b[o];
        }
    }
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
inline unsigned short bf16_part(float v, int part)
{
    const unsigned short hi = bf16_rne(v);
    return part ? bf16_rne(v - bf16_to_f32(hi)) : hi;
}

// The placement tables: weight independent, built once per process.
const PackTables& pack_tables()
{
    static const PackTables tables = [] {
        PackTables t;
        std::vector<float> fwd, bwd;
        std::vector<WordB> fwd_b;
        for (int l = 0; l < NUM_LAYERS; ++l) {
            t.offs.off[l] = (unsigned)fwd.size();
            if (fwd.size() != layer_offset(l)) throw_error("internal: layer %d starts at %zu, layer_spec.h says %u", l, fwd.size(), layer_offset(l));
            if (fwd_b.size() != layer_offset_b(l)) throw_error("internal: layer %d (bf16x3) starts at %zu, layer_spec.h says %u", l, fwd_b.size(), layer_offset_b(l));
            const Mat m = id_matrix(l);
            const Pairs p = layer_pairs(l);
            emit(fwd, l, m, p);
            emit_b_ids(fwd_b, l, m, p);
            emit_bwd(bwd, l, m, p);
        }
        // slack behind the last layer (prefetch rings never read past a layer's own steps, this is belt and braces)
        fwd.resize(fwd.size() + 2 * 64 * 4, 0.0f);
        bwd.resize(bwd.size() + 2 * 64 * 4, 0.0f);
        fwd_b.resize(fwd_b.size() + 2 * 64 * 4, WordB{0, 0, 0});
        t.fwd.assign(fwd.begin(), fwd.end());
        t.bwd.assign(bwd.begin(), bwd.end());
        t.fwd_b.resize(2 * fwd_b.size());
        for (size_t i = 0; i < fwd_b.size(); ++i) {
            t.fwd_b[2 * i] = fwd_b[i].id0 | (fwd_b[i].part << 30);
            t.fwd_b[2 * i + 1] = fwd_b[i].id1;
        }
        t.n_eff = eff_offset(NUM_LAYERS);
        for (const std::vector<int>* v : {&t.fwd, &t.bwd, &t.fwd_b})
            for (int id : *v)
                if ((id & 0x3fffffff) > (int)t.n_eff) throw_error("internal: placement id %d beyond the %u effective weights", id, t.n_eff);
        return t;
    }();
    return tables;
}

void layer_sources(const VanerfWeightTable& w, LayerSrc out[NUM_LAYERS]) { sources(w, out); }

void pack_weights_host(const VanerfWeightTable& w, std::vector<float>& out, LayerOffsets& offs, int mode, std::vector<float>* bwd)
{
    const PackTables& t = pack_tables();
    std::vector<float> eff;
    fold_host(w, eff);
    auto value = [&](int id) { return id ? eff[(size_t)id - 1] : 0.0f; };
    for (int l = 0; l < NUM_LAYERS; ++l) offs.off[l] = mode ? layer_offset_b(l) : t.offs.off[l];
    if (mode) {
        const size_t n = t.fwd_b.size() / 2;
        out.resize(n);
        unsigned* dst = reinterpret_cast<unsigned*>(out.data());
        for (size_t i = 0; i < n; ++i) {
            const int a = t.fwd_b[2 * i], part = (a >> 30) & 1;
            dst[i] = (unsigned)bf16_part(value(a & 0x3fffffff), part) | ((unsigned)bf16_part(value(t.fwd_b[2 * i + 1]), part) << 16);
        }
    } else {
        out.resize(t.fwd.size());
        for (size_t i = 0; i < out.size(); ++i) out[i] = value(t.fwd[i]);
    }
    if (bwd) {
        bwd->resize(t.bwd.size());
        for (size_t i = 0; i < bwd->size(); ++i) (*bwd)[i] = value(t.bwd[i]);
    }
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
