// pass.cpp -- vanerf_render_pass: one whole pass of the hot path as a single C entry point (reference
// VANeRF.batch_render_pifu_nerf, src/model.py:1102-1360).  Host-side sequencing only: it enqueues the library's own entry points
// on the caller's stream, in the order vanerf_amd/renderer.py:render_pass does, with every temporary carved out of one
// caller-provided scratch block -- no allocation, no host synchronisation, nothing kept between calls.
#include "common.h"

using namespace vanerf;

namespace {

constexpr int64_t ALIGN = 256;
constexpr int64_t PARTITION_MIN_SAMPLES = 1 << 18; // as renderer.PARTITION_MIN_SAMPLES: below it the three partition kernels cost more than they save

struct Carver {
    char* base;
    int64_t off = 0;
    explicit Carver(void* b) : base(static_cast<char*>(b)) {}
    template <class T> T* take(int64_t n)
    {
        T* p = base ? reinterpret_cast<T*>(base + off) : nullptr;
        off += ((n * (int64_t)sizeof(T) + ALIGN - 1) / ALIGN) * ALIGN;
        return p;
    }
};

// The temporaries of a pass.  Laid out by the same code for the size query (base == NULL) and for the run.
struct Layout {
    float *rays_d, *cam_pos, *near, *far, *pts, *q_sdf_c, *rgba_c, *contrib, *z_new, *q_sdf_f, *rgba_f, *z_fine, *color_f3, *s1;
    float *raw_c, *rgba_cf;      // noisy reuse: the coarse points' raw outputs, and their eval_func with the fine batch's draws
    uint8_t *valid_c, *valid_f;
    uint8_t *q_vis;
    int32_t *knn, *order, *src;
    unsigned long long* queue_words; // work-queue heads of the pass's four big launches (mesh query and per-sample networks, coarse and fine)
    void* order_scratch;
    int64_t order_scratch_bytes, total;
};

// reuse: 0 the fine march evaluates all Sc + Sf samples; 1 it evaluates the Sf new ones and the composite gathers the coarse evaluations;
// 2 the same under per-sample noise: raw network outputs once per point, eval_func once per set of draws (vanerf_eval_func)
Layout carve(void* base, int R, int Sc, int Sf, int fine, int reuse)
{
    Carver c(base);
    Layout L{};
    const int64_t nc = (int64_t)R * Sc, nf = fine ? (int64_t)R * (reuse ? Sf : Sc + Sf) : 0, nmax = nc > nf ? nc : nf;
    L.queue_words = c.take<unsigned long long>(4);
    L.rays_d = c.take<float>(3LL * R);
    L.cam_pos = c.take<float>(4);
    L.near = c.take<float>(R);
    L.far = c.take<float>(R);
    L.pts = c.take<float>(3 * nmax);
    L.q_vis = c.take<uint8_t>(nmax);
    L.knn = c.take<int32_t>(nmax);
    L.order = c.take<int32_t>(nmax);
    L.order_scratch_bytes = vanerf_query_order_scratch(nmax);
    L.order_scratch = c.take<uint8_t>(L.order_scratch_bytes);
    L.q_sdf_c = c.take<float>(nc);
    L.rgba_c = c.take<float>(5 * nc);
    L.contrib = c.take<float>(nc);
    if (fine) {
        L.z_new = c.take<float>((int64_t)R * Sf);
        L.src = c.take<int32_t>((int64_t)R * (Sc + Sf));
        L.z_fine = c.take<float>((int64_t)R * (Sc + Sf));
        L.q_sdf_f = c.take<float>(nf);
        L.rgba_f = c.take<float>(5 * nf);
        if (reuse == 2) {
            L.raw_c = c.take<float>(5 * nc);
            L.rgba_cf = c.take<float>(5 * nc);
            L.valid_c = c.take<uint8_t>(nc);
            L.valid_f = c.take<uint8_t>(nf);
        }
    }
    L.color_f3 = c.take<float>(3LL * R); // stand-ins for optional outputs the caller did not ask for
    L.s1 = c.take<float>(4LL * R);
    L.total = c.off;
    return L;
}

void ok(int rc, const char* what)
{
    if (rc != VANERF_OK) {
        const std::string inner = vanerf_last_error();
        throw Error(rc, std::string("vanerf_render_pass: ") + what + ": " + inner);
    }
}

} // namespace

extern "C" int64_t vanerf_render_pass_scratch(int n_rays, int Sc, int Sf, int fine, int reuse_coarse)
{
    if (n_rays <= 0 || Sc <= 0 || Sf < 0) return 0;
    return carve(nullptr, n_rays, Sc, Sf, fine, reuse_coarse).total;
}

extern "C" int vanerf_render_pass(const VanerfWeights* w, const VanerfFrame* frame, const VanerfMeshAccel* accel, const float* verts, int nv,
                                  const int32_t* faces, int nf, const VanerfPassDesc* desc, const VanerfPassOut* out, void* scratch,
                                  int64_t scratch_bytes, void* stream)
{
    return guarded([&] {
        if (!w || !frame || !accel || !verts || !faces || !desc || !out || !scratch) throw_error("vanerf_render_pass: null argument");
        const VanerfPassDesc& d = *desc;
        const VanerfPassOut& o = *out;
        const int R = d.nx * d.ny, Sc = d.Sc, Sf = d.Sf, fine = d.fine != 0;
        if (d.nx <= 0 || d.ny <= 0 || Sc < 2 || (fine && Sf < 1)) throw_error("vanerf_render_pass: nx=%d ny=%d Sc=%d Sf=%d", d.nx, d.ny, Sc, Sf);
        if (!o.index || !o.hit || !o.z || !o.color || !o.depth || !o.alpha) throw_error("vanerf_render_pass: a coarse output pointer is null");
        if (!d.t_lin_c || (fine && !d.u && !d.t_lin_f)) throw_error("vanerf_render_pass: linspace tables missing");
        if (fine && d.noise_c && !d.noise_f) throw_error("vanerf_render_pass: noise_c without noise_f");
        const int reuse = !d.reuse_coarse || !fine ? 0 : d.noise_c ? 2 : 1;
        const Layout L = carve(scratch, R, Sc, Sf, fine, reuse);
        if (scratch_bytes < L.total) throw_error("vanerf_render_pass: scratch of %lld bytes, %lld needed (vanerf_render_pass_scratch)", (long long)scratch_bytes, (long long)L.total);
        // a1-a4: pixel grid, rays, bbox clip, coarse depths
        if (d.pixels_xy)
            ok(vanerf_ray_setup_pixels(d.pixels_xy, R, d.width, d.invK_T, d.RT, d.znear, d.zfar, d.bounds, Sc, d.t_lin_c, d.jitter, o.index, L.rays_d,
                                       L.cam_pos, L.near, L.far, o.hit, o.z, stream), "ray setup");
        else if (d.row_blocks)
            ok(vanerf_ray_setup_blocks(d.row_blocks, d.x0, d.step_x, d.y_block, d.nx, d.ny, d.width, d.invK_T, d.RT, d.znear, d.zfar, d.bounds, Sc, d.t_lin_c,
                                       d.jitter, o.index, L.rays_d, L.cam_pos, L.near, L.far, o.hit, o.z, stream), "ray setup");
        else
            ok(vanerf_ray_setup(d.x0, d.y0, d.step_x, d.step_y, d.y_block, d.nx, d.ny, d.width, d.invK_T, d.RT, d.znear, d.zfar, d.bounds, Sc, d.t_lin_c,
                                d.jitter, o.index, L.rays_d, L.cam_pos, L.near, L.far, o.hit, o.z, stream), "ray setup");
        // one march: points, mesh query (+ 1-NN), validity partition, per-sample networks
        int marches = 0; // every launch with a work queue gets a word of its own from the scratch block (nothing is shared between launches in flight)
        auto march = [&](const float* z, int S, const float* noise, float* q_sdf, float* rgba, uint8_t* valid_raw = nullptr) { // valid_raw: raw outputs + flags
            unsigned long long* const qw = L.queue_words + 2 * marches++;
            const int64_t n = (int64_t)R * S;
            ok(vanerf_sample_points(L.rays_d, L.cam_pos, z, R, S, L.pts, stream), "sample points");
            const bool grid = d.pixels_xy == nullptr;
            ok(vanerf_mesh_query_accel(accel, verts, nv, faces, nf, frame->vert_vis, L.pts, n, q_sdf, L.q_vis, nullptr, L.knn, grid ? d.nx : 0,
                                       grid ? d.ny : 0, grid ? S : 0, qw, stream), "mesh query");
            const int32_t* order = nullptr;
            if (n >= PARTITION_MIN_SAMPLES) {
                ok(vanerf_query_order(frame, L.pts, n, L.order, L.order_scratch, L.order_scratch_bytes, stream), "validity partition");
                order = L.order;
            }
            ok(vanerf_query_samples(w, frame, L.pts, q_sdf, L.q_vis, L.knn, noise, order, valid_raw ? 1 : 0, n, rgba, valid_raw, qw + 1, stream), "per-sample networks");
        };
        if (reuse == 2) { // the networks once per point; eval_func with the coarse draws here, with the fine batch's draws below
            march(o.z, Sc, nullptr, L.q_sdf_c, L.raw_c, L.valid_c);
            ok(vanerf_eval_func(L.raw_c, L.valid_c, nullptr, nullptr, nullptr, d.noise_c, Sc, 0, R, frame->invalid_sdf, L.rgba_c, nullptr, stream), "eval_func (coarse)");
        } else {
            march(o.z, Sc, d.noise_c, L.q_sdf_c, L.rgba_c);
        }
        composite_with_handle(w, L.rgba_c, o.z, L.q_sdf_c, nullptr, nullptr, nullptr, Sc, 0, R, o.color, o.depth, o.alpha, L.s1, L.contrib, stream);
        if (!fine) return;
        float* z_fine = o.z_fine ? o.z_fine : L.z_fine;
        float* cf = o.color_fine ? o.color_fine : L.color_f3;
        float* df = o.depth_fine ? o.depth_fine : L.s1 + R;
        float* af = o.alpha_fine ? o.alpha_fine : L.s1 + 2LL * R;
        float* sf = o.sdf ? o.sdf : L.s1 + 3LL * R;
        ok(vanerf_importance_merge(L.contrib, o.z, d.u, d.u ? nullptr : d.t_lin_f, R, Sc, Sf, L.z_new, z_fine, L.src, nullptr, stream), "importance sampling");
        if (reuse == 2) {
            march(L.z_new, Sf, nullptr, L.q_sdf_f, L.rgba_f, L.valid_f);
            // noise_f holds one draw per position of the merged order (the reference draws them for the re-evaluated fine batch, src/model.py:1155-1156)
            ok(vanerf_eval_func(L.raw_c, L.valid_c, L.rgba_f, L.valid_f, L.src, d.noise_f, Sc, Sf, R, frame->invalid_sdf, L.rgba_cf, L.rgba_f, stream), "eval_func (fine)");
            composite_with_handle(w, L.rgba_cf, z_fine, L.q_sdf_c, L.rgba_f, L.q_sdf_f, L.src, Sc, Sf, R, cf, df, af, sf, nullptr, stream);
        } else if (reuse) {
            march(L.z_new, Sf, nullptr, L.q_sdf_f, L.rgba_f);
            composite_with_handle(w, L.rgba_c, z_fine, L.q_sdf_c, L.rgba_f, L.q_sdf_f, L.src, Sc, Sf, R, cf, df, af, sf, nullptr, stream);
        } else {
            march(z_fine, Sc + Sf, d.noise_f, L.q_sdf_f, L.rgba_f);
            composite_with_handle(w, L.rgba_f, z_fine, L.q_sdf_f, nullptr, nullptr, nullptr, Sc + Sf, 0, R, cf, df, af, sf, nullptr, stream);
        }
    });
}
