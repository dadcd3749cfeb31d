/*
 * mesh_oracle.c -- CPU restatement (TEST INFRASTRUCTURE, never on the product path) of the
 * mesh-geometry inputs of the VANeRF hot path:
 *
 *   reference src/lib/dataset/mesh_util.py:498-524  cal_vis_sdf_batch
 *       kaolin.metrics.trianglemesh.point_to_mesh_distance  (mesh_util.py:509)
 *       kaolin.ops.mesh.check_sign                          (mesh_util.py:511)
 *       barycentric_coordinates_of_projection               (mesh_util.py:321-356)
 *   reference src/lib/dataset/mesh_util.py:284-318  get_visibility
 *       pytorch3d.renderer.mesh.rasterize_meshes            (mesh_util.py:303)
 *   reference src/networks.py:27-33                  KNN_vis -> pytorch3d.ops.knn_points (K=1)
 *
 * PARITY UNPINNED at this boundary: kaolin 0.15.0 and pytorch3d 0.7.5 are un-vendored
 * third-party CUDA libraries that are absent from /root/reference and from this image, and
 * the reference holds no fixture for them.  This file restates their *documented* semantics
 * (exact point->triangle distance with first-minimum face, watertight inside test, single
 * layer z-buffer with back-face culling, exact 1-NN with first-minimum tie break) with a
 * fully specified fp32 operation order, so that the HIP kernels in
 * vanerf_amd/csrc/mesh_kernels.hip can be compared BIT-EXACTLY (same IEEE operations, no
 * fused multiply-add: build with -ffp-contract=off).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct { float x, y, z; } f3;

static inline f3 sub3(f3 a, f3 b) { f3 r = {a.x - b.x, a.y - b.y, a.z - b.z}; return r; }
static inline float dot3(f3 a, f3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
static inline f3 madd3(f3 a, f3 d, float t) { f3 r = {a.x + d.x * t, a.y + d.y * t, a.z + d.z * t}; return r; }

/* Squared distance from p to triangle (a,b,c): closest-point regions (vertex / edge / face). */
static float point_tri_dist2(f3 p, f3 a, f3 b, f3 c)
{
    f3 ab = sub3(b, a), ac = sub3(c, a), ap = sub3(p, a);
    float d1 = dot3(ab, ap), d2 = dot3(ac, ap);
    f3 q;
    if (d1 <= 0.0f && d2 <= 0.0f) { q = a; goto done; }
    {
        f3 bp = sub3(p, b);
        float d3 = dot3(ab, bp), d4 = dot3(ac, bp);
        if (d3 >= 0.0f && d4 <= d3) { q = b; goto done; }
        float vc = d1 * d4 - d3 * d2;
        if (vc <= 0.0f && d1 >= 0.0f && d3 <= 0.0f) { q = madd3(a, ab, d1 / (d1 - d3)); goto done; }
        f3 cp = sub3(p, c);
        float d5 = dot3(ab, cp), d6 = dot3(ac, cp);
        if (d6 >= 0.0f && d5 <= d6) { q = c; goto done; }
        float vb = d5 * d2 - d1 * d6;
        if (vb <= 0.0f && d2 >= 0.0f && d6 <= 0.0f) { q = madd3(a, ac, d2 / (d2 - d6)); goto done; }
        float va = d3 * d6 - d5 * d4;
        float e1 = d4 - d3, e2 = d5 - d6;
        if (va <= 0.0f && e1 >= 0.0f && e2 >= 0.0f) { q = madd3(b, sub3(c, b), e1 / (e1 + e2)); goto done; }
        float den = 1.0f / ((va + vb) + vc);
        float v = vb * den, w = vc * den;
        q.x = (a.x + ab.x * v) + ac.x * w;
        q.y = (a.y + ab.y * v) + ac.y * w;
        q.z = (a.z + ab.z * v) + ac.z * w;
    }
done:;
    f3 r = sub3(p, q);
    return dot3(r, r);
}

/* Canonical (index-ordered) 2-D edge function in the (y,z) plane, exactly antisymmetric
 * between the two triangles that share the edge; `pos` = q is on the positive side, ties go
 * to the triangle that traverses the edge from the lower to the higher vertex index. */
static inline int edge_side(const float* V, int ia, int ib, float qy, float qz, float* Eout)
{
    int lo = ia < ib ? ia : ib, hi = ia < ib ? ib : ia;
    float ly = V[3 * lo + 1], lz = V[3 * lo + 2], hy = V[3 * hi + 1], hz = V[3 * hi + 2];
    float E = (hy - ly) * (qz - lz) - (hz - lz) * (qy - ly);
    int fwd = ia < ib;
    if (!fwd) E = -E;
    *Eout = E;
    return (E > 0.0f) || (E == 0.0f && fwd);
}

/* +x ray parity: inside iff the ray from p along +x crosses the closed surface an odd number of times. */
static int point_inside(const float* V, const int32_t* F, int nf, f3 p)
{
    int cnt = 0;
    for (int f = 0; f < nf; ++f) {
        int i0 = F[3 * f], i1 = F[3 * f + 1], i2 = F[3 * f + 2];
        float E0, E1, E2;
        int s0 = edge_side(V, i1, i2, p.y, p.z, &E0); /* weight of vertex 0 */
        int s1 = edge_side(V, i2, i0, p.y, p.z, &E1);
        int s2 = edge_side(V, i0, i1, p.y, p.z, &E2);
        if (!((s0 && s1 && s2) || (!s0 && !s1 && !s2))) continue;
        float den = (E0 + E1) + E2;
        if (den == 0.0f) continue;
        float xh = ((E0 * V[3 * i0] + E1 * V[3 * i1]) + E2 * V[3 * i2]) / den;
        if (xh > p.x) ++cnt;
    }
    return cnt & 1;
}

/* mesh_util.py:321-356: barycentric coordinates of the projection of p on triangle (v0,v1,v2). */
static void bary_of_projection(f3 p, f3 v0, f3 v1, f3 v2, float w[3])
{
    f3 u = sub3(v1, v0), v = sub3(v2, v0);
    f3 n = {u.y * v.z - u.z * v.y, u.z * v.x - u.x * v.z, u.x * v.y - u.y * v.x};
    float s = dot3(n, n);
    if (s == 0.0f) s = 1e-6f;
    float inv = 1.0f / s;
    f3 wv = sub3(p, v0);
    f3 c1 = {u.y * wv.z - u.z * wv.y, u.z * wv.x - u.x * wv.z, u.x * wv.y - u.y * wv.x};
    f3 c2 = {wv.y * v.z - wv.z * v.y, wv.z * v.x - wv.x * v.z, wv.x * v.y - wv.y * v.x};
    float b2 = dot3(c1, n) * inv;
    float b1 = dot3(c2, n) * inv;
    w[0] = (1.0f - b1) - b2; w[1] = b1; w[2] = b2;
}

/* cal_vis_sdf_batch (mesh_util.py:498-524) for one mesh:
 *   sdf[i]   = sqrt(d2 + 1e-6) * (inside ? -1 : +1)
 *   face[i]  = argmin face (first minimum)
 *   vis[i]   = (sum_k bary_k * vert_vis[face_k]) >= 0.1                                   */
void mesh_query(const float* V, int nv, const int32_t* F, int nf, const float* vert_vis,
                const float* P, int64_t n, float* sdf, uint8_t* vis, int32_t* face)
{
    (void)nv;
#pragma omp parallel for schedule(dynamic, 256)
    for (int64_t i = 0; i < n; ++i) {
        f3 p = {P[3 * i], P[3 * i + 1], P[3 * i + 2]};
        float best = INFINITY; int bf = 0;
        for (int f = 0; f < nf; ++f) {
            const float* a = V + 3 * F[3 * f]; const float* b = V + 3 * F[3 * f + 1]; const float* c = V + 3 * F[3 * f + 2];
            f3 A = {a[0], a[1], a[2]}, B = {b[0], b[1], b[2]}, C = {c[0], c[1], c[2]};
            float d = point_tri_dist2(p, A, B, C);
            if (d < best) { best = d; bf = f; }
        }
        int inside = point_inside(V, F, nf, p);
        float dist = sqrtf(best + 1e-6f);
        sdf[i] = inside ? -dist : dist;
        face[i] = bf;
        int i0 = F[3 * bf], i1 = F[3 * bf + 1], i2 = F[3 * bf + 2];
        f3 A = {V[3 * i0], V[3 * i0 + 1], V[3 * i0 + 2]}, B = {V[3 * i1], V[3 * i1 + 1], V[3 * i1 + 2]},
           C = {V[3 * i2], V[3 * i2 + 1], V[3 * i2 + 2]};
        float w[3];
        bary_of_projection(p, A, B, C, w);
        float s = (w[0] * vert_vis[i0] + w[1] * vert_vis[i1]) + w[2] * vert_vis[i2];
        vis[i] = s >= 0.1f;
    }
}

/* get_visibility (mesh_util.py:284-318): xyz = ((x01,y01,z01)+1)/2 rasterised at SxS pixel
 * centres (NDC c = -1 + (2i+1)/S), one face per pixel (smallest interpolated z, first on ties),
 * back faces (cross(v1-v0, v2-v0) > 0 in the fed coordinates) and zero-area faces skipped,
 * perspective-correct barycentrics.  Vertices of every face that owns a pixel are visible; the
 * reference also indexes faces[-1] through the -1 background id (mesh_util.py:314), so the
 * last face's vertices are always visible whenever some pixel is empty.                       */
void mesh_vertex_visibility(const float* xy01, const float* z01, int nv, const int32_t* F, int nf,
                            int S, float* vert_vis, int32_t* pix_to_face)
{
    float* X = (float*)malloc(sizeof(float) * 3 * nv);
    for (int i = 0; i < nv; ++i) {
        X[3 * i] = (xy01[2 * i] + 1.0f) / 2.0f;
        X[3 * i + 1] = (xy01[2 * i + 1] + 1.0f) / 2.0f;
        X[3 * i + 2] = (z01[i] + 1.0f) / 2.0f;
    }
    memset(vert_vis, 0, sizeof(float) * nv);
    int any_empty = 0;
    for (int py = 0; py < S; ++py)
        for (int px = 0; px < S; ++px) {
            float cx = -1.0f + (2.0f * (float)px + 1.0f) / (float)S;
            float cy = -1.0f + (2.0f * (float)py + 1.0f) / (float)S;
            float bestz = INFINITY; int bf = -1;
            for (int f = 0; f < nf; ++f) {
                const float* v0 = X + 3 * F[3 * f]; const float* v1 = X + 3 * F[3 * f + 1]; const float* v2 = X + 3 * F[3 * f + 2];
                float area = (v2[0] - v0[0]) * (v1[1] - v0[1]) - (v2[1] - v0[1]) * (v1[0] - v0[0]);
                if (area < 0.0f) continue;          /* back face */
                if (fabsf(area) <= 1e-8f) continue; /* zero area */
                float w0 = ((cx - v1[0]) * (v2[1] - v1[1]) - (cy - v1[1]) * (v2[0] - v1[0])) / area;
                float w1 = ((cx - v2[0]) * (v0[1] - v2[1]) - (cy - v2[1]) * (v0[0] - v2[0])) / area;
                float w2 = ((cx - v0[0]) * (v1[1] - v0[1]) - (cy - v0[1]) * (v1[0] - v0[0])) / area;
                /* perspective correction */
                float t0 = w0 * (v1[2] * v2[2]), t1 = w1 * (v0[2] * v2[2]), t2 = w2 * (v0[2] * v1[2]);
                float den = (t0 + t1) + t2;
                if (den == 0.0f) continue;
                float b0 = t0 / den, b1 = t1 / den, b2 = t2 / den;
                float pz = (b0 * v0[2] + b1 * v1[2]) + b2 * v2[2];
                if (pz < 0.0f) continue;
                if (!(b0 > 0.0f && b1 > 0.0f && b2 > 0.0f)) continue;
                if (pz < bestz) { bestz = pz; bf = f; }
            }
            if (pix_to_face) pix_to_face[py * S + px] = bf;
            if (bf < 0) { any_empty = 1; continue; }
            vert_vis[F[3 * bf]] = 1.0f; vert_vis[F[3 * bf + 1]] = 1.0f; vert_vis[F[3 * bf + 2]] = 1.0f;
        }
    if (any_empty) {
        int lf = nf - 1;
        vert_vis[F[3 * lf]] = 1.0f; vert_vis[F[3 * lf + 1]] = 1.0f; vert_vis[F[3 * lf + 2]] = 1.0f;
    }
    free(X);
}

/* knn_points(K=1): squared distance ((dx*dx + dy*dy) + dz*dz), first minimum. */
void knn1(const float* Q, int64_t n, const float* V, int nv, int32_t* idx)
{
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) {
        float qx = Q[3 * i], qy = Q[3 * i + 1], qz = Q[3 * i + 2];
        float best = INFINITY; int bi = 0;
        for (int j = 0; j < nv; ++j) {
            float dx = qx - V[3 * j], dy = qy - V[3 * j + 1], dz = qz - V[3 * j + 2];
            float d = (dx * dx + dy * dy) + dz * dz;
            if (d < best) { best = d; bi = j; }
        }
        idx[i] = bi;
    }
}
