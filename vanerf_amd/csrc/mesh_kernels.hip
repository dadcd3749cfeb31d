// mesh_kernels.hip -- mesh-geometry inputs of the per-sample networks:
//   cal_vis_sdf_batch   src/lib/dataset/mesh_util.py:498-524 (kaolin point_to_mesh_distance / check_sign)
//   get_visibility      src/lib/dataset/mesh_util.py:284-318 (pytorch3d rasterize_meshes)
//   knn_points (K=1)    src/networks.py:28
// The arithmetic is the operation-for-operation twin of oracle/mesh_oracle.c (third-party semantics, parity
// unpinned against the reference; bit-exact against the oracle).  Built with -ffp-contract=off.
// (A branch-free point_tri_dist2 with one shared division was measured: bit-exact, but 7 % slower -- the lanes of a wave mostly
// agree on the Voronoi region, so the branches are cheap.)
#include "common.h"

using namespace vanerf;

namespace {

struct f3 { float x, y, z; };
__device__ __forceinline__ f3 sub3(f3 a, f3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ float dot3(f3 a, f3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
__device__ __forceinline__ f3 madd3(f3 a, f3 d, float t) { return {a.x + d.x * t, a.y + d.y * t, a.z + d.z * t}; }

// squared distance from p to triangle (a, b, c); q = the closest point of the triangle
__device__ __forceinline__ float point_tri_dist2_q(f3 p, f3 a, f3 b, f3 c, f3& q)
{
    const f3 ab = sub3(b, a), ac = sub3(c, a), ap = sub3(p, a);
    const float d1 = dot3(ab, ap), d2 = dot3(ac, ap);
    if (d1 <= 0.0f && d2 <= 0.0f) {
        q = a;
    } else {
        const f3 bp = sub3(p, b);
        const float d3 = dot3(ab, bp), d4 = dot3(ac, bp);
        if (d3 >= 0.0f && d4 <= d3) {
            q = b;
        } else {
            const float vc = d1 * d4 - d3 * d2;
            if (vc <= 0.0f && d1 >= 0.0f && d3 <= 0.0f) {
                q = madd3(a, ab, d1 / (d1 - d3));
            } else {
                const f3 cp = sub3(p, c);
                const float d5 = dot3(ab, cp), d6 = dot3(ac, cp);
                if (d6 >= 0.0f && d5 <= d6) {
                    q = c;
                } else {
                    const float vb = d5 * d2 - d1 * d6;
                    if (vb <= 0.0f && d2 >= 0.0f && d6 <= 0.0f) {
                        q = madd3(a, ac, d2 / (d2 - d6));
                    } else {
                        const float va = d3 * d6 - d5 * d4;
                        const float e1 = d4 - d3, e2 = d5 - d6;
                        if (va <= 0.0f && e1 >= 0.0f && e2 >= 0.0f) {
                            q = madd3(b, sub3(c, b), e1 / (e1 + e2));
                        } else {
                            const float den = 1.0f / ((va + vb) + vc);
                            const float v = vb * den, w = vc * den;
                            q.x = (a.x + ab.x * v) + ac.x * w;
                            q.y = (a.y + ab.y * v) + ac.y * w;
                            q.z = (a.z + ab.z * v) + ac.z * w;
                        }
                    }
                }
            }
        }
    }
    const f3 r = sub3(p, q);
    return dot3(r, r);
}

__device__ __forceinline__ float point_tri_dist2(f3 p, f3 a, f3 b, f3 c)
{
    f3 q;
    return point_tri_dist2_q(p, a, b, c, q);
}

// canonical (index-ordered) edge function in the (y,z) plane, see oracle/mesh_oracle.c:edge_side
__device__ __forceinline__ bool edge_side(f3 va, f3 vb, int ia, int ib, float qy, float qz, float& E)
{
    const bool fwd = ia < ib;
    const f3 lo = fwd ? va : vb, hi = fwd ? vb : va;
    float e = (hi.y - lo.y) * (qz - lo.z) - (hi.z - lo.z) * (qy - lo.y);
    if (!fwd) e = -e;
    E = e;
    return (e > 0.0f) || (e == 0.0f && fwd);
}

constexpr int MQ_BLOCK = 256;
constexpr int MQ_TILE = 512; // triangles staged per LDS tile

__global__ __launch_bounds__(MQ_BLOCK) void mesh_query_kernel(const float* __restrict__ V, const int32_t* __restrict__ F, int nf,
                                                              const float* __restrict__ vert_vis, const float* __restrict__ P,
                                                              long long n, float* __restrict__ sdf, uint8_t* __restrict__ vis,
                                                              int32_t* __restrict__ face)
{
    __shared__ float s_tri[MQ_TILE][9];
    __shared__ int s_idx[MQ_TILE][3];
    const long long i = (long long)blockIdx.x * MQ_BLOCK + threadIdx.x;
    const bool live = i < n;
    const long long ii = live ? i : n - 1;
    const f3 p = {P[3 * ii], P[3 * ii + 1], P[3 * ii + 2]};
    float best = INFINITY;
    int bf = 0, cnt = 0;
    for (int f0 = 0; f0 < nf; f0 += MQ_TILE) {
        const int m = min(MQ_TILE, nf - f0);
        __syncthreads();
        for (int k = threadIdx.x; k < m * 3; k += MQ_BLOCK) {
            const int f = k / 3, c = k % 3;
            const int vi = F[3 * (f0 + f) + c];
            s_idx[f][c] = vi;
            s_tri[f][3 * c + 0] = V[3 * vi]; s_tri[f][3 * c + 1] = V[3 * vi + 1]; s_tri[f][3 * c + 2] = V[3 * vi + 2];
        }
        __syncthreads();
        for (int f = 0; f < m; ++f) {
            const f3 a = {s_tri[f][0], s_tri[f][1], s_tri[f][2]}, b = {s_tri[f][3], s_tri[f][4], s_tri[f][5]},
                     c = {s_tri[f][6], s_tri[f][7], s_tri[f][8]};
            const float d = point_tri_dist2(p, a, b, c);
            if (d < best) { best = d; bf = f0 + f; }
            // +x ray parity (oracle/mesh_oracle.c:point_inside)
            const int i0 = s_idx[f][0], i1 = s_idx[f][1], i2 = s_idx[f][2];
            float E0, E1, E2;
            const bool s0 = edge_side(b, c, i1, i2, p.y, p.z, E0);
            const bool s1 = edge_side(c, a, i2, i0, p.y, p.z, E1);
            const bool s2 = edge_side(a, b, i0, i1, p.y, p.z, E2);
            if ((s0 && s1 && s2) || (!s0 && !s1 && !s2)) {
                const float den = (E0 + E1) + E2;
                if (den != 0.0f) {
                    const float xh = ((E0 * a.x + E1 * b.x) + E2 * c.x) / den;
                    if (xh > p.x) ++cnt;
                }
            }
        }
    }
    if (!live) return;
    const float dist = sqrtf(best + 1e-6f);
    sdf[i] = (cnt & 1) ? -dist : dist;
    if (face) face[i] = bf;
    // barycentric_coordinates_of_projection (mesh_util.py:321-356) on the closest face
    const int i0 = F[3 * bf], i1 = F[3 * bf + 1], i2 = F[3 * bf + 2];
    const f3 v0 = {V[3 * i0], V[3 * i0 + 1], V[3 * i0 + 2]}, v1 = {V[3 * i1], V[3 * i1 + 1], V[3 * i1 + 2]},
             v2 = {V[3 * i2], V[3 * i2 + 1], V[3 * i2 + 2]};
    const f3 u = sub3(v1, v0), v = sub3(v2, v0);
    const f3 nrm = {u.y * v.z - u.z * v.y, u.z * v.x - u.x * v.z, u.x * v.y - u.y * v.x};
    float s = dot3(nrm, nrm);
    if (s == 0.0f) s = 1e-6f;
    const float inv = 1.0f / s;
    const f3 wv = sub3(p, v0);
    const f3 c1 = {u.y * wv.z - u.z * wv.y, u.z * wv.x - u.x * wv.z, u.x * wv.y - u.y * wv.x};
    const f3 c2 = {wv.y * v.z - wv.z * v.y, wv.z * v.x - wv.x * v.z, wv.x * v.y - wv.y * v.x};
    const float b2 = dot3(c1, nrm) * inv;
    const float b1 = dot3(c2, nrm) * inv;
    const float w0 = (1.0f - b1) - b2;
    const float sv = (w0 * vert_vis[i0] + b1 * vert_vis[i1]) + b2 * vert_vis[i2];
    vis[i] = sv >= 0.1f;
}

// get_visibility: one thread per raster pixel, one block per 16 x 16 pixel tile.  Faces are streamed through LDS 256 at a time; a face is
// staged only if it can cover a pixel centre of the tile (front facing, not degenerate, bounding box within 1e-3 units of the tile -- the
// inside test of a pixel further than that from the box of an ordinary face fails in fp32 whatever the rounding; faces with a vertex behind
// the camera plane and needles are staged for every tile).  The staged faces keep their order (the first of equal depths wins, as in the full scan),
// and the per-pixel arithmetic is unchanged, so the result is the full scan's, bit for bit (13 x fewer tests on the two-hand mesh).
constexpr int RV_BLOCK = 256;
constexpr int RV_T = 16;

__global__ __launch_bounds__(RV_BLOCK) void raster_kernel(const float* __restrict__ xy01, const float* __restrict__ z01,
                                                          const int32_t* __restrict__ F, int nf, int S, int32_t* __restrict__ pix_to_face)
{
    __shared__ float s_v[RV_BLOCK][9];
    __shared__ int s_id[RV_BLOCK];
    __shared__ int s_wcnt[RV_BLOCK / 64];
    const int tiles_x = (S + RV_T - 1) / RV_T;
    const int px0 = (blockIdx.x % tiles_x) * RV_T, py0 = (blockIdx.x / tiles_x) * RV_T;
    const int px = px0 + (threadIdx.x & (RV_T - 1)), py = py0 + threadIdx.x / RV_T;
    const bool live = px < S && py < S;
    const float cx = -1.0f + (2.0f * (float)px + 1.0f) / (float)S;
    const float cy = -1.0f + (2.0f * (float)py + 1.0f) / (float)S;
    const float tol = 1e-3f;
    const float tx_lo = -1.0f + (2.0f * (float)px0 + 1.0f) / (float)S - tol, tx_hi = -1.0f + (2.0f * (float)min(px0 + RV_T - 1, S - 1) + 1.0f) / (float)S + tol;
    const float ty_lo = -1.0f + (2.0f * (float)py0 + 1.0f) / (float)S - tol, ty_hi = -1.0f + (2.0f * (float)min(py0 + RV_T - 1, S - 1) + 1.0f) / (float)S + tol;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float bestz = INFINITY;
    int bf = -1;
    for (int f0 = 0; f0 < nf; f0 += RV_BLOCK) {
        const int f = f0 + threadIdx.x;
        float v[9];
        bool keep = false;
        if (f < nf) {
            for (int c = 0; c < 3; ++c) {
                const int vi = F[3 * f + c];
                v[3 * c + 0] = (xy01[2 * vi] + 1.0f) / 2.0f;
                v[3 * c + 1] = (xy01[2 * vi + 1] + 1.0f) / 2.0f;
                v[3 * c + 2] = (z01[vi] + 1.0f) / 2.0f;
            }
            const float area = (v[6] - v[0]) * (v[4] - v[1]) - (v[7] - v[1]) * (v[3] - v[0]);
            const float xlo = fminf(v[0], fminf(v[3], v[6])), xhi = fmaxf(v[0], fmaxf(v[3], v[6]));
            const float ylo = fminf(v[1], fminf(v[4], v[7])), yhi = fmaxf(v[1], fmaxf(v[4], v[7]));
            // the box test only speaks for an ordinary face: all three depths positive (then the signs of the perspective-corrected weights are
            // the signs of the edge functions) and not a needle (area against its box: the edge functions of a needle cancel to rounding noise)
            const bool ordinary = v[2] > 0.0f && v[5] > 0.0f && v[8] > 0.0f && area > 1e-4f * ((xhi - xlo) * (yhi - ylo));
            const bool off_tile = xlo > tx_hi || xhi < tx_lo || ylo > ty_hi || yhi < ty_lo;
            keep = !(area < 0.0f) && !(fabsf(area) <= 1e-8f) && !(ordinary && off_tile); // (the first two: the early-outs of the loop below)
        }
        const unsigned long long m = __ballot(keep);
        __syncthreads(); // the previous pass's readers are done
        if (lane == 0) s_wcnt[wave] = __popcll(m);
        __syncthreads();
        int at = __popcll(m & ((1ull << lane) - 1ull)), cnt = 0;
        for (int w = 0; w < RV_BLOCK / 64; ++w) {
            if (w < wave) at += s_wcnt[w];
            cnt += s_wcnt[w];
        }
        if (keep) {
            for (int k = 0; k < 9; ++k) s_v[at][k] = v[k];
            s_id[at] = f;
        }
        __syncthreads();
        for (int k = 0; k < cnt; ++k) {
            const float* v0 = &s_v[k][0]; const float* v1 = &s_v[k][3]; const float* v2 = &s_v[k][6];
            const float area = (v2[0] - v0[0]) * (v1[1] - v0[1]) - (v2[1] - v0[1]) * (v1[0] - v0[0]);
            if (area < 0.0f) continue;
            if (fabsf(area) <= 1e-8f) continue;
            const float w0 = ((cx - v1[0]) * (v2[1] - v1[1]) - (cy - v1[1]) * (v2[0] - v1[0])) / area;
            const float w1 = ((cx - v2[0]) * (v0[1] - v2[1]) - (cy - v2[1]) * (v0[0] - v2[0])) / area;
            const float w2 = ((cx - v0[0]) * (v1[1] - v0[1]) - (cy - v0[1]) * (v1[0] - v0[0])) / area;
            const float t0 = w0 * (v1[2] * v2[2]), t1 = w1 * (v0[2] * v2[2]), t2 = w2 * (v0[2] * v1[2]);
            const float den = (t0 + t1) + t2;
            if (den == 0.0f) continue;
            const float b0 = t0 / den, b1 = t1 / den, b2 = t2 / den;
            const float pz = (b0 * v0[2] + b1 * v1[2]) + b2 * v2[2];
            if (pz < 0.0f) continue;
            if (!(b0 > 0.0f && b1 > 0.0f && b2 > 0.0f)) continue;
            if (pz < bestz) { bestz = pz; bf = s_id[k]; }
        }
    }
    if (live) pix_to_face[py * S + px] = bf;
}

__global__ void mark_visible_kernel(const int32_t* __restrict__ pix_to_face, int npix, const int32_t* __restrict__ F, int nf,
                                    float* __restrict__ vert_vis)
{
    const int pix = blockIdx.x * blockDim.x + threadIdx.x;
    if (pix >= npix) return;
    int f = pix_to_face[pix];
    if (f < 0) f = nf - 1; // faces[-1] through the background id (mesh_util.py:314)
    vert_vis[F[3 * f]] = 1.0f; vert_vis[F[3 * f + 1]] = 1.0f; vert_vis[F[3 * f + 2]] = 1.0f;
}

__global__ __launch_bounds__(256) void knn1_kernel(const float4* __restrict__ verts, int nv, const float* __restrict__ pts, long long n,
                                                   int32_t* __restrict__ idx)
{
    extern __shared__ float4 s_vv[];
    for (int i = threadIdx.x; i < nv; i += blockDim.x) s_vv[i] = verts[i];
    __syncthreads();
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float qx = pts[3 * i], qy = pts[3 * i + 1], qz = pts[3 * i + 2];
    float best = INFINITY;
    int bi = 0;
    for (int j = 0; j < nv; ++j) {
        const float4 v = s_vv[j];
        const float dx = qx - v.x, dy = qy - v.y, dz = qz - v.z;
        const float d = (dx * dx + dy * dy) + dz * dz;
        if (d < best) { best = d; bi = j; }
    }
    idx[i] = bi;
}


// ---------------------------------------------------------------------------------------------------------------
// Accelerated cal_vis_sdf_batch: identical results to mesh_query_kernel (same per-triangle arithmetic), but
//   * closest face: triangles are Morton-sorted into clusters of 16 with an AABB per cluster and a bounding sphere
//     per triangle; a cluster / triangle is skipped only when its distance lower bound exceeds the best distance so
//     far by a safety margin (1e-4 relative: far above fp32 rounding), so the exact minimum and every tie survive;
//     ties resolve to the lowest ORIGINAL face index, as the brute-force scan does;
//   * inside test: a G x G grid over the mesh's (y,z) extent lists the triangles whose (y,z) bounding box touches each
//     cell; the +x ray of a point can only cross triangles of its own cell.
// The structure is built per source frame by vanerf_mesh_accel_build (end of this file).
#ifndef VANERF_MA_CL
#define VANERF_MA_CL 4
#endif
constexpr int CL = VANERF_MA_CL; // triangles (and vertices) per cluster.  Measured on the benchmark view: 16 -> 6.35 ms, 8 -> 4.80, 4 -> 4.37
                                 // (box tests are cheap since they are done per wave first; small clusters mean fewer per-triangle tests)
#ifndef VANERF_MA_BLOCK
#define VANERF_MA_BLOCK 1024 // one block per CU: 16 waves share the LDS copy of the vertex / box tables (53 KB for the two-hand mesh)
#endif
constexpr int MA_BLOCK = VANERF_MA_BLOCK;
constexpr int MA_MAX_CLUSTERS = 4096;
constexpr int MA_MAX_VCLUSTERS = 1024;
// tile searches (below): clusters a wave's candidate list can hold, and the 64-candidate rounds that makes
#ifndef VANERF_TL_LIST
#define VANERF_TL_LIST 160 // measured on the benchmark view with the work queue: 64 -> 2.34 ms, 128 -> 2.00, 160 -> 1.93, 192 -> 1.95, 256 -> 2.16
#endif
constexpr int TL_LIST = VANERF_TL_LIST;
constexpr int TL_IT = TL_LIST * CL / 64;
#ifndef VANERF_MA_ND
#define VANERF_MA_ND 2
#endif
constexpr int ND_MAX = VANERF_MA_ND; // consecutive depths of a pixel tile a wave takes at once (template parameter ND of the kernel): one tile search, ND per-lane evaluations
#ifndef VANERF_TL_CAND
#define VANERF_TL_CAND 40
#endif
constexpr int TL_CAND = VANERF_TL_CAND; // triangles (9 floats + original index, padded to 12) a wave's candidate table holds
static_assert(TL_LIST * CL % 64 == 0 && 64 % CL == 0 && MA_MAX_CLUSTERS <= 65536, "candidate lists hold 16-bit cluster ids in whole rounds of 64");

// Diagnostic build only (-DVANERF_MESH_PHASES): s_memtime deltas per phase, summed over waves (tools/perf_mesh.py --phases)
#ifdef VANERF_MESH_PHASES
__device__ unsigned long long g_ma_phase[16];
#define MPH(k) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); ph[k] += t_ - tprev; tprev = t_; } while (0)
#else
#define MPH(k) do { } while (0)
#endif

// Wave-wide reductions on the DPP path (quad swaps, half-row / row mirrors, row broadcasts; the result is read from lane 63 into a scalar
// register): six VALU moves instead of the six LDS round trips of __shfl_xor's ds_bpermute, and no address registers.  The searches below
// reduce a few dozen times per 64 points and are bound by exactly this kind of dependent latency.
template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ int dpp_i(int v) { return __builtin_amdgcn_update_dpp(v, v, CTRL, ROW_MASK, 0xF, false); } // masked-off rows keep v
template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ float dpp_f(float v) { return __int_as_float(dpp_i<CTRL, ROW_MASK>(__float_as_int(v))); }
#define VANERF_WAVE_REDUCE(T, DPP, OP)                                                                                                   \
    v = OP(v, DPP<0xB1>(v));        /* quad_perm [1,0,3,2] */                                                                             \
    v = OP(v, DPP<0x4E>(v));        /* quad_perm [2,3,0,1] */                                                                             \
    v = OP(v, DPP<0x141>(v));       /* row_half_mirror */                                                                                 \
    v = OP(v, DPP<0x140>(v));       /* row_mirror: every lane holds its row of 16 */                                                      \
    v = OP(v, DPP<0x142, 0xA>(v));  /* row_bcast15 into rows 1, 3 */                                                                      \
    v = OP(v, DPP<0x143, 0xC>(v));  /* row_bcast31 into rows 2, 3: lane 63 holds all 64 */
__device__ __forceinline__ float wave_min_f(float v) { VANERF_WAVE_REDUCE(float, dpp_f, fminf) return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63)); }
__device__ __forceinline__ float wave_max_f(float v) { VANERF_WAVE_REDUCE(float, dpp_f, fmaxf) return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63)); }
__device__ __forceinline__ int wave_min_i(int v) { VANERF_WAVE_REDUCE(int, dpp_i, min) return __builtin_amdgcn_readlane(v, 63); }
#undef VANERF_WAVE_REDUCE

// a value that is the same in every lane, moved to a scalar register (the tile's box, centre, radius: they live through the whole search)
__device__ __forceinline__ float uni(float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); }

__device__ __forceinline__ float box_dist2(f3 p, const float* b)
{
    const float dx = fmaxf(fmaxf(b[0] - p.x, p.x - b[3]), 0.0f);
    const float dy = fmaxf(fmaxf(b[1] - p.y, p.y - b[4]), 0.0f);
    const float dz = fmaxf(fmaxf(b[2] - p.z, p.z - b[5]), 0.0f);
    return (dx * dx + dy * dy) + dz * dz;
}

#ifndef VANERF_MA_ATTR
#define VANERF_MA_ATTR
#endif

template <int ND> // 1 or ND_MAX: small launches take one depth per item (twice the items: a launch of a few thousand items is bound by the latency of single items)
__global__ __launch_bounds__(MA_BLOCK) VANERF_MA_ATTR void mesh_query_accel_kernel(const VanerfMeshAccel A, const float* __restrict__ V,
                                                                    const int32_t* __restrict__ F, const float* __restrict__ vert_vis,
                                                                    const float* __restrict__ P, long long n, float* __restrict__ sdf,
                                                                    uint8_t* __restrict__ vis, int32_t* __restrict__ face,
                                                                    int32_t* __restrict__ knn, int gnx, int gny, int gS,
                                                                    unsigned long long* __restrict__ queue)
{
    // dynamic LDS: [nvc*16] sorted vertices (float4) | [nvc][6] vertex-cluster boxes | [nc][6] triangle-cluster boxes
    extern __shared__ float4 s_dyn[];
    __shared__ unsigned short s_list[MA_BLOCK / 64][TL_LIST]; // per-wave candidate lists of the tile searches
    __shared__ __attribute__((aligned(16))) float s_cand[MA_BLOCK / 64][TL_CAND][12];
    __shared__ float s_dit[MA_BLOCK / 64][TL_IT][64]; // per-wave, per-lane distances of the searches' rounds (registers are the scarce resource here)
    float4* s_vs = s_dyn;
    float* s_vbox = reinterpret_cast<float*>(s_vs + A.nvc * CL);
    float* s_box = s_vbox + A.nvc * 6;
    for (int k = threadIdx.x; k < A.nc * 6; k += MA_BLOCK) s_box[k] = A.cbox[k];
    for (int k = threadIdx.x; k < A.nvc * 6; k += MA_BLOCK) s_vbox[k] = A.vbox[k];
    for (int k = threadIdx.x; k < A.nvc * CL; k += MA_BLOCK) s_vs[k] = reinterpret_cast<const float4*>(A.vsort)[k];
    // (y,z) grid of the inside test: written on the device by vanerf_mesh_accel_build (uniform loads)
    const float grid_y0 = A.grid[0], grid_z0 = A.grid[1], grid_cy = A.grid[2], grid_cz = A.grid[3];
    const int grid_G = __float_as_int(A.grid[4]);
    __syncthreads();
    // Work mapping.  The pruning below is data dependent, so a wave runs the UNION of its lanes' candidate lists.  With the
    // ray-grid hint (gnx x gny rays, gS samples per ray, sample index = ray * gS + depth) a wave takes one depth of an 8 x 8
    // pixel tile: 64 points a few millimetres apart that prune almost identically.  Without the hint (gnx == 0) consecutive
    // lanes take consecutive samples (one whole ray per wave: its points span the bounding box, ~4x more work per wave).
    const int lane = threadIdx.x & 63;
    // the lane number as a value the compiler must take afresh: index and address arithmetic on it is then redone where it is used instead of
    // being hoisted out of the work loop into registers that stay occupied for the whole kernel (the kernel lives at the 128-register limit)
    auto lane_now = [&]() { int l = lane; asm volatile("" : "+v"(l)); return l; };
    // Work queue: a wave claims its next 64 points with one atomic.  The cost of a wave's search varies 10x with the tile's distance from
    // the mesh; a fixed assignment left 40 % of the wave slots idle behind the slowest waves (2.45 of 4 waves per SIMD on average).
    auto claim = [&]() {
        unsigned long long v = 0ull;
        if (lane == 0) v = atomicAdd(queue, 1ull);
        return (long long)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) |
                           (unsigned)__builtin_amdgcn_readfirstlane((int)v));
    };
    const int ntx = (gnx + 7) >> 3, nty = (gny + 7) >> 3;
    const int gD = (gS + ND - 1) / ND; // depth groups per tile
    const long long nwork = gnx > 0 ? (long long)ntx * nty * gD : (n + 64 * ND - 1) / (64 * ND);
#ifdef VANERF_MESH_PHASES
    unsigned long long ph[16] = {}, tprev = __builtin_amdgcn_s_memtime();
#endif
    for (long long w = claim(); w < nwork; w = claim()) {
        // every lane keeps a point (the searches below are wave-cooperative): a lane beyond the grid border / the end of the batch
        // repeats a neighbour's point and does not store
        // sample index of depth k of this item for this lane, and whether the lane stores it (recomputed where needed rather than kept)
        // (the item's depth group and pixel tile: wave-uniform, divided out once; the host checks that the item count fits 32 bits)
        int item_d0 = 0, item_x0 = 0, item_y0 = 0;
        if (gnx > 0) {
            const unsigned wu = (unsigned)w, tile = wu / (unsigned)gD;
            item_d0 = __builtin_amdgcn_readfirstlane((int)(wu % (unsigned)gD) * ND);
            item_x0 = __builtin_amdgcn_readfirstlane((int)(tile % (unsigned)ntx) * 8);
            item_y0 = __builtin_amdgcn_readfirstlane((int)(tile / (unsigned)ntx) * 8);
        }
        auto locate = [&](int k, bool& act) -> long long {
            long long i;
            if (gnx > 0) {
                const int d = item_d0 + k;
                const int l = lane_now();
                const int rx = item_x0 + (l & 7), ry = item_y0 + (l >> 3);
                act = rx < gnx && ry < gny && d < gS;
                i = ((long long)min(ry, gny - 1) * gnx + min(rx, gnx - 1)) * gS + min(d, gS - 1);
            } else {
                i = (w * ND + k) * 64 + lane;
                act = i < n;
                i = act ? i : n - 1;
            }
            return i;
        };
        f3 pk[ND];
#pragma unroll
        for (int k = 0; k < ND; ++k) {
            bool act;
            const long long i = locate(k, act);
            pk[k] = {P[3 * i], P[3 * i + 1], P[3 * i + 2]};
        }
        MPH(0); // point load
        // ---- 1-NN vertex (knn_points K=1, src/networks.py:28): clusters of 16 Morton-sorted vertices; squared distance
        //      ((dx*dx + dy*dy) + dz*dz), first minimum in ORIGINAL vertex order (oracle/mesh_oracle.c:knn1)
        // Both searches prune per WAVE first: the 64 points of a wave lie in a small box T, and a cluster whose box is farther from
        // T than the largest bound any lane holds cannot matter to any lane.  The lanes test 64 clusters at a time against T (one
        // cluster per lane, __ballot), and only the surviving clusters are then tested by every lane against its own point and
        // bound -- the flat per-lane loop over all boxes this replaces was 2/3 of the kernel's instructions.
        float tlo[3] = {pk[0].x, pk[0].y, pk[0].z}, thi[3] = {pk[0].x, pk[0].y, pk[0].z};
#pragma unroll
        for (int k = 1; k < ND; ++k) {
            tlo[0] = fminf(tlo[0], pk[k].x); tlo[1] = fminf(tlo[1], pk[k].y); tlo[2] = fminf(tlo[2], pk[k].z);
            thi[0] = fmaxf(thi[0], pk[k].x); thi[1] = fmaxf(thi[1], pk[k].y); thi[2] = fmaxf(thi[2], pk[k].z);
        }
#pragma unroll
        for (int a = 0; a < 3; ++a) { tlo[a] = wave_min_f(tlo[a]); thi[a] = wave_max_f(thi[a]); }
        auto tile_dist2 = [&](const float* b) { // box-to-box: <= box_dist2(q, b) for every q in T (same monotone fp32 expression)
            const float dx = fmaxf(fmaxf(b[0] - thi[0], tlo[0] - b[3]), 0.0f);
            const float dy = fmaxf(fmaxf(b[1] - thi[1], tlo[1] - b[4]), 0.0f);
            const float dz = fmaxf(fmaxf(b[2] - thi[2], tlo[2] - b[5]), 0.0f);
            return (dx * dx + dy * dy) + dz * dz;
        };
        auto wave_max = [&](float v) { return wave_max_f(v); };
        auto wave_min = [&](float v) { return wave_min_f(v); };
        // ---- Tile searches.  With the ray-grid hint the 64 points of a wave lie within a few millimetres of each other -- less than a
        //      triangle -- so the wave first answers the query for ONE point, the centre tc of its tile, with the lanes working on 64
        //      CANDIDATES at a time (clusters, then the triangles / vertices of the surviving clusters), and only then every lane evaluates,
        //      for its own point, the few candidates that can still matter.  Which ones can (rho = max |p - tc| over the wave, D = the
        //      minimum at tc, attained on t_D at the point q_D, u_t = the unit vector from t's closest point to tc, w = p - tc):
        //        (i)  distances are 1-Lipschitz:  d(p, t*) <= d(p, t_D) <= D + rho  and  d(tc, t*) <= d(p, t*) + rho, so the minimum t* of
        //             a lane has  d(tc, t*) - D <= 2 rho;
        //        (ii) the distance to a convex set is convex:  d(p, t) >= d(tc, t) + u_t.w,  while  d(p, t_D) <= |p - q_D| <=
        //             D + u_D.w + |w|^2 / (2 D);  so t can beat t_D at some p of the tile only if
        //             d(tc, t) - D <= |u_t - u_D| rho + rho^2 / (2 D).
        //      (ii) is what keeps tiles centimetres from the mesh cheap: there dozens of triangles lie within 2 rho of the minimum, but
        //      their u_t are nearly parallel.  `slack` adds 10 um on top of 2 rho and (ii) carries 20 um, two orders of magnitude above the
        //      fp32 rounding of a computed distance at these coordinates (~0.1 um), so every candidate whose COMPUTED distance could win or
        //      tie survives: the per-lane arg-min (ties to the lowest original index) is the exhaustive scan's, bit for bit.  A tile too
        //      large for the candidate tables (rays on the silhouette of the bounding box, samples decimetres from the mesh, launches
        //      without the hint) falls back to the per-lane search below.
        //      Measured and dropped: taking the seed cluster's best triangle as the reference of (ii) from the start, with a cylinder
        //      around each cluster as its lower bound (no search for D first): looser at every level -- 106 instead of 62 listed clusters,
        //      19 instead of 13 per-lane evaluations, 44 % instead of 71 % of the waves within the tables; 4.6 ms against 3.4.
        unsigned short* const my_list = s_list[threadIdx.x >> 6];
        const f3 tc = {0.5f * (tlo[0] + thi[0]), 0.5f * (tlo[1] + thi[1]), 0.5f * (tlo[2] + thi[2])};
        float slack;
        {
            float e2 = 0.0f;
#pragma unroll
            for (int k = 0; k < ND; ++k) {
                const float ex = pk[k].x - tc.x, ey = pk[k].y - tc.y, ez = pk[k].z - tc.z;
                e2 = fmaxf(e2, (ex * ex + ey * ey) + ez * ez);
            }
            slack = uni(2.0f * sqrtf(wave_max(e2)) * (1.0f + 1e-5f) + 1e-5f);
        }
        const float rho = 0.5f * slack;
        const bool try_tile = gnx > 0 && slack < 0.05f; // (a NaN slack compares false)
        auto sq_plus = [&](float d2) { const float r = sqrtf(d2) + slack; return (r * r) * (1.0f + 1e-4f) + 1e-12f; }; // (i), squared
        // (i) and (ii) for a candidate at distance d from tc whose unit vector differs from u_R by du (R = the reference's distance)
        auto may_win = [&](float d, float R, float du, bool have_uR) {
            const float lim = (have_uR && R > 1e-4f) ? fminf(slack, du * rho * (1.0f + 1e-4f) + (rho * rho) / (2.0f * R) * (1.0f + 1e-4f) + 2e-5f) : slack;
            return d - R <= lim;
        };
        auto mbcnt = [&](unsigned long long m) { return (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u)); };
        // appends the clusters of [0, ncl) whose box is within thr (squared) of tc to the wave's list; returns how many there are
        auto collect = [&](const float* boxes, int ncl, float thr) {
            int ns = 0;
            for (int c0 = 0; c0 < ncl; c0 += 64) {
                const int cl_ = c0 + lane_now();
                const bool keep = cl_ < ncl && box_dist2(tc, boxes + 6 * min(cl_, ncl - 1)) <= thr;
                const unsigned long long m = __ballot(keep);
                const int pos = ns + mbcnt(m);
                if (keep && pos < TL_LIST) my_list[pos] = (unsigned short)cl_;
                ns += __builtin_popcountll(m);
            }
            __builtin_amdgcn_wave_barrier();
            return ns;
        };
        float vbk[ND];
        int vik[ND];
#pragma unroll
        for (int k = 0; k < ND; ++k) { vbk[k] = INFINITY; vik[k] = 0x7fffffff; }
        auto knn_tile = [&](int cm) -> bool { // cm: the vertex cluster nearest to tc
            float U = INFINITY;
            if (lane < CL) {
                const float4 v = s_vs[cm * CL + lane];
                const float dx = tc.x - v.x, dy = tc.y - v.y, dz = tc.z - v.z;
                U = (dx * dx + dy * dy) + dz * dz;
            }
            U = wave_min(U);
            int ns = collect(s_vbox, A.nvc, sq_plus(U));
            if (ns > TL_LIST) {
                // decimetres from the mesh every vertex cluster of the near side is within 2 rho of the minimum.  As for the faces below: the exact
                // minimum at tc first (all vertices, from LDS), then only the clusters that can hold a vertex passing (ii) -- a vertex of cluster c
                // lies in the ball around its box, so its unit vector differs from the box centre's by at most 1.05 R_c / |tc - m_c|.
                float b = INFINITY;
                int sb = 0;
                for (int j = lane_now(); j < A.nvc * CL; j += 64) {
                    const float4 v = s_vs[j];
                    const float dx = tc.x - v.x, dy = tc.y - v.y, dz = tc.z - v.z;
                    const float d = (dx * dx + dy * dy) + dz * dz;
                    if (d < b) { b = d; sb = j; }
                }
                const float D2a = wave_min(b), Da = sqrtf(D2a);
                const unsigned long long mD = __ballot(b == D2a);
                if (!mD || !(Da > 1e-4f)) return false;
                const float4 va = s_vs[__builtin_amdgcn_readlane(sb, __builtin_ctzll(mD))];
                const f3 ua = {(tc.x - va.x) / Da, (tc.y - va.y) / Da, (tc.z - va.z) / Da};
                __builtin_amdgcn_wave_barrier();
                ns = 0;
                for (int c0 = 0; c0 < A.nvc; c0 += 64) {
                    const int cl_ = min(c0 + lane_now(), A.nvc - 1);
                    const float* bx = s_vbox + 6 * cl_;
                    const f3 e = {tc.x - 0.5f * (bx[0] + bx[3]), tc.y - 0.5f * (bx[1] + bx[4]), tc.z - 0.5f * (bx[2] + bx[5])};
                    const float hx = bx[3] - bx[0], hy = bx[4] - bx[1], hz = bx[5] - bx[2];
                    const float L = sqrtf(dot3(e, e)), x = (0.5f * sqrtf((hx * hx + hy * hy) + hz * hz) * (1.0f + 1e-5f) + 1e-7f) / fmaxf(L, 1e-20f);
                    float du = 2.0f;
                    if (x < 0.5f) {
                        const float inv = 1.0f / fmaxf(L, 1e-20f);
                        const float ex = e.x * inv - ua.x, ey = e.y * inv - ua.y, ez = e.z * inv - ua.z;
                        du = sqrtf((ex * ex + ey * ey) + ez * ez) + 1.05f * x + 1e-5f;
                    }
                    const bool keep = c0 + lane < A.nvc && may_win(sqrtf(box_dist2(tc, bx)), Da, du, true);
                    const unsigned long long m = __ballot(keep);
                    const int pos = ns + mbcnt(m);
                    if (keep && pos < TL_LIST) my_list[pos] = (unsigned short)cl_;
                    ns += __builtin_popcountll(m);
                }
                __builtin_amdgcn_wave_barrier();
                if (ns > TL_LIST) return false;
            }
            const int nt = ns * CL;
            float* const dit = s_dit[threadIdx.x >> 6][0] + lane; // round `it`, this lane: dit[it * 64]
            // vertex slot number 64 it + lane of the listed clusters (the list stays in LDS; lanes beyond the list read an unused entry)
            auto slot_of = [&](int it) { const int l = lane_now(); return (int)my_list[it * (64 / CL) + l / CL] * CL + l % CL; };
            // (the rounds are run-time loops: their distances live in LDS, not in register arrays, and unrolled eight-fold they made the kernel
            //  107 KB of code for a 64 KB instruction cache shared by two CUs)
#pragma nounroll
            for (int it = 0; it * 64 < nt; ++it) {
                dit[it * 64] = INFINITY;
                const int j = it * 64 + lane;
                if (j < nt) {
                    const float4 v = s_vs[slot_of(it)];
                    const float dx = tc.x - v.x, dy = tc.y - v.y, dz = tc.z - v.z;
                    dit[it * 64] = (dx * dx + dy * dy) + dz * dz;
                }
                U = fminf(U, dit[it * 64]);
            }
            const float D2 = uni(wave_min(U)), Tf = sq_plus(D2), D = sqrtf(D2);
            // reference: a vertex that attains the minimum at tc (always found: the seed cluster is on the list; if it ever were not, (i) alone decides)
            int slotD = 0;
            bool have = false;
#pragma nounroll
            for (int it = 0; it * 64 < nt && !have; ++it) { // the first round that holds the minimum
                const unsigned long long m = __ballot(dit[it * 64] == D2);
                if (m) { slotD = __builtin_amdgcn_readlane(slot_of(it), __builtin_ctzll(m)); have = true; }
            }
            const float4 vD = s_vs[slotD];
            const float invD = D > 1e-4f ? 1.0f / D : 0.0f;
            const f3 uD = {uni((tc.x - vD.x) * invD), uni((tc.y - vD.y) * invD), uni((tc.z - vD.z) * invD)};
#pragma nounroll
            for (int it = 0; it * 64 < nt; ++it) {
                {
                    bool cand = dit[it * 64] <= Tf;
                    if (cand) {
                        const float4 v = s_vs[slot_of(it)];
                        const float d = sqrtf(dit[it * 64]), inv = 1.0f / fmaxf(d, 1e-20f);
                        const float ex = (tc.x - v.x) * inv - uD.x, ey = (tc.y - v.y) * inv - uD.y, ez = (tc.z - v.z) * inv - uD.z;
                        cand = may_win(d, D, sqrtf((ex * ex + ey * ey) + ez * ez), have);
                    }
                    unsigned long long m = __ballot(cand);
                    const int sl = slot_of(it);
                    while (m) {
                        const int src = __builtin_ctzll(m);
                        m &= m - 1;
                        const float4 v = s_vs[__builtin_amdgcn_readlane(sl, src)];
                        const int oi = __float_as_int(v.w);
#pragma unroll
                        for (int k = 0; k < ND; ++k) {
                            const float dx = pk[k].x - v.x, dy = pk[k].y - v.y, dz = pk[k].z - v.z;
                            const float d = (dx * dx + dy * dy) + dz * dz;
                            if (d < vbk[k] || (d == vbk[k] && oi < vik[k])) { vbk[k] = d; vik[k] = oi; }
                        }
                    }
                }
            }
            return true;
        };
        {
            auto eval_v = [&](const f3& p, float& vb, int& vi, int c) {
                for (int k = 0; k < CL; ++k) {
                    const float4 v = s_vs[c * CL + k];
                    const float dx = p.x - v.x, dy = p.y - v.y, dz = p.z - v.z;
                    const float d = (dx * dx + dy * dy) + dz * dz;
                    const int oi = __float_as_int(v.w);
                    if (d < vb || (d == vb && oi < vi)) { vb = d; vi = oi; }
                }
            };
            // seed: the vertex cluster nearest to the centre of T, evaluated by every lane
            float smin = INFINITY;
            int cm = 0;
            for (int c = lane_now(); c < A.nvc; c += 64) {
                const float lb = box_dist2(tc, s_vbox + 6 * c);
                if (lb < smin) { smin = lb; cm = c; }
            }
            cm = wave_min_i(smin == wave_min_f(smin) ? cm : 0x7fffffff); // the nearest box, the lowest cluster among equals
            if (cm == 0x7fffffff) cm = 0;                                 // (a NaN tile)
            if (!(try_tile && knn_tile(cm))) { // per-lane search: every lane prunes against its own point and bound
#pragma unroll
                for (int k = 0; k < ND; ++k) {
                    const f3 p = pk[k];
                    float vb = INFINITY;
                    int vi = 0x7fffffff;
                    eval_v(p, vb, vi, cm);
                    const float capw = wave_max(vb) * (1.0f + 1e-4f) + 1e-12f;
                    for (int c0 = 0; c0 < A.nvc; c0 += 64) {
                        const int cl_ = c0 + lane;
                        unsigned long long m = __ballot(cl_ < A.nvc && cl_ != cm && tile_dist2(s_vbox + 6 * min(cl_, A.nvc - 1)) <= capw);
                        while (m) {
                            const int c = c0 + __builtin_ctzll(m);
                            m &= m - 1;
                            if (box_dist2(p, s_vbox + 6 * c) > vb * (1.0f + 1e-4f) + 1e-12f) continue;
                            eval_v(p, vb, vi, c);
                        }
                    }
                    vbk[k] = vb;
                    vik[k] = vi;
                }
            }
#pragma unroll
            for (int k = 0; k < ND; ++k) {
                if (vik[k] == 0x7fffffff) vik[k] = 0; // a NaN point compares false with everything: index 0, like the exhaustive scan
                bool act;
                const long long i = locate(k, act);
                if (knn && act) knn[i] = vik[k];
            }
        }
        MPH(1); // 1-NN
        // ---- closest face.  The nearest vertex belongs to some triangle, so its distance bounds the closest-face distance from
        //      above; a cluster / triangle whose LOWER bound exceeds min(best, that bound) by the safety margin cannot hold the
        //      minimum or a tie.  Lower bounds: a triangle lies in the plane through its centroid c (unit normal n), within r of c,
        //      so with e = p - c, h = n.e:  d^2 >= h^2 + max(0, sqrt(|e|^2 - h^2) - r)^2.  For a sample far from the mesh, where
        //      hundreds of triangles are nearly equidistant from p, this keeps the few it is roughly above -- the bounding-sphere
        //      bound (|e| - r)^2 it replaces kept every triangle of a patch of radius ~sqrt(2 d r) (5x the exact evaluations).
        //      Rounding: n and the dot products are off by <= 1e-6 |e|, the stored centres by <= a = 1e-6 x (largest
        //      coordinate) -- the radii are padded by a on the host -- so |h| is off by delta <= 1e-6 |e| + a and
        //      h^2 by 2 |h| delta <= kappa |e|^2 + w with w = 1e5 a^2 (tnorm.w); h^2 is lowered by that much where it adds
        //      to the bound and raised where it is subtracted.
        constexpr float kappa = 3e-5f;
        float bestk[ND];
        int bfk[ND];
#pragma unroll
        for (int k = 0; k < ND; ++k) { bestk[k] = INFINITY; bfk[k] = 0x7fffffff; }
        // disc lower bound (squared) of triangle t for point q
        auto disc_lb2 = [&](const float4 sp, const float4 tn, f3 q) {
            const float ex = q.x - sp.x, ey = q.y - sp.y, ez = q.z - sp.z;
            const float e2 = (ex * ex + ey * ey) + ez * ez;
            const float h = (tn.x * ex + tn.y * ey) + tn.z * ez;
            const float h2 = h * h;
            const float gr = fmaxf(sqrtf(fmaxf(e2 * (1.0f - kappa) - h2 - tn.w, 0.0f)) - sp.w, 0.0f);
            return fmaxf(h2 - kappa * e2 - tn.w, 0.0f) + gr * gr;
        };
        // The records of a cluster come through wave-uniform (scalar) loads: the cluster index is the same in every lane.  All CL bound
        // records are fetched before the first is used (one scalar-memory latency per cluster instead of one per triangle).
        auto eval_cluster = [&](const f3& p, float& best, int& bf, const float vb, int c) {
            float4 sp[CL], tn[CL];
#pragma unroll
            for (int k = 0; k < CL; ++k) {
                sp[k] = reinterpret_cast<const float4*>(A.sphere)[c * CL + k];
                tn[k] = reinterpret_cast<const float4*>(A.tnorm)[c * CL + k];
            }
#pragma unroll
            for (int k = 0; k < CL; ++k) {
                if (disc_lb2(sp[k], tn[k], p) > fminf(best, vb) * (1.0f + 1e-4f) + 1e-12f) continue;
                const int t = c * CL + k;
                const float* q = A.tri + (size_t)t * 9;
                const f3 a = {q[0], q[1], q[2]}, b = {q[3], q[4], q[5]}, c3 = {q[6], q[7], q[8]};
                const float d = point_tri_dist2(p, a, b, c3);
#ifdef VANERF_MESH_PHASES
                if (__builtin_ctzll(__ballot(1)) == lane) ph[7] += 1; // exact evaluations (some lane)
#endif
                const int of = A.orig[t];
                if (d < best || (d == best && of < bf)) { best = d; bf = of; }
            }
        };
        auto face_tile = [&](int cseed) -> bool { // cseed: the triangle cluster nearest to tc
            struct Tri { f3 a, b, c; };
            auto load_tri = [&](int t) {
                const float* r = A.tri + (size_t)t * 9;
                return Tri{{r[0], r[1], r[2]}, {r[3], r[4], r[5]}, {r[6], r[7], r[8]}};
            };
            float U = INFINITY;
            if (lane < CL) { const Tri T = load_tri(cseed * CL + lane); U = point_tri_dist2(tc, T.a, T.b, T.c); }
            U = wave_min(U);
            const float thr = sq_plus(U);
            // clusters within thr of tc by BOTH lower bounds: the axis-aligned box (LDS) and the cylinder around the cluster (axis = mean normal
            // through the mean vertex; global table, one cluster per lane).  For a tile centimetres from the mesh the box alone is too loose --
            // a corner of it sticks out towards the tile by up to half its diagonal -- and lists every cluster of the near side.
            int ns = 0;
            for (int c0 = 0; c0 < A.nc; c0 += 64) {
                const int cl_ = min(c0 + lane_now(), A.nc - 1);
                bool keep = c0 + lane < A.nc && box_dist2(tc, s_box + 6 * cl_) <= thr;
                if (keep) {
                    const float4 cd = reinterpret_cast<const float4*>(A.cdisc)[2 * cl_], cn = reinterpret_cast<const float4*>(A.cdisc)[2 * cl_ + 1];
                    const f3 e = {tc.x - cd.x, tc.y - cd.y, tc.z - cd.z};
                    const float L2 = dot3(e, e), h = (cn.x * e.x + cn.y * e.y) + cn.z * e.z;
                    const float dh = fmaxf(fabsf(h) * (1.0f - 1e-5f) - cn.w, 0.0f), dl = fmaxf(sqrtf(fmaxf(L2 - h * h, 0.0f)) * (1.0f - 1e-5f) - cd.w, 0.0f);
                    keep = (dh * dh + dl * dl) * (1.0f - 1e-5f) <= thr;
                }
                const unsigned long long m = __ballot(keep);
                const int pos = ns + mbcnt(m);
                if (keep && pos < TL_LIST) my_list[pos] = (unsigned short)cl_;
                ns += __builtin_popcountll(m);
            }
            __builtin_amdgcn_wave_barrier();
            float thr_eval = thr;
            if (ns > TL_LIST) {
                // A tile decimetres from the mesh: hundreds of clusters lie within 2 rho of the seed's distance.  Two stages instead.
                // 1. the exact minimum D at tc: only clusters within the seed's distance U itself can hold it (a short list);
                // 2. the clusters that can hold a triangle passing (ii): every closest point of cluster c lies in the ball (m_c, R_c) around
                //    its cylinder, so  |u_t - u_D| <= |u_m - u_D| + 1.05 R_c / |tc - m_c|  (R_c < |tc - m_c| / 2; chord <= angle <= 1.05 sine),
                //    and d(tc, t) >= the cluster's lower bound: c matters only if  bound_c - D  passes may_win with that difference.
                auto cyl = [&](int cl_, float& lb2, float& du) { // lower bound (squared) of cluster cl_ at tc; du needs uD (stage 2)
                    const float4 cd = reinterpret_cast<const float4*>(A.cdisc)[2 * cl_], cn = reinterpret_cast<const float4*>(A.cdisc)[2 * cl_ + 1];
                    const f3 e = {tc.x - cd.x, tc.y - cd.y, tc.z - cd.z};
                    const float L2 = dot3(e, e), h = (cn.x * e.x + cn.y * e.y) + cn.z * e.z;
                    const float dh = fmaxf(fabsf(h) * (1.0f - 1e-5f) - cn.w, 0.0f), dl = fmaxf(sqrtf(fmaxf(L2 - h * h, 0.0f)) * (1.0f - 1e-5f) - cd.w, 0.0f);
                    lb2 = fmaxf(box_dist2(tc, s_box + 6 * cl_), (dh * dh + dl * dl) * (1.0f - 1e-5f));
                    du = sqrtf(cd.w * cd.w + cn.w * cn.w) * (1.0f + 1e-5f) / fmaxf(sqrtf(L2), 1e-20f); // R_c / |tc - m_c|
                    return e;
                };
                auto list_clusters = [&](auto keep_fn) {
                    int n_ = 0;
                    for (int c0 = 0; c0 < A.nc; c0 += 64) {
                        const int cl_ = min(c0 + lane_now(), A.nc - 1);
                        const bool keep = c0 + lane < A.nc && keep_fn(cl_);
                        const unsigned long long m = __ballot(keep);
                        const int pos = n_ + mbcnt(m);
                        if (keep && pos < TL_LIST) my_list[pos] = (unsigned short)cl_;
                        n_ += __builtin_popcountll(m);
                    }
                    __builtin_amdgcn_wave_barrier();
                    return n_;
                };
                const float thr1 = U * (1.0f + 1e-4f) + 1e-12f;
                __builtin_amdgcn_wave_barrier(); // (the first collect's writers are done with the list)
                const int n1 = list_clusters([&](int cl_) { float lb2, du; cyl(cl_, lb2, du); return lb2 <= thr1; });
                float b1 = INFINITY;
                int t1 = cseed * CL;
                if (lane < CL) { const Tri T = load_tri(cseed * CL + lane); b1 = point_tri_dist2(tc, T.a, T.b, T.c); t1 = cseed * CL + lane; }
                if (n1 <= TL_LIST) {
                    for (int j = lane_now(); j < n1 * CL; j += 64) {
                        const int t = (int)my_list[j / CL] * CL + j % CL;
                        const float4 sp = reinterpret_cast<const float4*>(A.sphere)[t], tn = reinterpret_cast<const float4*>(A.tnorm)[t];
                        if (disc_lb2(sp, tn, tc) <= thr1) {
                            const Tri T = load_tri(t);
                            const float d = point_tri_dist2(tc, T.a, T.b, T.c);
                            if (d < b1) { b1 = d; t1 = t; }
                        }
                    }
                } else { // (a flat part of the mesh faces the tile: hundreds of clusters are as near as the seed) every triangle, 64 at a time
                    float bw = thr1;
                    for (int t0 = 0; t0 < A.nfp; t0 += 64) {
                        const int t = min(t0 + lane, A.nfp - 1);
                        const float4 sp = reinterpret_cast<const float4*>(A.sphere)[t], tn = reinterpret_cast<const float4*>(A.tnorm)[t];
                        if (disc_lb2(sp, tn, tc) <= bw) {
                            const Tri T = load_tri(t);
                            const float d = point_tri_dist2(tc, T.a, T.b, T.c);
                            if (d < b1) { b1 = d; t1 = t; }
                        }
                        bw = fminf(bw, wave_min(b1) * (1.0f + 1e-4f) + 1e-12f);
                    }
                }
                const float D2a = wave_min(b1), Da = sqrtf(D2a);
                const unsigned long long mD = __ballot(b1 == D2a);
                if (!mD || !(Da > 1e-4f)) return false; // (a NaN tile; a tile this close to the mesh does not overflow the list)
                f3 qa;
                { const Tri T = load_tri(__builtin_amdgcn_readlane(t1, __builtin_ctzll(mD))); point_tri_dist2_q(tc, T.a, T.b, T.c, qa); }
                const f3 ua = {(tc.x - qa.x) / Da, (tc.y - qa.y) / Da, (tc.z - qa.z) / Da};
                __builtin_amdgcn_wave_barrier(); // (stage 1's readers are done with the list)
                ns = list_clusters([&](int cl_) {
                    float lb2, x;
                    const f3 e = cyl(cl_, lb2, x);
                    float du = 2.0f; // |u - u'| <= 2 always
                    if (x < 0.5f) {
                        const float inv = 1.0f / fmaxf(sqrtf(dot3(e, e)), 1e-20f);
                        const float ex = e.x * inv - ua.x, ey = e.y * inv - ua.y, ez = e.z * inv - ua.z;
                        du = sqrtf((ex * ex + ey * ey) + ez * ez) + 1.05f * x + 1e-5f;
                    }
                    return may_win(sqrtf(lb2), Da, du, true);
                });
                thr_eval = sq_plus(D2a);
            }
#ifdef VANERF_MESH_PHASES
            if (lane == 0 && ns > TL_LIST) ph[14] += 1; // list overflow
#endif
            if (ns > TL_LIST) return false;
#ifdef VANERF_MESH_PHASES
            if (lane == 0) ph[12] += ns; // clusters on the list
#endif
            const int nt = ns * CL;
            // round `it`: lane l looks at triangle number 64 it + l of the listed clusters.  First the disc bounds of all rounds (their loads
            // are independent and overlap), then the exact distances at tc of the triangles whose bound is within the seed's threshold.
            float* const dit = s_dit[threadIdx.x >> 6][0] + lane; // round `it`, this lane: dit[it * 64]
            // triangle number 64 it + lane of the listed clusters (the list stays in LDS; lanes beyond the list read an unused entry)
            auto tri_of = [&](int it) { const int l = lane_now(); return (int)my_list[it * (64 / CL) + l / CL] * CL + l % CL; };
#pragma nounroll
            for (int it = 0; it * 64 < nt; ++it) {
                dit[it * 64] = INFINITY;
                const int j = it * 64 + lane;
                if (j < nt) {
                    const int t = tri_of(it);
                    const float4 sp = reinterpret_cast<const float4*>(A.sphere)[t], tn = reinterpret_cast<const float4*>(A.tnorm)[t];
                    dit[it * 64] = disc_lb2(sp, tn, tc) <= thr_eval ? 0.0f : INFINITY; // 0 = "evaluate me"
                }
            }
#pragma nounroll
            for (int it = 0; it * 64 < nt; ++it) {
                if (dit[it * 64] == 0.0f) { const Tri T = load_tri(tri_of(it)); dit[it * 64] = point_tri_dist2(tc, T.a, T.b, T.c); }
                U = fminf(U, dit[it * 64]);
            }
            const float D2 = uni(wave_min(U)), Tf = sq_plus(D2), D = sqrtf(D2);
            // reference: (one of) the triangle(s) that attain the minimum at tc, with its closest point q_D
            int tD = cseed * CL;
            bool have = false;
#pragma nounroll
            for (int it = 0; it * 64 < nt && !have; ++it) { // the first round that holds the minimum
                const unsigned long long m = __ballot(dit[it * 64] == D2);
                if (m) { tD = __builtin_amdgcn_readlane(tri_of(it), __builtin_ctzll(m)); have = true; }
            }
            f3 qD;
            { const Tri T = load_tri(tD); point_tri_dist2_q(tc, T.a, T.b, T.c, qD); }
            const float invD = D > 1e-4f ? 1.0f / D : 0.0f;
            const f3 uD = {uni((tc.x - qD.x) * invD), uni((tc.y - qD.y) * invD), uni((tc.z - qD.z) * invD)};
            // the triangles that pass (i) get a second look (their closest point, for (ii)); those that pass both go to the wave's LDS
            // candidate table, from where every lane evaluates them for its own point
            float* const ctab = s_cand[threadIdx.x >> 6][0];
            int K = 0;
            unsigned cbits = 0u; // bit `it`: this lane's triangle of round `it` is a candidate
#pragma nounroll
            for (int it = 0; it * 64 < nt; ++it) {
                if (__ballot(dit[it * 64] <= Tf)) {
                    bool cand = dit[it * 64] <= Tf;
                    Tri T = {};
                    if (cand) {
                        T = load_tri(tri_of(it));
                        f3 qt;
                        point_tri_dist2_q(tc, T.a, T.b, T.c, qt);
                        const float d = sqrtf(dit[it * 64]), inv = 1.0f / fmaxf(d, 1e-20f);
                        const float ex = (tc.x - qt.x) * inv - uD.x, ey = (tc.y - qt.y) * inv - uD.y, ez = (tc.z - qt.z) * inv - uD.z;
                        cand = may_win(d, D, sqrtf((ex * ex + ey * ey) + ez * ez), have);
                    }
                    const unsigned long long m = __ballot(cand);
                    cbits |= cand ? 1u << it : 0u;
                    const int pos = K + mbcnt(m);
                    if (cand && pos < TL_CAND) {
                        float* e = ctab + pos * 12;
                        e[0] = T.a.x; e[1] = T.a.y; e[2] = T.a.z; e[3] = T.b.x; e[4] = T.b.y; e[5] = T.b.z; e[6] = T.c.x; e[7] = T.c.y; e[8] = T.c.z;
                        e[9] = __int_as_float(A.orig[tri_of(it)]);
                    }
                    K += __builtin_popcountll(m);
                }
            }
#ifndef VANERF_TL_KMAX
#define VANERF_TL_KMAX 100000
#endif
            if (K > VANERF_TL_KMAX) return false; // (a tile on the surface: every triangle within 2 rho is a candidate; the per-lane search prunes those by each lane's own bound)
#ifdef VANERF_MESH_PHASES
            if (lane == 0 && K > TL_CAND) ph[15] += 1; // more candidates than the table holds: evaluated in batches
            if (lane == 0) ph[11] += K; // per-lane evaluations of the tile search
#endif
            // every lane evaluates the candidates for its own point, a table-full at a time (a tile facing a flat part of the mesh from
            // decimetres away has ~60: any triangle under the tile's footprint can be the closest of one of its points)
            for (int base = 0; base < K; base += TL_CAND) {
                if (base > 0) {
                    __builtin_amdgcn_wave_barrier(); // the previous batch's readers are done
                    int k0 = 0;
#pragma nounroll
                    for (int it = 0; it * 64 < nt; ++it) {
                        const unsigned long long m = __ballot((cbits >> it) & 1u);
                        if (m) {
                            const int pos = k0 + mbcnt(m) - base;
                            if (((m >> lane) & 1ull) && pos >= 0 && pos < TL_CAND) {
                                const int t = tri_of(it);
                                const Tri T = load_tri(t);
                                float* e = ctab + pos * 12;
                                e[0] = T.a.x; e[1] = T.a.y; e[2] = T.a.z; e[3] = T.b.x; e[4] = T.b.y; e[5] = T.b.z; e[6] = T.c.x; e[7] = T.c.y; e[8] = T.c.z;
                                e[9] = __int_as_float(A.orig[t]);
                            }
                            k0 += __builtin_popcountll(m);
                        }
                    }
                }
                __builtin_amdgcn_wave_barrier();
                const int kn = min(TL_CAND, K - base);
#pragma unroll
                for (int q = 0; q < ND; ++q) { // one point after the other: the registers of one exact distance at a time
                    __builtin_amdgcn_sched_barrier(0);
                    const f3 p = pk[q];
                    float best = bestk[q];
                    int bf = bfk[q];
                    for (int k = 0; k < kn; ++k) {
                        const float4 e0 = reinterpret_cast<const float4*>(ctab + k * 12)[0], e1 = reinterpret_cast<const float4*>(ctab + k * 12)[1],
                                     e2 = reinterpret_cast<const float4*>(ctab + k * 12)[2];
                        const f3 a = {e0.x, e0.y, e0.z}, b = {e0.w, e1.x, e1.y}, c3 = {e1.z, e1.w, e2.x};
                        const float d = point_tri_dist2(p, a, b, c3);
                        const int of = __float_as_int(e2.y);
                        if (d < best || (d == best && of < bf)) { best = d; bf = of; }
                    }
                    bestk[q] = best;
                    bfk[q] = bf;
                }
            }
            return true;
        };
        // (1) seed: the cluster nearest to the centre of T, all of its triangles, so that every lane holds a bound close to its answer
        int cseed = 0;
        {
            float smin = INFINITY;
            for (int c = lane_now(); c < A.nc; c += 64) {
                const float lb = box_dist2(tc, s_box + 6 * c);
                if (lb < smin) { smin = lb; cseed = c; }
            }
            cseed = wave_min_i(smin == wave_min_f(smin) ? cseed : 0x7fffffff); // the nearest box, the lowest cluster among equals
            if (cseed == 0x7fffffff) cseed = 0;                                 // (a NaN tile)
        }
        const bool tiled = try_tile && face_tile(cseed);
#ifdef VANERF_MESH_PHASES
        if (lane == 0 && !try_tile) ph[13] += 1;
        if (lane == 0) { ph[tiled ? 8 : 9] += 1; const unsigned long long t_ = __builtin_amdgcn_s_memtime(); if (tiled) ph[10] += t_ - tprev; }
#endif
        if (!tiled) {
#pragma unroll
            for (int k = 0; k < ND; ++k) {
                const f3 p = pk[k];
                const float vb = vbk[k];
                float best = INFINITY;
                int bf = 0x7fffffff;
                eval_cluster(p, best, bf, vb, cseed);
                // (2) the other clusters: 64 at a time against T with the wave's largest bound, the survivors by every lane against its own
                const float capf = wave_max(fminf(best, vb)) * (1.0f + 1e-4f) + 1e-12f;
                for (int c0 = 0; c0 < A.nc; c0 += 64) {
                    const int cl_ = c0 + lane;
                    unsigned long long cm_ = __ballot(cl_ < A.nc && cl_ != cseed && tile_dist2(s_box + 6 * min(cl_, A.nc - 1)) <= capf);
                    while (cm_) {
                        const int c = c0 + __builtin_ctzll(cm_);
                        cm_ &= cm_ - 1;
#ifdef VANERF_MESH_PHASES
                        if (lane == 0) ph[5] += 1; // clusters surviving the wave-level test
#endif
                        if (box_dist2(p, s_box + 6 * c) > fminf(best, vb) * (1.0f + 1e-4f) + 1e-12f) continue;
#ifdef VANERF_MESH_PHASES
                        if (__builtin_ctzll(__ballot(1)) == lane) ph[6] += 1; // ... whose triangles some lane looks at
#endif
                        eval_cluster(p, best, bf, vb, c);
                    }
                }
                bestk[k] = best;
                bfk[k] = bf;
            }
        }
        MPH(2); // closest face
#pragma unroll
        for (int k = 0; k < ND; ++k) {
        __builtin_amdgcn_sched_barrier(0); // one point after the other (no interleaving of the unrolled copies: registers)
        const f3 p = pk[k];
        bool act;
        const long long i = locate(k, act);
        const float best = bestk[k];
        // a point without a finite distance (NaN or infinite coordinates) answers face 0, as the exhaustive scan's strict `d < best` leaves it; the
        // sentinel never indexes the face table
        const int bf = (bfk[k] == 0x7fffffff || !(best < INFINITY)) ? 0 : bfk[k];
        // inside test on the (y,z) grid
        int cnt = 0;
        {
            const float fy = floorf((p.y - grid_y0) / grid_cy), fz = floorf((p.z - grid_z0) / grid_cz);
            int cy = (int)fy, cz = (int)fz;
            cy = min(max(cy, 0), grid_G - 1);
            cz = min(max(cz, 0), grid_G - 1);
            const int cell = cy * grid_G + cz;
            // A point more than a whole cell outside the mesh's (y,z) extent lies beside every triangle by ~1 mm or more: its edge functions
            // (|edge| x distance against 1e-7 relative rounding) cannot all agree in sign, so the exhaustive scan counts no crossing either.
            // (Points within a cell of the border keep going through the border cells' lists, as before.)
            const bool beside = fy < -1.0f || fy > (float)grid_G || fz < -1.0f || fz > (float)grid_G;
            const int e = beside ? 0 : A.cell_start[cell + 1];
            for (int k = beside ? 0 : A.cell_start[cell]; k < e; ++k) {
                // one 48-byte record per list entry (vertex ids for the canonical edge orientation, then the corners: copies of F and V), so that
                // an entry costs one independent load instead of the chain face id -> vertex ids -> corners
                const float4 r0 = reinterpret_cast<const float4*>(A.cell_rec)[3 * k], r1 = reinterpret_cast<const float4*>(A.cell_rec)[3 * k + 1],
                             r2 = reinterpret_cast<const float4*>(A.cell_rec)[3 * k + 2];
                const int i0 = __float_as_int(r0.x), i1 = __float_as_int(r0.y), i2 = __float_as_int(r0.z);
                const f3 a = {r0.w, r1.x, r1.y}, b = {r1.z, r1.w, r2.x}, c3 = {r2.y, r2.z, r2.w};
                float E0, E1, E2;
                const bool s0 = edge_side(b, c3, i1, i2, p.y, p.z, E0);
                const bool s1 = edge_side(c3, a, i2, i0, p.y, p.z, E1);
                const bool s2 = edge_side(a, b, i0, i1, p.y, p.z, E2);
                if ((s0 && s1 && s2) || (!s0 && !s1 && !s2)) {
                    const float den = (E0 + E1) + E2;
                    if (den != 0.0f) {
                        const float xh = ((E0 * a.x + E1 * b.x) + E2 * c3.x) / den;
                        if (xh > p.x) ++cnt;
                    }
                }
            }
        }
        MPH(3); // inside test
        const float dist = sqrtf(best + 1e-6f);
        if (act) sdf[i] = (cnt & 1) ? -dist : dist;
        if (face && act) face[i] = bf;
        const int i0 = F[3 * bf], i1 = F[3 * bf + 1], i2 = F[3 * bf + 2];
        const f3 v0 = {V[3 * i0], V[3 * i0 + 1], V[3 * i0 + 2]}, v1 = {V[3 * i1], V[3 * i1 + 1], V[3 * i1 + 2]},
                 v2 = {V[3 * i2], V[3 * i2 + 1], V[3 * i2 + 2]};
        const f3 u = sub3(v1, v0), v = sub3(v2, v0);
        const f3 nrm = {u.y * v.z - u.z * v.y, u.z * v.x - u.x * v.z, u.x * v.y - u.y * v.x};
        float s = dot3(nrm, nrm);
        if (s == 0.0f) s = 1e-6f;
        const float inv = 1.0f / s;
        const f3 wv = sub3(p, v0);
        const f3 c1 = {u.y * wv.z - u.z * wv.y, u.z * wv.x - u.x * wv.z, u.x * wv.y - u.y * wv.x};
        const f3 c2 = {wv.y * v.z - wv.z * v.y, wv.z * v.x - wv.x * v.z, wv.x * v.y - wv.y * v.x};
        const float b2 = dot3(c1, nrm) * inv;
        const float b1 = dot3(c2, nrm) * inv;
        const float w0 = (1.0f - b1) - b2;
        const float sv = (w0 * vert_vis[i0] + b1 * vert_vis[i1]) + b2 * vert_vis[i2];
        if (act) vis[i] = sv >= 0.1f;
        }
        MPH(4); // visibility + stores
    }
#ifdef VANERF_MESH_PHASES
    if ((threadIdx.x & 63) == 0)
        for (int k = 0; k < 16; ++k) atomicAdd(&g_ma_phase[k], ph[k]);
#endif
}

} // namespace

extern "C" int vanerf_vertex_visibility(const float* vert_xy01, const float* vert_z01, int nv, const int32_t* faces, int nf,
                                        int raster, int32_t* pix_to_face, float* vert_vis, void* stream)
{
    return guarded([&] {
        if (!vert_xy01 || !vert_z01 || !faces || !pix_to_face || !vert_vis) throw_error("vanerf_vertex_visibility: null argument");
        if (nv <= 0 || nf <= 0 || raster <= 0 || raster > 4096) throw_error("vanerf_vertex_visibility: nv=%d nf=%d raster=%d", nv, nf, raster);
        hipStream_t st = (hipStream_t)stream;
        const int npix = raster * raster;
        HIP_CHECK(hipMemsetAsync(vert_vis, 0, sizeof(float) * nv, st));
        const int tiles = (raster + RV_T - 1) / RV_T;
        hipLaunchKernelGGL(raster_kernel, dim3((unsigned)(tiles * tiles)), dim3(RV_BLOCK), 0, st, vert_xy01, vert_z01, faces, nf, raster, pix_to_face);
        hipLaunchKernelGGL(mark_visible_kernel, dim3((npix + 255) / 256), dim3(256), 0, st, pix_to_face, npix, faces, nf, vert_vis);
        HIP_CHECK(hipGetLastError());
    });
}

extern "C" int vanerf_mesh_query(const float* verts, int nv, const int32_t* faces, int nf, const float* vert_vis, const float* pts,
                                 int64_t n, float* sdf, uint8_t* vis, int32_t* face, void* stream)
{
    return guarded([&] {
        if (n == 0) return;
        if (!verts || !faces || !vert_vis || !pts || !sdf || !vis) throw_error("vanerf_mesh_query: null argument");
        if (nv <= 0 || nf <= 0 || n < 0) throw_error("vanerf_mesh_query: nv=%d nf=%d n=%lld", nv, nf, (long long)n);
        if (n == 0) return;
        hipLaunchKernelGGL(mesh_query_kernel, dim3((unsigned)((n + MQ_BLOCK - 1) / MQ_BLOCK)), dim3(MQ_BLOCK), 0, (hipStream_t)stream,
                           verts, faces, nf, vert_vis, pts, (long long)n, sdf, vis, face);
        HIP_CHECK(hipGetLastError());
    });
}

extern "C" int vanerf_knn1(const float* verts4, int nv, const float* pts, int64_t n, int32_t* idx, void* stream)
{
    return guarded([&] {
        if (n == 0) return;
        if (!verts4 || !pts || !idx) throw_error("vanerf_knn1: null argument");
        if (nv <= 0 || nv > 8192 || n < 0) throw_error("vanerf_knn1: nv=%d n=%lld", nv, (long long)n);
        if (n == 0) return;
        hipLaunchKernelGGL(knn1_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), sizeof(float4) * nv, (hipStream_t)stream,
                           reinterpret_cast<const float4*>(verts4), nv, pts, (long long)n, idx);
        HIP_CHECK(hipGetLastError());
    });
}

extern "C" int vanerf_mesh_query_accel(const VanerfMeshAccel* accel, const float* verts, int nv, const int32_t* faces, int nf,
                                       const float* vert_vis, const float* pts, int64_t n, float* sdf, uint8_t* vis, int32_t* face,
                                       int32_t* knn_idx, int grid_nx, int grid_ny, int grid_s, void* queue_word, void* stream)
{
    return guarded([&] {
        if (n == 0) return; // an empty batch is valid (and has null data pointers)
        if (!accel || !verts || !faces || !vert_vis || !pts || !sdf || !vis) throw_error("vanerf_mesh_query_accel: null argument");
        const VanerfMeshAccel& A = *accel;
        if (!A.tri || !A.sphere || !A.tnorm || !A.orig || !A.cbox || !A.cdisc || !A.cell_start || !A.cell_tri || !A.cell_rec || !A.grid) throw_error("vanerf_mesh_query_accel: accel has a null pointer");
        if (A.nc <= 0 || A.nc > MA_MAX_CLUSTERS || A.nfp != A.nc * CL || A.nfp < nf) throw_error("vanerf_mesh_query_accel: bad cluster table (nc=%d nfp=%d nf=%d)", A.nc, A.nfp, nf);
        if (nv <= 0 || nf <= 0 || n < 0) throw_error("vanerf_mesh_query_accel: nv=%d nf=%d n=%lld", nv, nf, (long long)n);
        if (!A.vsort || !A.vbox || A.nvc <= 0 || A.nvc > MA_MAX_VCLUSTERS || A.nvc * CL < nv)
            throw_error("vanerf_mesh_query_accel: bad vertex cluster table (nvc=%d nv=%d)", A.nvc, nv);
        const size_t lds = sizeof(float) * ((size_t)A.nvc * CL * 4 + (size_t)A.nvc * 6 + (size_t)A.nc * 6);
        if (lds > 140 * 1024) throw_error("vanerf_mesh_query_accel: mesh too large for the LDS-resident tables (%zu bytes)", lds);
        if (lds > 64 * 1024) { // above the default dynamic-LDS limit (gfx950 has 160 KB per CU)
            HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(mesh_query_accel_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(mesh_query_accel_kernel<ND_MAX>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        }
        if (n == 0) return;
        if (grid_nx != 0 && (grid_nx < 0 || grid_ny <= 0 || grid_s <= 0 || (long long)grid_nx * grid_ny * grid_s != n))
            throw_error("vanerf_mesh_query_accel: ray-grid hint %d x %d x %d does not match n = %lld", grid_nx, grid_ny, grid_s, (long long)n);
        if (n >= (1ll << 36)) throw_error("vanerf_mesh_query_accel: n = %lld points in one launch (limit 2^36)", (long long)n);
        // as many blocks as the chip holds at once (the work queue hands out the points; its head is the caller's queue_word, zeroed on the stream)
        if (!queue_word || (reinterpret_cast<uintptr_t>(queue_word) & 7u)) throw_error("vanerf_mesh_query_accel: queue_word must be 8 bytes of device memory, 8-byte aligned");
        struct PerDevice { std::atomic<size_t> lds{~(size_t)0}; std::atomic<int> resident{0}; };
        static PerDevice per_device[64]; // host-side cache of the occupancy query, per device (and again when the mesh, hence the LDS footprint, changes)
        int dev = 0;
        HIP_CHECK(hipGetDevice(&dev));
        PerDevice& pd = per_device[dev & 63];
        if (pd.resident.load() == 0 || pd.lds.load() != lds) {
            int cus = 256, per_cu = 1;
            HIP_CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
            HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, mesh_query_accel_kernel<ND_MAX>, MA_BLOCK, lds));
            pd.resident.store(cus * (per_cu > 0 ? per_cu : 1));
            pd.lds.store(lds);
        }
        const int resident = pd.resident.load();
        unsigned long long* const queue = static_cast<unsigned long long*>(queue_word);
        HIP_CHECK(hipMemsetAsync(queue, 0, sizeof(unsigned long long), (hipStream_t)stream));
        long long blocks = (n + MA_BLOCK - 1) / MA_BLOCK;
        if (blocks > resident) blocks = resident;
        // Several depths per item only when every wave gets a handful of them: below that a launch is bound by the latency of single items, and
        // twice as many lighter items run better (a 128x128 view at 32 samples: 4 096 items of two depths for 4 096 waves).
        const long long waves = (long long)resident * (MA_BLOCK / 64), items_nd = (n + 64 * ND_MAX - 1) / (64 * ND_MAX);
        if (ND_MAX > 1 && items_nd >= 4 * waves)
            hipLaunchKernelGGL(mesh_query_accel_kernel<ND_MAX>, dim3((unsigned)blocks), dim3(MA_BLOCK), lds, (hipStream_t)stream, A, verts, faces, vert_vis,
                               pts, (long long)n, sdf, vis, face, knn_idx, grid_nx, grid_ny, grid_s, queue);
        else
            hipLaunchKernelGGL(mesh_query_accel_kernel<1>, dim3((unsigned)blocks), dim3(MA_BLOCK), lds, (hipStream_t)stream, A, verts, faces, vert_vis,
                               pts, (long long)n, sdf, vis, face, knn_idx, grid_nx, grid_ny, grid_s, queue);
        HIP_CHECK(hipGetLastError());
    });
}

#ifdef VANERF_MESH_PHASES
extern "C" int vanerf_debug_mesh_phases(unsigned long long* out8, int reset)
{
    return guarded([&] {
        HIP_CHECK(hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_ma_phase), sizeof(unsigned long long) * 16));
        if (reset) { unsigned long long z[16] = {}; HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_ma_phase), z, sizeof z)); }
    });
}
#endif

extern "C" int vanerf_mesh_cluster_size(void) { return CL; }

// ---------------------------------------------------------------------------------------------------------------
// vanerf_mesh_accel_build: the tables of VanerfMeshAccel, built on the device without a host synchronisation (one per source frame).
// Seven small launches on the caller's stream:
//   bounds_sort   (2 blocks) vertex bounds -> grid record; Morton keys of triangle centroids (block 0) and of vertices (block 1), bitonic sort in LDS
//   tables        per triangle cluster: sorted corners, bounding spheres / discs, cluster AABB and cylinder; per vertex cluster: vsort, vbox
//   cell_count    per triangle: +1 in every (y,z) cell its bounding box touches
//   cell_scan     (1 block)  CSR offsets; if the lists would not fit `cell_capacity` the grid degrades to ONE cell holding every triangle
//                 (the inside test then scans all of them: slower, same answer)
//   cell_fill     per triangle: its id into the cells' lists (atomic cursors)
//   cell_order    one wave per cell: ascending ids (a deterministic table whatever the order of the atomics)
//   cell_record   per list entry: the triangle's vertex ids and corners, 48 bytes (what the inside test reads)
// Every bound carries the slack of the fp32 arithmetic that produced it (1e-5 relative + 1e-6 of the largest coordinate), as the pruning
// tests of mesh_query_accel_kernel assume.
namespace {

constexpr int AB_THREADS = 1024;

__device__ __forceinline__ unsigned part1by2(unsigned x)
{
    x = (x | (x << 16)) & 0x030000FFu;
    x = (x | (x << 8)) & 0x0300F00Fu;
    x = (x | (x << 4)) & 0x030C30C3u;
    return (x | (x << 2)) & 0x09249249u;
}

__device__ __forceinline__ unsigned morton30(f3 c, f3 lo, f3 hi)
{
    auto q = [](float v, float l, float h) { return (unsigned)fminf(fmaxf((v - l) / (h - l + 1e-9f) * 1023.0f, 0.0f), 1023.0f); };
    return part1by2(q(c.x, lo.x, hi.x)) | (part1by2(q(c.y, lo.y, hi.y)) << 1) | (part1by2(q(c.z, lo.z, hi.z)) << 2);
}

__device__ __forceinline__ f3 vert3(const float* V, int i) { return {V[3 * i], V[3 * i + 1], V[3 * i + 2]}; }

__device__ void bitonic_sort_lds(unsigned long long* s, int n)
{
    for (int k = 2; k <= n; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < n; i += AB_THREADS) {
                const int o = i ^ j;
                if (o > i) {
                    const unsigned long long a = s[i], b = s[o];
                    if ((a > b) == ((i & k) == 0)) { s[i] = b; s[o] = a; }
                }
            }
            __syncthreads();
        }
}

__global__ __launch_bounds__(AB_THREADS) void accel_bounds_sort_kernel(const float* __restrict__ V, int nv, const int32_t* __restrict__ F, int nf, int G,
                                                                       int nt2, int nv2, float* __restrict__ grid, int32_t* __restrict__ tri_order,
                                                                       int32_t* __restrict__ vert_order, int32_t* __restrict__ counts)
{
    extern __shared__ unsigned long long s_keys[];
    __shared__ float s_red[6][AB_THREADS / 64];
    __shared__ float s_b[6];
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int i = threadIdx.x; i < nv; i += AB_THREADS)
        for (int a = 0; a < 3; ++a) {
            const float v = V[3 * i + a];
            lo[a] = fminf(lo[a], v);
            hi[a] = fmaxf(hi[a], v);
        }
    for (int a = 0; a < 3; ++a)
        for (int o = 32; o > 0; o >>= 1) {
            lo[a] = fminf(lo[a], __shfl_xor(lo[a], o));
            hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], o));
        }
    if ((threadIdx.x & 63) == 0)
        for (int a = 0; a < 3; ++a) {
            s_red[a][threadIdx.x >> 6] = lo[a];
            s_red[3 + a][threadIdx.x >> 6] = hi[a];
        }
    if (blockIdx.x == 0)
        for (int k = threadIdx.x; k < G * G; k += AB_THREADS) counts[k] = 0;
    __syncthreads();
    if (threadIdx.x < 6) {
        float r = s_red[threadIdx.x][0];
        for (int w = 1; w < AB_THREADS / 64; ++w) r = threadIdx.x < 3 ? fminf(r, s_red[threadIdx.x][w]) : fmaxf(r, s_red[threadIdx.x][w]);
        s_b[threadIdx.x] = r;
    }
    __syncthreads();
    const f3 blo = {s_b[0], s_b[1], s_b[2]}, bhi = {s_b[3], s_b[4], s_b[5]};
    if (blockIdx.x == 1) { // the second block orders the vertices while the first orders the triangles (both know the bounds)
        for (int i = threadIdx.x; i < nv2; i += AB_THREADS)
            s_keys[i] = i < nv ? ((unsigned long long)morton30(vert3(V, i), blo, bhi) << 32) | (unsigned)i : ~0ull;
        __syncthreads();
        bitonic_sort_lds(s_keys, nv2);
        for (int i = threadIdx.x; i < nv; i += AB_THREADS) vert_order[i] = (int32_t)(s_keys[i] & 0xffffffffu);
        return;
    }
    if (threadIdx.x == 0) {
        // cell = clamp(floor((c - c0) / size), 0, G - 1): the SAME fp32 expression in cell_range() below and in the query kernel, so the monotone
        // map sends every point of a triangle's (y,z) bounding box into the triangle's cell range
        const float cy = (bhi.y - blo.y) / (float)G, cz = (bhi.z - blo.z) / (float)G;
        grid[0] = blo.y;
        grid[1] = blo.z;
        grid[2] = cy > 0.0f ? cy : 1e-6f;
        grid[3] = cz > 0.0f ? cz : 1e-6f;
        grid[4] = __int_as_float(G);
    }
    for (int i = threadIdx.x; i < nt2; i += AB_THREADS) {
        unsigned long long key = ~0ull;
        if (i < nf) {
            const f3 a = vert3(V, F[3 * i]), b = vert3(V, F[3 * i + 1]), c = vert3(V, F[3 * i + 2]);
            const f3 cen = {((a.x + b.x) + c.x) / 3.0f, ((a.y + b.y) + c.y) / 3.0f, ((a.z + b.z) + c.z) / 3.0f};
            key = ((unsigned long long)morton30(cen, blo, bhi) << 32) | (unsigned)i;
        }
        s_keys[i] = key;
    }
    __syncthreads();
    bitonic_sort_lds(s_keys, nt2);
    for (int i = threadIdx.x; i < nf; i += AB_THREADS) tri_order[i] = (int32_t)(s_keys[i] & 0xffffffffu);
}

__global__ __launch_bounds__(256) void accel_tables_kernel(const float* __restrict__ V, int nv, const int32_t* __restrict__ F, int nf, int nc, int nvc,
                                                           const int32_t* __restrict__ tri_order, const int32_t* __restrict__ vert_order,
                                                           float* __restrict__ tri, float* __restrict__ sphere, float* __restrict__ tnorm,
                                                           int32_t* __restrict__ orig, float* __restrict__ cbox, float* __restrict__ cdisc,
                                                           float* __restrict__ vsort, float* __restrict__ vbox)
{
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t < nc) {
        f3 P[CL * 3], nsum = {0.0f, 0.0f, 0.0f};
        float pad = 0.0f;
        for (int k = 0; k < CL; ++k) {
            const int slot = t * CL + k;
            int f = 0x7FFFFFFF;
            f3 a, b, c;
            if (slot < nf) {
                f = tri_order[slot];
                a = vert3(V, F[3 * f]);
                b = vert3(V, F[3 * f + 1]);
                c = vert3(V, F[3 * f + 2]);
            } else { // far-away (degenerate) triangles that can never be the closest
                a = {1.0e4f, 1.0e4f, 1.0e4f};
                b = {1.0e4f + 1.0f, 1.0e4f + 1.0f, 1.0e4f + 1.0f};
                c = {1.0e4f + 2.0f, 1.0e4f + 2.0f, 1.0e4f + 2.0f};
            }
            P[3 * k] = a, P[3 * k + 1] = b, P[3 * k + 2] = c;
            const f3 cr[3] = {a, b, c};
            float amax = 0.0f;
            for (int j = 0; j < 3; ++j) {
                tri[9 * slot + 3 * j] = cr[j].x, tri[9 * slot + 3 * j + 1] = cr[j].y, tri[9 * slot + 3 * j + 2] = cr[j].z;
                amax = fmaxf(amax, fmaxf(fabsf(cr[j].x), fmaxf(fabsf(cr[j].y), fabsf(cr[j].z))));
            }
            orig[slot] = f;
            const f3 cen = {((a.x + b.x) + c.x) / 3.0f, ((a.y + b.y) + c.y) / 3.0f, ((a.z + b.z) + c.z) / 3.0f};
            const float a_t = amax * 1e-6f + 1e-9f; // absolute slack of the oriented bounds (fp32 error of centres and normals)
            float r2 = 0.0f;
            for (int j = 0; j < 3; ++j) {
                const f3 d = sub3(cr[j], cen);
                r2 = fmaxf(r2, dot3(d, d));
            }
            const float rad = sqrtf(r2) * (1.0f + 1e-5f) + a_t;
            sphere[4 * slot] = cen.x, sphere[4 * slot + 1] = cen.y, sphere[4 * slot + 2] = cen.z, sphere[4 * slot + 3] = rad;
            // oriented bound: the triangle lies in the plane through its centroid, within `rad` of it
            const f3 e1 = sub3(b, a), e2 = sub3(c, a);
            const f3 nrm = {e1.y * e2.z - e1.z * e2.y, e1.z * e2.x - e1.x * e2.z, e1.x * e2.y - e1.y * e2.x};
            const float nlen = sqrtf(dot3(nrm, nrm));
            const float inv = nlen > 1e-20f ? 1.0f / fmaxf(nlen, 1e-30f) : 0.0f;
            tnorm[4 * slot] = nrm.x * inv, tnorm[4 * slot + 1] = nrm.y * inv, tnorm[4 * slot + 2] = nrm.z * inv, tnorm[4 * slot + 3] = 1e5f * a_t * a_t;
            nsum = {nsum.x + nrm.x, nsum.y + nrm.y, nsum.z + nrm.z};
            pad = fmaxf(pad, a_t);
        }
        f3 lo = P[0], hi = P[0], cc = {0.0f, 0.0f, 0.0f};
        for (int j = 0; j < CL * 3; ++j) {
            lo = {fminf(lo.x, P[j].x), fminf(lo.y, P[j].y), fminf(lo.z, P[j].z)};
            hi = {fmaxf(hi.x, P[j].x), fmaxf(hi.y, P[j].y), fmaxf(hi.z, P[j].z)};
            cc = {cc.x + P[j].x, cc.y + P[j].y, cc.z + P[j].z};
        }
        cbox[6 * t] = lo.x, cbox[6 * t + 1] = lo.y, cbox[6 * t + 2] = lo.z, cbox[6 * t + 3] = hi.x, cbox[6 * t + 4] = hi.y, cbox[6 * t + 5] = hi.z;
        // cylinder around the cluster (tile search): centre = mean corner, axis = normalised sum of the triangles' area normals, radius /
        // half height = the largest lateral / axial offset of a corner
        cc = {cc.x / (float)(CL * 3), cc.y / (float)(CL * 3), cc.z / (float)(CL * 3)};
        const float cal = sqrtf(dot3(nsum, nsum));
        const float cinv = cal > 1e-20f ? 1.0f / fmaxf(cal, 1e-30f) : 0.0f;
        const f3 ax = {nsum.x * cinv, nsum.y * cinv, nsum.z * cinv};
        float lat = 0.0f, hh = 0.0f;
        for (int j = 0; j < CL * 3; ++j) {
            const f3 off = sub3(P[j], cc);
            const float h = dot3(off, ax);
            lat = fmaxf(lat, sqrtf(fmaxf(dot3(off, off) - h * h, 0.0f)));
            hh = fmaxf(hh, fabsf(h));
        }
        float* cd = cdisc + 8 * t;
        cd[0] = cc.x, cd[1] = cc.y, cd[2] = cc.z, cd[3] = lat * (1.0f + 1e-5f) + pad;
        cd[4] = ax.x, cd[5] = ax.y, cd[6] = ax.z, cd[7] = hh * (1.0f + 1e-5f) + pad;
    } else if (t - nc < nvc) {
        const int c = t - nc;
        f3 lo = {INFINITY, INFINITY, INFINITY}, hi = {-INFINITY, -INFINITY, -INFINITY};
        for (int k = 0; k < CL; ++k) {
            const int slot = c * CL + k;
            int idx = 0x7FFFFFFF;
            f3 v = {1.0e4f, 1.0e4f, 1.0e4f};
            if (slot < nv) {
                idx = vert_order[slot];
                v = vert3(V, idx);
            }
            vsort[4 * slot] = v.x, vsort[4 * slot + 1] = v.y, vsort[4 * slot + 2] = v.z, vsort[4 * slot + 3] = __int_as_float(idx);
            lo = {fminf(lo.x, v.x), fminf(lo.y, v.y), fminf(lo.z, v.z)};
            hi = {fmaxf(hi.x, v.x), fmaxf(hi.y, v.y), fmaxf(hi.z, v.z)};
        }
        vbox[6 * c] = lo.x, vbox[6 * c + 1] = lo.y, vbox[6 * c + 2] = lo.z, vbox[6 * c + 3] = hi.x, vbox[6 * c + 4] = hi.y, vbox[6 * c + 5] = hi.z;
    }
}

// cells of the (y,z) grid the bounding box of triangle f touches: [ylo, yhi] x [zlo, zhi]
__device__ __forceinline__ void cell_range(const float* V, const int32_t* F, int f, const float* grid, int G, int& ylo, int& yhi, int& zlo, int& zhi)
{
    const f3 a = vert3(V, F[3 * f]), b = vert3(V, F[3 * f + 1]), c = vert3(V, F[3 * f + 2]);
    const float y0 = grid[0], z0 = grid[1], sy = grid[2], sz = grid[3];
    auto cell = [G](float v, float c0, float size) { return min(max((int)floorf((v - c0) / size), 0), G - 1); };
    ylo = cell(fminf(a.y, fminf(b.y, c.y)), y0, sy);
    yhi = cell(fmaxf(a.y, fmaxf(b.y, c.y)), y0, sy);
    zlo = cell(fminf(a.z, fminf(b.z, c.z)), z0, sz);
    zhi = cell(fmaxf(a.z, fmaxf(b.z, c.z)), z0, sz);
}

__global__ __launch_bounds__(256) void accel_cell_count_kernel(const float* __restrict__ V, const int32_t* __restrict__ F, int nf, int G,
                                                               const float* __restrict__ grid, int32_t* __restrict__ counts)
{
    const int f = blockIdx.x * 256 + threadIdx.x;
    if (f >= nf) return;
    int ylo, yhi, zlo, zhi;
    cell_range(V, F, f, grid, G, ylo, yhi, zlo, zhi);
    for (int y = ylo; y <= yhi; ++y)
        for (int z = zlo; z <= zhi; ++z) atomicAdd(&counts[y * G + z], 1);
}

__global__ __launch_bounds__(AB_THREADS) void accel_cell_scan_kernel(int G, int nf, int capacity, float* __restrict__ grid, int32_t* __restrict__ counts,
                                                                     int32_t* __restrict__ cell_start)
{
    __shared__ int s_part[AB_THREADS];
    const int n = G * G, per = (n + AB_THREADS - 1) / AB_THREADS, b0 = threadIdx.x * per, b1 = min(n, b0 + per);
    int sum = 0;
    for (int k = b0; k < b1; ++k) sum += counts[k];
    s_part[threadIdx.x] = sum;
    __syncthreads();
    for (int o = 1; o < AB_THREADS; o <<= 1) { // inclusive scan of the partial sums
        const int v = (int)threadIdx.x >= o ? s_part[threadIdx.x - o] : 0;
        __syncthreads();
        s_part[threadIdx.x] += v;
        __syncthreads();
    }
    const int total = s_part[AB_THREADS - 1];
    if (total > capacity) { // the lists do not fit: one cell with every triangle
        if (threadIdx.x == 0) {
            grid[4] = __int_as_float(1);
            grid[2] *= (float)G; // the one cell spans the whole extent (the query's "beside the mesh" test counts in cells)
            grid[3] *= (float)G;
            cell_start[0] = 0;
            cell_start[1] = nf;
        }
        return;
    }
    int run = s_part[threadIdx.x] - sum;
    for (int k = b0; k < b1; ++k) {
        cell_start[k] = run;
        run += counts[k];
        counts[k] = 0; // the fill's cursors
    }
    if (threadIdx.x == 0) cell_start[n] = total;
}

__global__ __launch_bounds__(256) void accel_cell_fill_kernel(const float* __restrict__ V, const int32_t* __restrict__ F, int nf, int G,
                                                              const float* __restrict__ grid, int32_t* __restrict__ cursors,
                                                              const int32_t* __restrict__ cell_start, int32_t* __restrict__ unordered,
                                                              int32_t* __restrict__ cell_tri)
{
    const int f = blockIdx.x * 256 + threadIdx.x;
    if (f >= nf) return;
    if (__float_as_int(grid[4]) != G) { // degraded to one cell (accel_cell_scan_kernel)
        cell_tri[f] = f;
        return;
    }
    int ylo, yhi, zlo, zhi;
    cell_range(V, F, f, grid, G, ylo, yhi, zlo, zhi);
    for (int y = ylo; y <= yhi; ++y)
        for (int z = zlo; z <= zhi; ++z) {
            const int cell = y * G + z;
            unordered[cell_start[cell] + atomicAdd(&cursors[cell], 1)] = f;
        }
}

__global__ __launch_bounds__(256) void accel_cell_order_kernel(int G, const float* __restrict__ grid, const int32_t* __restrict__ cell_start,
                                                               const int32_t* __restrict__ unordered, int32_t* __restrict__ cell_tri)
{
    // one wave per cell: an id's place in the ascending list is the number of smaller ids (ids are unique within a cell)
    const int cell = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (cell >= G * G || __float_as_int(grid[4]) != G) return;
    const int b = cell_start[cell], e = cell_start[cell + 1];
    for (int i = b + lane; i < e; i += 64) {
        const int v = unordered[i];
        int rank = 0;
        for (int j = b; j < e; ++j) rank += unordered[j] < v;
        cell_tri[b + rank] = v;
    }
}

// the cells' lists as self-contained records (see the inside test of mesh_query_accel_kernel)
__global__ __launch_bounds__(256) void accel_cell_record_kernel(const float* __restrict__ V, const int32_t* __restrict__ F, const float* __restrict__ grid,
                                                                const int32_t* __restrict__ cell_start, const int32_t* __restrict__ cell_tri,
                                                                int capacity, float* __restrict__ cell_rec)
{
    const int e = blockIdx.x * 256 + threadIdx.x, Ge = __float_as_int(grid[4]); // (the grid in force: G, or 1 if it was degraded)
    if (e >= capacity || e >= cell_start[Ge * Ge]) return;
    const int f = cell_tri[e];
    const int i0 = F[3 * f], i1 = F[3 * f + 1], i2 = F[3 * f + 2];
    const f3 a = vert3(V, i0), b = vert3(V, i1), c = vert3(V, i2);
    float4* r = reinterpret_cast<float4*>(cell_rec) + 3 * (size_t)e;
    r[0] = make_float4(__int_as_float(i0), __int_as_float(i1), __int_as_float(i2), a.x);
    r[1] = make_float4(a.y, a.z, b.x, b.y);
    r[2] = make_float4(b.z, c.x, c.y, c.z);
}

struct AccelLayout {
    size_t tri, sphere, tnorm, orig, cbox, cdisc, cell_start, cell_tri, cell_rec, grid, vsort, vbox, tri_order, vert_order, counts, unordered, total;
    int nfp, nc, nvp, nvc;
};

AccelLayout accel_layout(int nv, int nf, int G, int capacity)
{
    AccelLayout L{};
    L.nc = (nf + CL - 1) / CL, L.nfp = L.nc * CL, L.nvc = (nv + CL - 1) / CL, L.nvp = L.nvc * CL;
    size_t at = 0;
    auto take = [&at](size_t bytes) { const size_t o = at; at += (bytes + 255) & ~(size_t)255; return o; };
    L.tri = take(sizeof(float) * 9 * L.nfp);
    L.sphere = take(sizeof(float) * 4 * L.nfp);
    L.tnorm = take(sizeof(float) * 4 * L.nfp);
    L.orig = take(sizeof(int32_t) * L.nfp);
    L.cbox = take(sizeof(float) * 6 * L.nc);
    L.cdisc = take(sizeof(float) * 8 * L.nc);
    L.cell_start = take(sizeof(int32_t) * ((size_t)G * G + 1));
    L.cell_tri = take(sizeof(int32_t) * (size_t)capacity);
    L.cell_rec = take(sizeof(float) * 12 * (size_t)capacity);
    L.grid = take(sizeof(float) * 8);
    L.vsort = take(sizeof(float) * 4 * L.nvp);
    L.vbox = take(sizeof(float) * 6 * L.nvc);
    L.tri_order = take(sizeof(int32_t) * nf);
    L.vert_order = take(sizeof(int32_t) * nv);
    L.counts = take(sizeof(int32_t) * (size_t)G * G);
    L.unordered = take(sizeof(int32_t) * (size_t)capacity);
    L.total = at;
    return L;
}

void accel_check_sizes(int nv, int nf, int G, int capacity)
{
    if (nv <= 0 || nf <= 0) throw_error("vanerf_mesh_accel_build: nv=%d nf=%d", nv, nf);
    if ((nf + CL - 1) / CL > MA_MAX_CLUSTERS || (nv + CL - 1) / CL > MA_MAX_VCLUSTERS)
        throw_error("vanerf_mesh_accel_build: mesh too large (at most %d faces, %d vertices)", MA_MAX_CLUSTERS * CL, MA_MAX_VCLUSTERS * CL);
    if (G < 1 || G > 256) throw_error("vanerf_mesh_accel_build: grid size %d outside [1, 256]", G);
    if (capacity < nf) throw_error("vanerf_mesh_accel_build: cell_capacity %d is below the number of faces %d", capacity, nf);
}

int pow2_at_least(int n)
{
    int p = 1;
    while (p < n) p <<= 1;
    return p;
}

} // namespace

extern "C" int64_t vanerf_mesh_accel_bytes(int nv, int nf, int G, int cell_capacity)
{
    int64_t bytes = -1;
    const int rc = guarded([&] {
        accel_check_sizes(nv, nf, G, cell_capacity);
        bytes = (int64_t)accel_layout(nv, nf, G, cell_capacity).total;
    });
    return rc == VANERF_OK ? bytes : (int64_t)rc;
}

extern "C" int vanerf_mesh_accel_build(const float* verts, int nv, const int32_t* faces, int nf, int G, int cell_capacity, void* tables,
                                       int64_t tables_bytes, VanerfMeshAccel* out, void* stream)
{
    return guarded([&] {
        if (!verts || !faces || !tables || !out) throw_error("vanerf_mesh_accel_build: null argument");
        accel_check_sizes(nv, nf, G, cell_capacity);
        const AccelLayout L = accel_layout(nv, nf, G, cell_capacity);
        if (tables_bytes < (int64_t)L.total) throw_error("vanerf_mesh_accel_build: table block of %lld bytes, %zu needed", (long long)tables_bytes, L.total);
        if (reinterpret_cast<uintptr_t>(tables) % 16 != 0) throw_error("vanerf_mesh_accel_build: table block must be 16-byte aligned");
        char* base = static_cast<char*>(tables);
        auto F32 = [base](size_t o) { return reinterpret_cast<float*>(base + o); };
        auto I32 = [base](size_t o) { return reinterpret_cast<int32_t*>(base + o); };
        const hipStream_t st = (hipStream_t)stream;
        const int nt2 = pow2_at_least(nf), nv2 = pow2_at_least(nv);
        const size_t lds = sizeof(unsigned long long) * (size_t)(nt2 > nv2 ? nt2 : nv2);
        if (lds > 64 * 1024)
            HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(accel_bounds_sort_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(accel_bounds_sort_kernel, dim3(2), dim3(AB_THREADS), lds, st, verts, nv, faces, nf, G, nt2, nv2, F32(L.grid), I32(L.tri_order),
                           I32(L.vert_order), I32(L.counts));
        HIP_CHECK(hipGetLastError());
        hipLaunchKernelGGL(accel_tables_kernel, dim3((unsigned)((L.nc + L.nvc + 255) / 256)), dim3(256), 0, st, verts, nv, faces, nf, L.nc, L.nvc,
                           I32(L.tri_order), I32(L.vert_order), F32(L.tri), F32(L.sphere), F32(L.tnorm), I32(L.orig), F32(L.cbox), F32(L.cdisc),
                           F32(L.vsort), F32(L.vbox));
        HIP_CHECK(hipGetLastError());
        const unsigned fb = (unsigned)((nf + 255) / 256);
        hipLaunchKernelGGL(accel_cell_count_kernel, dim3(fb), dim3(256), 0, st, verts, faces, nf, G, F32(L.grid), I32(L.counts));
        HIP_CHECK(hipGetLastError());
        hipLaunchKernelGGL(accel_cell_scan_kernel, dim3(1), dim3(AB_THREADS), 0, st, G, nf, cell_capacity, F32(L.grid), I32(L.counts), I32(L.cell_start));
        HIP_CHECK(hipGetLastError());
        hipLaunchKernelGGL(accel_cell_fill_kernel, dim3(fb), dim3(256), 0, st, verts, faces, nf, G, F32(L.grid), I32(L.counts), I32(L.cell_start),
                           I32(L.unordered), I32(L.cell_tri));
        HIP_CHECK(hipGetLastError());
        hipLaunchKernelGGL(accel_cell_order_kernel, dim3((unsigned)((G * G + 3) / 4)), dim3(256), 0, st, G, F32(L.grid), I32(L.cell_start),
                           I32(L.unordered), I32(L.cell_tri));
        HIP_CHECK(hipGetLastError());
        hipLaunchKernelGGL(accel_cell_record_kernel, dim3((unsigned)((cell_capacity + 255) / 256)), dim3(256), 0, st, verts, faces, F32(L.grid),
                           I32(L.cell_start), I32(L.cell_tri), cell_capacity, F32(L.cell_rec));
        HIP_CHECK(hipGetLastError());
        VanerfMeshAccel A{};
        A.tri = F32(L.tri), A.sphere = F32(L.sphere), A.tnorm = F32(L.tnorm), A.orig = I32(L.orig), A.cbox = F32(L.cbox), A.cdisc = F32(L.cdisc);
        A.nfp = L.nfp, A.nc = L.nc;
        A.cell_start = I32(L.cell_start), A.cell_tri = I32(L.cell_tri), A.cell_rec = F32(L.cell_rec), A.grid = F32(L.grid);
        A.vsort = F32(L.vsort), A.vbox = F32(L.vbox), A.nvc = L.nvc;
        *out = A;
    });
}
