// render_kernels.hip -- the per-ray ends of the march: pixel grid + ray generation + bbox clipping +
// coarse depths (src/model.py:1191-1238, 1496-1570), sample positions (1234-1235), alpha compositing
// (879-882, 1464-1494) and inverse-CDF importance sampling + sort-merge (1424-1462, 1301-1307).
// All of it is HBM-bound streaming work: one thread per ray, rows of S floats.
// Built with -ffp-contract=off (mul and add stay separate IEEE operations, as in eager PyTorch).
#include <algorithm>

#include "common.h"

using namespace vanerf;

namespace {

struct RayParams {
    int x0, y0, step_x, step_y, y_block, nx, ny, width;
    const int32_t* pixels; // optional explicit (x, y) list of nx*ny pixels (training patches), else the regular grid
    const int32_t* row_blocks; // optional first row of every block of y_block rows (multi-GPU shards dealt by cost), else y0 + block * step_y
    float invK_T[9];
    float RT[12];
    float znear, zfar;
    float bounds[6];
    int S;
    const float* t_lin;
    const float* jitter;
    int64_t* index;
    float* rays_d;
    float* cam_pos;
    float* near;
    float* far;
    uint8_t* hit;
    float* z;
};

__device__ __forceinline__ float norm3(float a, float b, float c) { return sqrtf((a * a + b * b) + c * c); }

// ray_bbox_intersection (src/model.py:1496-1570): bounds +-(0.01), |d| < 1e-5 -> 1e-5, six plane hits, inside test with
// eps 1e-6, valid iff exactly two hits; near/far = min/max of |p - o| / |d|, else 1.0.
__device__ __forceinline__ bool ray_bbox(const float* bounds, float ox, float oy, float oz, float dx, float dy, float dz, float& z1, float& z2)
{
    const float lo[3] = {bounds[0] - 0.01f, bounds[1] - 0.01f, bounds[2] - 0.01f};
    const float hi[3] = {bounds[3] + 0.01f, bounds[4] + 0.01f, bounds[5] + 0.01f};
    float d[3] = {dx, dy, dz};
    const float o[3] = {ox, oy, oz};
#pragma unroll
    for (int a = 0; a < 3; ++a)
        if (fabsf(d[a]) < 1e-5f) d[a] = 1e-5f;
    const float eps = 1e-6f;
    int cnt = 0;
    float dist[2] = {0.0f, 0.0f};
    const float nd = norm3(d[0], d[1], d[2]);
#pragma unroll
    for (int k = 0; k < 6; ++k) { // order: min_x, min_y, min_z, max_x, max_y, max_z
        const int a = k % 3;
        const float plane = k < 3 ? lo[a] : hi[a];
        const float t = (plane - o[a]) / d[a];
        const float p0 = t * d[0] + o[0], p1 = t * d[1] + o[1], p2 = t * d[2] + o[2];
        const bool in = (p0 >= lo[0] - eps) && (p0 <= hi[0] + eps) && (p1 >= lo[1] - eps) && (p1 <= hi[1] + eps) &&
                        (p2 >= lo[2] - eps) && (p2 <= hi[2] + eps);
        if (in) {
            if (cnt < 2) dist[cnt] = norm3(p0 - o[0], p1 - o[1], p2 - o[2]) / nd;
            ++cnt;
        }
    }
    const bool hit = cnt == 2;
    z1 = hit ? fminf(dist[0], dist[1]) : 1.0f;
    z2 = hit ? fmaxf(dist[0], dist[1]) : 1.0f;
    return hit;
}

struct BboxParams { float bounds[6]; float o[3]; };

__global__ void ray_bbox_kernel(const BboxParams B, const float* __restrict__ dirs, int R, float* __restrict__ near, float* __restrict__ far,
                                uint8_t* __restrict__ hit)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    float z1, z2;
    const bool h = ray_bbox(B.bounds, B.o[0], B.o[1], B.o[2], dirs[3 * r], dirs[3 * r + 1], dirs[3 * r + 2], z1, z2);
    near[r] = z1; far[r] = z2; hit[r] = h;
}

__global__ __launch_bounds__(256) void ray_setup_kernel(const RayParams P)
{
    __shared__ float s_near[256], s_far[256];
    const int R = P.nx * P.ny;
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    // camera centre: -(t . R)  (src/model.py:1213)
    const float* M = P.RT;
    const float tx = M[3], ty = M[7], tz = M[11];
    const float ox = -((tx * M[0] + ty * M[4]) + tz * M[8]);
    const float oy = -((tx * M[1] + ty * M[5]) + tz * M[9]);
    const float oz = -((tx * M[2] + ty * M[6]) + tz * M[10]);
    if (r == 0) { P.cam_pos[0] = ox; P.cam_pos[1] = oy; P.cam_pos[2] = oz; }
    float near = 0.0f, far = 0.0f;
    if (r < R) {
    const int ix = r % P.nx, iy = r / P.nx;
    const int gxi = P.pixels ? P.pixels[2 * r] : P.x0 + ix * P.step_x;
    const int gyi = P.pixels ? P.pixels[2 * r + 1]
                             : (P.row_blocks ? P.row_blocks[iy / P.y_block] : P.y0 + (iy / P.y_block) * P.step_y) + (iy % P.y_block) * P.step_x;
    P.index[r] = (int64_t)gxi + (int64_t)gyi * P.width;
    const float gx = (float)gxi, gy = (float)gyi;
    const float* K = P.invK_T; // row-major 3x3: c_j = gx*K[0][j] + gy*K[1][j] + K[2][j]
    float c0 = (gx * K[0] + gy * K[3]) + K[6];
    float c1 = (gx * K[1] + gy * K[4]) + K[7];
    float c2 = (gx * K[2] + gy * K[5]) + K[8];
    // znear/zfar along the ray: || (z * [x, y, 1]) K^-T ||  (src/model.py:1204-1211)
    float nx_ = P.znear * gx, ny_ = P.znear * gy, nz_ = P.znear;
    float zn = norm3((nx_ * K[0] + ny_ * K[3]) + nz_ * K[6], (nx_ * K[1] + ny_ * K[4]) + nz_ * K[7], (nx_ * K[2] + ny_ * K[5]) + nz_ * K[8]);
    float fx_ = P.zfar * gx, fy_ = P.zfar * gy, fz_ = P.zfar;
    float zf = norm3((fx_ * K[0] + fy_ * K[3]) + fz_ * K[6], (fx_ * K[1] + fy_ * K[4]) + fz_ * K[7], (fx_ * K[2] + fy_ * K[5]) + fz_ * K[8]);
    // world direction: normalize(c . R)
    float dx = (c0 * M[0] + c1 * M[4]) + c2 * M[8];
    float dy = (c0 * M[1] + c1 * M[5]) + c2 * M[9];
    float dz = (c0 * M[2] + c1 * M[6]) + c2 * M[10];
    float nrm = fmaxf(norm3(dx, dy, dz), 1e-12f);
    dx /= nrm; dy /= nrm; dz /= nrm;
    P.rays_d[3 * r] = dx; P.rays_d[3 * r + 1] = dy; P.rays_d[3 * r + 2] = dz;

    float z1, z2;
    const bool hit = ray_bbox(P.bounds, ox, oy, oz, dx, dy, dz, z1, z2);
    near = (hit && z1 > zn) ? z1 : zn; // src/model.py:1217-1220
    far = (hit && z2 < zf) ? z2 : zf;
    P.near[r] = near; P.far[r] = far; P.hit[r] = hit;
    }
    // coarse depths (src/model.py:1222-1232), written by the whole block in memory order (a thread per ray wrote 64 cache lines per store)
    s_near[threadIdx.x] = near; s_far[threadIdx.x] = far;
    __syncthreads();
    const int S = P.S;
    const int r0 = blockIdx.x * blockDim.x, nr = min((int)blockDim.x, R - r0);
    for (int k = threadIdx.x; k < nr * S; k += blockDim.x) {
        const int rl = k / S, i = k - rl * S;
        float t = P.t_lin[i];
        if (P.jitter) {
            float lo_t = i == 0 ? P.t_lin[0] : 0.5f * (P.t_lin[i] + P.t_lin[i - 1]);
            float hi_t = i == S - 1 ? P.t_lin[S - 1] : 0.5f * (P.t_lin[i + 1] + P.t_lin[i]);
            t = lo_t + P.jitter[(size_t)(r0 + rl) * S + i] * (hi_t - lo_t);
        }
        P.z[(size_t)(r0 + rl) * S + i] = s_near[rl] + (s_far[rl] - s_near[rl]) * t;
    }
}

__global__ void sample_points_kernel(const float* __restrict__ rays_d, const float* __restrict__ cam_pos,
                                     const float* __restrict__ z, long long n, int S, float* __restrict__ pts)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const long long r = i / S;
    const float t = z[i];
    pts[3 * i + 0] = cam_pos[0] + rays_d[3 * r + 0] * t;
    pts[3 * i + 1] = cam_pos[1] + rays_d[3 * r + 1] * t;
    pts[3 * i + 2] = cam_pos[2] + rays_d[3 * r + 2] * t;
}

// sdf_activation + rgba2out.  Sums are accumulated in fp64 from fp32 products so the result does not
// depend on the summation order (the reference's float .sum() order is an implementation detail).
// With `src` the S samples of a ray are gathered from two per-ray tables (the coarse samples and the newly drawn
// importance samples) in merged depth order: src >= 0 -> table A entry src, src < 0 -> table B entry ~src.  The
// per-sample networks are pure functions of the sample position, so re-using the coarse evaluations in the fine
// composite gives the same bits as evaluating them again (the reference re-evaluates them, src/model.py:1305-1345).
__global__ void composite_kernel(const float* __restrict__ rgba, const float* __restrict__ z, const float* __restrict__ msdf,
                                 const float* __restrict__ rgba_b, const float* __restrict__ msdf_b, const int32_t* __restrict__ src,
                                 int Sa, int Sb, int R, int S, float beta, const float* __restrict__ beta_dev, float* __restrict__ color, float* __restrict__ depth,
                                 float* __restrict__ alpha, float* __restrict__ sdf, float* __restrict__ contrib)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    if (beta_dev) beta = *beta_dev; // a handle re-packed on the device (vanerf_weights_update) carries sigmoid_beta there, already clamped
    const float* qa = rgba + (size_t)r * Sa * 5;
    const float* ma = msdf + (size_t)r * Sa;
    const float* qb = rgba_b ? rgba_b + (size_t)r * Sb * 5 : nullptr;
    const float* mb = msdf_b ? msdf_b + (size_t)r * Sb : nullptr;
    const int32_t* sr = src ? src + (size_t)r * S : nullptr;
    const float* zr = z + (size_t)r * S;
    double cr = 0, cg = 0, cb = 0, ca = 0, cs = 0, cd = 0;
    float T = 1.0f;
    float zi = zr[0];
    for (int i = 0; i < S; ++i) {
        const float zn = i + 1 < S ? zr[i + 1] : 0.0f;
        const float dist = i + 1 < S ? zn - zi : 1e10f;
        const float* q;
        float m;
        if (sr) {
            const int k = sr[i];
            if (k >= 0) { q = qa + 5 * k; m = ma[k]; } else { q = qb + 5 * (~k); m = mb[~k]; }
        } else {
            q = qa + 5 * i; m = ma[i];
        }
        const float a = q[0] + m;
        const float sg = (1.0f / (1.0f + expf(-(-a / beta)))) / beta; // sigmoid(-a / beta) / beta
        const float c = 1.0f - expf(-sg * dist);
        const float w = c * T;
        T = T * (1.0f - c);
        if (contrib) contrib[(size_t)r * S + i] = w;
        cr += (double)(q[2] * w);
        cg += (double)(q[3] * w);
        cb += (double)(q[4] * w);
        ca += (double)w;
        cs += (double)(q[1] * w);
        cd += (double)(zi * w);
        zi = zn;
    }
    const float acc = (float)ca;
    color[3 * r] = (float)cr; color[3 * r + 1] = (float)cg; color[3 * r + 2] = (float)cb;
    alpha[r] = acc;
    sdf[r] = (float)cs / (acc + 1e-8f);
    depth[r] = (float)cd / (acc + 1e-8f);
}

// importance_sample + sort-merge, one thread per ray; per-thread arrays live in LDS with a 64-thread stride
// (element i of thread t at [i*64 + t]: conflict-free).
constexpr int IM_BLOCK = 64;

// MID = true: the reference's own call shape importance_sample(contrib[1:-1], z_mid, ...) (src/model.py:1304): `contrib` holds the
// Sc-2 inner contributions and `z` the Sc-1 mid-points; only z_new / idx are produced.
template <bool MID>
__global__ __launch_bounds__(IM_BLOCK) void importance_merge_kernel(const float* __restrict__ contrib, const float* __restrict__ z,
                                                                    const float* __restrict__ u, const float* __restrict__ t_lin, int R,
                                                                    int Sc, int Sf, float* __restrict__ z_new, float* __restrict__ z_fine,
                                                                    int32_t* __restrict__ src, int32_t* __restrict__ idx_out)
{
    extern __shared__ float lds[];
    const int t = threadIdx.x;
    const int r = blockIdx.x * IM_BLOCK + t;
    const int nb = Sc - 2; // bins
    float* cdf = lds + t;                          // nb + 1 entries
    float* zmid = cdf + (size_t)(nb + 1) * IM_BLOCK; // nb + 1 entries
    float* smp = zmid + (size_t)(nb + 1) * IM_BLOCK; // Sf entries
    if (r >= R) return;
    const float* c = MID ? contrib + (size_t)r * nb - 1 : contrib + (size_t)r * Sc; // c[i + 1] = i-th inner contribution
    const float* zr = MID ? z + (size_t)r * (nb + 1) : z + (size_t)r * Sc;
    // pdf = (contrib[1:-1] + 1e-5) / sum ; cdf = [0, cumsum(pdf)] (fp64 accumulate, fp32 values)
    double tot = 0.0;
    for (int i = 0; i < nb; ++i) tot += (double)(c[i + 1] + 1e-5f);
    const float sum = (float)tot;
    double run = 0.0;
    cdf[0] = 0.0f;
    for (int i = 0; i < nb; ++i) {
        run += (double)((c[i + 1] + 1e-5f) / sum);
        cdf[(size_t)(i + 1) * IM_BLOCK] = (float)run;
    }
    for (int i = 0; i < nb + 1; ++i) zmid[(size_t)i * IM_BLOCK] = MID ? zr[i] : 0.5f * (zr[i + 1] + zr[i]);
    bool sorted = true;
    float prev = -INFINITY;
    for (int k = 0; k < Sf; ++k) {
        const float uk = u ? u[(size_t)r * Sf + k] : t_lin[k];
        // searchsorted(cdf, uk, right=True): first position with cdf > uk
        int lo = 0, hi = nb + 1;
        while (lo < hi) {
            int mid = (lo + hi) >> 1;
            if (cdf[(size_t)mid * IM_BLOCK] <= uk) lo = mid + 1; else hi = mid;
        }
        const int ip = max(lo - 1, 0), in = min(lo, nb);
        const float cp = cdf[(size_t)ip * IM_BLOCK], cn = cdf[(size_t)in * IM_BLOCK];
        const float zp = zmid[(size_t)ip * IM_BLOCK], zq = zmid[(size_t)in * IM_BLOCK];
        float den = cn - cp;
        if (den < 1e-5f) den = 1.0f;
        const float s = zp + ((uk - cp) / den) * (zq - zp);
        smp[(size_t)k * IM_BLOCK] = s;
        z_new[(size_t)r * Sf + k] = s;
        if (idx_out) idx_out[(size_t)r * Sf + k] = in;
        sorted = sorted && (s >= prev);
        prev = s;
    }
    if (MID) return;
    // sort-merge (th.sort(th.cat([z, z_fine]))): stable merge of the two runs; the new samples are sorted first if
    // needed (random u in training, or a last-bit inversion), remembering where each merged sample came from.
    int32_t* srow = src + (size_t)r * (Sc + Sf);
    float* frow = z_fine + (size_t)r * (Sc + Sf);
    int* perm = reinterpret_cast<int*>(smp + (size_t)Sf * IM_BLOCK); // Sf entries
    for (int k = 0; k < Sf; ++k) perm[(size_t)k * IM_BLOCK] = k;
    if (!sorted) { // insertion sort (values + origin)
        for (int k = 1; k < Sf; ++k) {
            const float v = smp[(size_t)k * IM_BLOCK];
            const int pv = perm[(size_t)k * IM_BLOCK];
            int m = k - 1;
            while (m >= 0 && smp[(size_t)m * IM_BLOCK] > v) {
                smp[(size_t)(m + 1) * IM_BLOCK] = smp[(size_t)m * IM_BLOCK];
                perm[(size_t)(m + 1) * IM_BLOCK] = perm[(size_t)m * IM_BLOCK];
                --m;
            }
            smp[(size_t)(m + 1) * IM_BLOCK] = v;
            perm[(size_t)(m + 1) * IM_BLOCK] = pv;
        }
    }
    int a = 0, b = 0;
    bool asc = true;
    float last = -INFINITY;
    for (int k = 0; k < Sc + Sf; ++k) {
        const float za = a < Sc ? zr[a] : INFINITY;
        const float zb = b < Sf ? smp[(size_t)b * IM_BLOCK] : INFINITY;
        const bool take_a = (a < Sc) && (b >= Sf || za <= zb);
        float v;
        if (take_a) { v = za; frow[k] = za; srow[k] = a; ++a; }
        else { v = zb; frow[k] = zb; srow[k] = ~perm[(size_t)b * IM_BLOCK]; ++b; }
        asc = asc && (v >= last);
        last = v;
    }
    if (!asc) { // the coarse depths were not ascending (near > far: a far plane in front of the bounding box): the reference sorts the
                // concatenation (src/model.py:1303) -- stable insertion sort of the merged row, values and origins together
        for (int k = 1; k < Sc + Sf; ++k) {
            const float v = frow[k];
            const int32_t o = srow[k];
            int m = k - 1;
            while (m >= 0 && frow[m] > v) { frow[m + 1] = frow[m]; srow[m + 1] = srow[m]; --m; }
            frow[m + 1] = v; srow[m + 1] = o;
        }
    }
}

} // namespace

static void ray_setup_impl(const int32_t* pixels, const int32_t* row_blocks, int x0, int y0, int step_x, int step_y, int y_block, int nx, int ny, int width, const float* invK_T,
                           const float* RT, float znear, float zfar, const float* bounds, int S, const float* t_lin, const float* jitter,
                           int64_t* index, float* rays_d, float* cam_pos, float* near, float* far, uint8_t* hit, float* z, void* stream);

extern "C" int vanerf_ray_setup(int x0, int y0, int step_x, int step_y, int y_block, int nx, int ny, int width, const float* invK_T, const float* RT,
                                float znear, float zfar, const float* bounds, int S, const float* t_lin, const float* jitter,
                                int64_t* index, float* rays_d, float* cam_pos, float* near, float* far, uint8_t* hit, float* z,
                                void* stream)
{
    return guarded([&] {
        ray_setup_impl(nullptr, nullptr, x0, y0, step_x, step_y, y_block, nx, ny, width, invK_T, RT, znear, zfar, bounds, S, t_lin, jitter, index, rays_d,
                       cam_pos, near, far, hit, z, stream);
    });
}

// the same grid with the rows given as a list of blocks: row iy is row_blocks[iy / y_block] + (iy % y_block) * step_x (device table of ny / y_block
// entries) -- a multi-GPU shard whose 8-row blocks were dealt by cost (vanerf_amd/parallel.py: deal_blocks) instead of round robin
extern "C" int vanerf_ray_setup_blocks(const int32_t* row_blocks, int x0, int step_x, int y_block, int nx, int ny, int width, const float* invK_T,
                                       const float* RT, float znear, float zfar, const float* bounds, int S, const float* t_lin, const float* jitter,
                                       int64_t* index, float* rays_d, float* cam_pos, float* near, float* far, uint8_t* hit, float* z, void* stream)
{
    return guarded([&] {
        if (!row_blocks) throw_error("vanerf_ray_setup_blocks: null block list");
        if (y_block <= 0 || ny % y_block != 0) throw_error("vanerf_ray_setup_blocks: ny = %d is not a whole number of blocks of %d rows", ny, y_block);
        ray_setup_impl(nullptr, row_blocks, x0, 0, step_x, 1, y_block, nx, ny, width, invK_T, RT, znear, zfar, bounds, S, t_lin, jitter, index, rays_d,
                       cam_pos, near, far, hit, z, stream);
    });
}

extern "C" int vanerf_ray_setup_pixels(const int32_t* pixels_xy, int n_rays, int width, const float* invK_T, const float* RT, float znear,
                                       float zfar, const float* bounds, int S, const float* t_lin, const float* jitter, int64_t* index,
                                       float* rays_d, float* cam_pos, float* near, float* far, uint8_t* hit, float* z, void* stream)
{
    return guarded([&] {
        if (!pixels_xy) throw_error("vanerf_ray_setup_pixels: null pixel list");
        ray_setup_impl(pixels_xy, nullptr, 0, 0, 1, 1, 1, n_rays, 1, width, invK_T, RT, znear, zfar, bounds, S, t_lin, jitter, index, rays_d, cam_pos,
                       near, far, hit, z, stream);
    });
}

static void ray_setup_impl(const int32_t* pixels, const int32_t* row_blocks, int x0, int y0, int step_x, int step_y, int y_block, int nx, int ny, int width, const float* invK_T,
                           const float* RT, float znear, float zfar, const float* bounds, int S, const float* t_lin, const float* jitter,
                           int64_t* index, float* rays_d, float* cam_pos, float* near, float* far, uint8_t* hit, float* z, void* stream)
{
    {
        if (!invK_T || !RT || !bounds || !t_lin || !index || !rays_d || !cam_pos || !near || !far || !hit || !z)
            throw_error("vanerf_ray_setup: null argument");
        if (nx <= 0 || ny <= 0 || step_x <= 0 || step_y <= 0 || y_block <= 0 || S < 2 || width <= 0)
            throw_error("vanerf_ray_setup: bad grid (nx=%d ny=%d step=%d,%d S=%d)", nx, ny, step_x, step_y, S);
        RayParams P;
        P.x0 = x0; P.y0 = y0; P.step_x = step_x; P.step_y = step_y; P.y_block = y_block; P.nx = nx; P.ny = ny; P.width = width; P.pixels = pixels; P.row_blocks = row_blocks;
        std::copy_n(invK_T, 9, P.invK_T);
        std::copy_n(RT, 12, P.RT);
        std::copy_n(bounds, 6, P.bounds);
        P.znear = znear; P.zfar = zfar; P.S = S; P.t_lin = t_lin; P.jitter = jitter;
        P.index = index; P.rays_d = rays_d; P.cam_pos = cam_pos; P.near = near; P.far = far; P.hit = hit; P.z = z;
        const int R = nx * ny;
        hipLaunchKernelGGL(ray_setup_kernel, dim3((R + 255) / 256), dim3(256), 0, (hipStream_t)stream, P);
        HIP_CHECK(hipGetLastError());
    }
}

extern "C" int vanerf_sample_points(const float* rays_d, const float* cam_pos, const float* z, int R, int S, float* pts, void* stream)
{
    return guarded([&] {
        if (!rays_d || !cam_pos || !z || !pts) throw_error("vanerf_sample_points: null argument");
        if (R <= 0 || S <= 0) throw_error("vanerf_sample_points: R=%d S=%d", R, S);
        const long long n = (long long)R * S;
        hipLaunchKernelGGL(sample_points_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, rays_d, cam_pos, z, n, S, pts);
        HIP_CHECK(hipGetLastError());
    });
}

// The same composite with one WAVE per ray (lane l holds samples l*SPL .. l*SPL+SPL-1): a ray's rows are read with coalesced loads
// (one thread per ray read 64 different cache lines per load instruction: 9.8 GB of line traffic for 0.6 GB of data, 1.33 ms).
// The transmittance T_i = prod_{j<i} (1 - c_j) becomes a per-lane product followed by a wave scan, so its rounding differs from the
// strictly sequential product in the last bits (~1e-7 relative; the reference's th.cumprod order is an implementation detail too).
// Deterministic; identical arithmetic for vanerf_composite and vanerf_composite_merged on the same samples.
template <int SPL>
__global__ __launch_bounds__(256) void composite_wave_kernel(const float* __restrict__ rgba, const float* __restrict__ z, const float* __restrict__ msdf,
                                                             const float* __restrict__ rgba_b, const float* __restrict__ msdf_b,
                                                             const int32_t* __restrict__ src, int Sa, int Sb, int R, int S, float beta, const float* __restrict__ beta_dev,
                                                             float* __restrict__ color, float* __restrict__ depth, float* __restrict__ alpha,
                                                             float* __restrict__ sdf, float* __restrict__ contrib)
{
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= R) return; // whole wave
    if (beta_dev) beta = *beta_dev; // a handle re-packed on the device (vanerf_weights_update) carries sigmoid_beta there, already clamped
    const float* qa = rgba + (size_t)r * Sa * 5;
    const float* ma = msdf + (size_t)r * Sa;
    const float* qb = rgba_b ? rgba_b + (size_t)r * Sb * 5 : nullptr;
    const float* mb = msdf_b ? msdf_b + (size_t)r * Sb : nullptr;
    const int32_t* sr = src ? src + (size_t)r * S : nullptr;
    const float* zr = z + (size_t)r * S;
    float c[SPL], zi[SPL], q1[SPL], q2[SPL], q3[SPL], q4[SPL];
    float lane_keep = 1.0f; // prod over this lane's samples of (1 - c)
#pragma unroll
    for (int k = 0; k < SPL; ++k) {
        const int i = lane * SPL + k;
        c[k] = 0.0f; zi[k] = 0.0f; q1[k] = q2[k] = q3[k] = q4[k] = 0.0f;
        if (i < S) {
            zi[k] = zr[i];
            const float dist = i + 1 < S ? zr[i + 1] - zi[k] : 1e10f;
            const float* q;
            float m;
            if (sr) {
                const int kk = sr[i];
                if (kk >= 0) { q = qa + 5 * kk; m = ma[kk]; } else { q = qb + 5 * (~kk); m = mb[~kk]; }
            } else {
                q = qa + 5 * i; m = ma[i];
            }
            const float a = q[0] + m;
            const float sg = (1.0f / (1.0f + expf(-(-a / beta)))) / beta; // sigmoid(-a / beta) / beta
            c[k] = 1.0f - expf(-sg * dist);
            q1[k] = q[1]; q2[k] = q[2]; q3[k] = q[3]; q4[k] = q[4];
        }
        lane_keep *= 1.0f - c[k];
    }
    // exclusive multiplicative scan of lane_keep over the lanes
    float incl = lane_keep;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const float v = __shfl_up(incl, d);
        if (lane >= d) incl *= v;
    }
    float T = __shfl_up(incl, 1);
    if (lane == 0) T = 1.0f;
    double cr = 0, cg = 0, cb = 0, ca = 0, cs = 0, cd = 0;
#pragma unroll
    for (int k = 0; k < SPL; ++k) {
        const int i = lane * SPL + k;
        const float w = c[k] * T;
        T = T * (1.0f - c[k]);
        if (i < S) {
            if (contrib) contrib[(size_t)r * S + i] = w;
            cr += (double)(q2[k] * w); cg += (double)(q3[k] * w); cb += (double)(q4[k] * w);
            ca += (double)w; cs += (double)(q1[k] * w); cd += (double)(zi[k] * w);
        }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        cr += __shfl_xor(cr, d); cg += __shfl_xor(cg, d); cb += __shfl_xor(cb, d);
        ca += __shfl_xor(ca, d); cs += __shfl_xor(cs, d); cd += __shfl_xor(cd, d);
    }
    if (lane == 0) {
        const float acc = (float)ca;
        color[3 * r] = (float)cr; color[3 * r + 1] = (float)cg; color[3 * r + 2] = (float)cb;
        alpha[r] = acc;
        sdf[r] = (float)cs / (acc + 1e-8f);
        depth[r] = (float)cd / (acc + 1e-8f);
    }
}

static void launch_composite(const float* rgba, const float* z, const float* msdf, const float* rgba_b, const float* msdf_b,
                             const int32_t* src, int Sa, int Sb, int R, int S, float beta, const float* beta_dev, float* color, float* depth,
                             float* alpha, float* sdf, float* contrib, void* stream)
{
    if (R <= 0 || S <= 0) throw_error("vanerf_composite: R=%d S=%d", R, S);
    if (!beta_dev && !(beta > 0.0f)) throw_error("vanerf_composite: beta must be positive");
    if (beta < 2e-3f) beta = 2e-3f; // sdf_activation clamp (src/model.py:880)
    const dim3 wg((R + 3) / 4), wb(256);
#define VANERF_COMPOSITE_WAVE(SPL)                                                                                                  \
    hipLaunchKernelGGL(composite_wave_kernel<SPL>, wg, wb, 0, (hipStream_t)stream, rgba, z, msdf, rgba_b, msdf_b, src, Sa, Sb, R, S, beta, \
                       beta_dev, color, depth, alpha, sdf, contrib)
    if (S <= 64) VANERF_COMPOSITE_WAVE(1);
    else if (S <= 128) VANERF_COMPOSITE_WAVE(2);
    else if (S <= 192) VANERF_COMPOSITE_WAVE(3);
    else if (S <= 256) VANERF_COMPOSITE_WAVE(4);
    else // more than 256 samples per ray: one thread per ray
        hipLaunchKernelGGL(composite_kernel, dim3((R + 63) / 64), dim3(64), 0, (hipStream_t)stream, rgba, z, msdf, rgba_b, msdf_b, src, Sa, Sb,
                           R, S, beta, beta_dev, color, depth, alpha, sdf, contrib);
#undef VANERF_COMPOSITE_WAVE
    HIP_CHECK(hipGetLastError());
}

// eval_func (src/model.py:1140-1160) on raw network outputs [sdf_pred, rad, r, g, b] -> [alpha, sdf, r, g, b]: alpha = mask relu(rad + noise),
// sdf = mask sdf_pred + (1 - mask) invalid_sdf -- the arithmetic of query_kernel's own epilogue, so the same bits.  With per-sample noise (training) the
// networks are evaluated once per point (raw) and this runs once per set of draws: src == NULL: entry i of table a with noise[i]; otherwise position p of
// a ray's merged order carries noise[r][p] and names its entry (src >= 0: table a, < 0: entry ~src of table b); every entry is written once
// (rgba_a / rgba_b may be the raw tables themselves).
__global__ __launch_bounds__(256) void eval_func_kernel(const float* __restrict__ raw_a, const uint8_t* __restrict__ valid_a, const float* __restrict__ raw_b,
                                                        const uint8_t* __restrict__ valid_b, const int32_t* __restrict__ src, const float* __restrict__ noise,
                                                        int Sa, int Sb, long long n, float invalid_sdf, float* __restrict__ rgba_a, float* __restrict__ rgba_b)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int S = Sa + Sb;
    const long long r = i / S;
    const float* q;
    float* o;
    float mask;
    if (src) {
        const int k = src[i];
        if (k >= 0) { const long long e = r * Sa + k; q = raw_a + 5 * e; o = rgba_a + 5 * e; mask = valid_a[e] ? 1.0f : 0.0f; }
        else { const long long e = r * Sb + (~k); q = raw_b + 5 * e; o = rgba_b + 5 * e; mask = valid_b[e] ? 1.0f : 0.0f; }
    } else {
        q = raw_a + 5 * i; o = rgba_a + 5 * i; mask = valid_a[i] ? 1.0f : 0.0f;
    }
    const float q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3], q4 = q[4];
    float rad = q1;
    if (noise) rad += noise[i];
    o[0] = mask * fmaxf(rad, 0.0f);
    o[1] = mask * q0 + (1.0f - mask) * invalid_sdf;
    o[2] = q2; o[3] = q3; o[4] = q4;
}

extern "C" int vanerf_eval_func(const float* raw_a, const uint8_t* valid_a, const float* raw_b, const uint8_t* valid_b, const int32_t* src,
                                const float* noise, int Sa, int Sb, int R, float invalid_sdf, float* rgba_a, float* rgba_b, void* stream)
{
    return guarded([&] {
        if (!raw_a || !valid_a || !rgba_a) throw_error("vanerf_eval_func: null argument");
        if (src && (!raw_b || !valid_b || !rgba_b || Sb <= 0)) throw_error("vanerf_eval_func: the merged order needs the second table");
        if (!src && Sb != 0) throw_error("vanerf_eval_func: Sb = %d without an origin map", Sb);
        if (R <= 0 || Sa <= 0) throw_error("vanerf_eval_func: R=%d Sa=%d", R, Sa);
        const long long n = (long long)R * (Sa + Sb);
        hipLaunchKernelGGL(eval_func_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, raw_a, valid_a, raw_b, valid_b, src, noise,
                           Sa, Sb, n, invalid_sdf, rgba_a, rgba_b);
        HIP_CHECK(hipGetLastError());
    });
}

// Backward of the composite (training; the reference differentiates rgba2out with autograd: src/model.py:1464-1494, sdf_activation 879-882).  One wave per
// ray, the forward quantities recomputed as composite_wave_kernel computes them, then with gw_i = dL/dw_i
//   dL/dsigma_i = dist_i (gw_i T_{i+1} - sum_{k>i} gw_k w_k)        (T_{i+1} = T_i (1 - c_i); no division by (1 - c_i): exact zeros stay zeros, which is what
//                                                                   torch.cumprod's backward branches -- and blocks the host -- for)
//   dL/dx_i = -dL/dsigma_i s(1 - s) / beta^2,  x = rgba0 + mesh_sdf, s = sigmoid(-x / beta);   dL/dbeta = sum_i dL/dsigma_i (s(1 - s) x / beta^3 - s / beta^2)
//   dL/drgb_i = g_color w_i,   dL/drgba1_i = g_sdf w_i / (acc + 1e-8)
// with gw_i = g_color . rgb_i + g_alpha + g_depth (z_i - depth) / (acc + 1e-8) + g_sdf (rgba1_i - sdf) / (acc + 1e-8).  The gradients go back to the table a
// sample came from (src as in the forward kernel: every table entry appears once in the merged order, so plain stores).  d_beta[r]: the ray's share.
template <int SPL>
__global__ __launch_bounds__(256) void composite_backward_kernel(const float* __restrict__ rgba, const float* __restrict__ z, const float* __restrict__ msdf,
                                                                 const float* __restrict__ rgba_b, const float* __restrict__ msdf_b,
                                                                 const int32_t* __restrict__ src, int Sa, int Sb, int R, int S, const float* __restrict__ beta_dev,
                                                                 const float* __restrict__ g_color, const float* __restrict__ g_depth,
                                                                 const float* __restrict__ g_alpha, const float* __restrict__ g_sdf,
                                                                 float* __restrict__ d_a, float* __restrict__ d_b, float* __restrict__ d_beta)
{
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= R) return; // whole wave
    const float beta = *beta_dev;
    const float* qa = rgba + (size_t)r * Sa * 5;
    const float* ma = msdf + (size_t)r * Sa;
    const float* qb = rgba_b ? rgba_b + (size_t)r * Sb * 5 : nullptr;
    const float* mb = msdf_b ? msdf_b + (size_t)r * Sb : nullptr;
    const int32_t* sr = src ? src + (size_t)r * S : nullptr;
    const float* zr = z + (size_t)r * S;
    float c[SPL], zi[SPL], dist[SPL], x[SPL], sg[SPL], q1[SPL], q2[SPL], q3[SPL], q4[SPL];
    int from[SPL]; // where the sample's gradient goes: >= 0 entry of table a, < 0 entry ~from of table b
    float lane_keep = 1.0f;
#pragma unroll
    for (int k = 0; k < SPL; ++k) {
        const int i = lane * SPL + k;
        c[k] = 0.0f; zi[k] = 0.0f; dist[k] = 0.0f; x[k] = 0.0f; sg[k] = 0.0f; q1[k] = q2[k] = q3[k] = q4[k] = 0.0f; from[k] = 0;
        if (i < S) {
            zi[k] = zr[i];
            dist[k] = i + 1 < S ? zr[i + 1] - zi[k] : 1e10f;
            const float* q;
            float m;
            if (sr) {
                const int kk = sr[i];
                from[k] = kk;
                if (kk >= 0) { q = qa + 5 * kk; m = ma[kk]; } else { q = qb + 5 * (~kk); m = mb[~kk]; }
            } else {
                from[k] = i;
                q = qa + 5 * i; m = ma[i];
            }
            x[k] = q[0] + m;
            sg[k] = 1.0f / (1.0f + expf(-(-x[k] / beta))); // sigmoid(-x / beta)
            c[k] = 1.0f - expf(-(sg[k] / beta) * dist[k]);
            q1[k] = q[1]; q2[k] = q[2]; q3[k] = q[3]; q4[k] = q[4];
        }
        lane_keep *= 1.0f - c[k];
    }
    float incl = lane_keep;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const float v = __shfl_up(incl, d);
        if (lane >= d) incl *= v;
    }
    float T0 = __shfl_up(incl, 1);
    if (lane == 0) T0 = 1.0f;
    float w[SPL], Tn[SPL]; // w_i and T_{i+1}
    float ca = 0.0f, cs = 0.0f, cd = 0.0f;
    {
        float T = T0;
#pragma unroll
        for (int k = 0; k < SPL; ++k) {
            w[k] = c[k] * T;
            T = T * (1.0f - c[k]);
            Tn[k] = T;
            ca += w[k]; cs += q1[k] * w[k]; cd += zi[k] * w[k];
        }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { ca += __shfl_xor(ca, d); cs += __shfl_xor(cs, d); cd += __shfl_xor(cd, d); }
    const float inv = 1.0f / (ca + 1e-8f), depth = cd * inv, sdfo = cs * inv;
    const float gc0 = g_color ? g_color[3 * r] : 0.0f, gc1 = g_color ? g_color[3 * r + 1] : 0.0f, gc2 = g_color ? g_color[3 * r + 2] : 0.0f;
    const float gd = g_depth ? g_depth[r] : 0.0f, ga = g_alpha ? g_alpha[r] : 0.0f, gs = g_sdf ? g_sdf[r] : 0.0f;
    float gw[SPL], lane_sum = 0.0f;
#pragma unroll
    for (int k = 0; k < SPL; ++k) {
        gw[k] = (gc0 * q2[k] + gc1 * q3[k] + gc2 * q4[k]) + ga + gd * (zi[k] - depth) * inv + gs * (q1[k] - sdfo) * inv;
        lane_sum += gw[k] * w[k];
    }
    // exclusive suffix sum of lane_sum over the lanes (what the lanes behind this one contribute)
    float sincl = lane_sum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const float v = __shfl_down(sincl, d);
        if (lane + d < 64) sincl += v;
    }
    float Sx = __shfl_down(sincl, 1);
    if (lane == 63) Sx = 0.0f;
    float dbeta = 0.0f;
#pragma unroll
    for (int k = SPL - 1; k >= 0; --k) {
        const int i = lane * SPL + k;
        if (i < S) {
            const float dsig = dist[k] * (gw[k] * Tn[k] - Sx); // (last sample: dist = 1e10, T_{S} and the suffix are what they are -- torch's own product)
            const float ss = sg[k] * (1.0f - sg[k]);
            const float b2 = beta * beta;
            float* o = from[k] >= 0 ? d_a + ((size_t)r * Sa + from[k]) * 5 : d_b + ((size_t)r * Sb + (~from[k])) * 5;
            o[0] = -dsig * ss / b2;
            o[1] = gs * w[k] * inv;
            o[2] = gc0 * w[k]; o[3] = gc1 * w[k]; o[4] = gc2 * w[k];
            dbeta += dsig * (ss * x[k] / (b2 * beta) - sg[k] / b2);
        }
        Sx += gw[k] * w[k];
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) dbeta += __shfl_xor(dbeta, d);
    if (lane == 0 && d_beta) d_beta[r] = dbeta;
}

extern "C" int vanerf_composite_backward(const VanerfWeights* w, const float* rgba, const float* z, const float* mesh_sdf, int Sa, const float* rgba_n,
                                         const float* mesh_sdf_n, int Sn, const int32_t* src, int R, const float* g_color, const float* g_depth,
                                         const float* g_alpha, const float* g_sdf, float* d_rgba, float* d_rgba_n, float* d_beta, void* stream)
{
    return guarded([&] {
        if (!w || !w->dev_beta || !rgba || !z || !mesh_sdf || !d_rgba) throw_error("vanerf_composite_backward: null argument");
        if (rgba_n && (!mesh_sdf_n || !src || Sn <= 0 || !d_rgba_n)) throw_error("vanerf_composite_backward: the second table needs mesh_sdf_n, src, d_rgba_n and Sn > 0");
        const int Sb = rgba_n ? Sn : 0, S = Sa + Sb;
        if (R <= 0 || Sa <= 0 || S > 256) throw_error("vanerf_composite_backward: R=%d, %d samples per ray (at most 256)", R, S);
        const dim3 wg((R + 3) / 4), wb(256);
#define VANERF_CB(SPL)                                                                                                                              \
    hipLaunchKernelGGL(composite_backward_kernel<SPL>, wg, wb, 0, (hipStream_t)stream, rgba, z, mesh_sdf, rgba_n, rgba_n ? mesh_sdf_n : nullptr,          \
                       rgba_n ? src : nullptr, Sa, Sb, R, S, w->dev_beta, g_color, g_depth, g_alpha, g_sdf, d_rgba, d_rgba_n, d_beta)
        if (S <= 64) VANERF_CB(1);
        else if (S <= 128) VANERF_CB(2);
        else if (S <= 192) VANERF_CB(3);
        else VANERF_CB(4);
#undef VANERF_CB
        HIP_CHECK(hipGetLastError());
    });
}

// pass.cpp's composites: sigmoid_beta from the weight handle's device copy (see vanerf_weights_update); rgba_b == null: one table of S = Sa samples
void vanerf::composite_with_handle(const VanerfWeights* w, const float* rgba, const float* z, const float* msdf, const float* rgba_b, const float* msdf_b,
                                   const int32_t* src, int Sa, int Sb, int R, float* color, float* depth, float* alpha, float* sdf, float* contrib,
                                   void* stream)
{
    launch_composite(rgba, z, msdf, rgba_b, msdf_b, src, Sa, Sb, R, Sa + Sb, w->beta, w->dev_beta, color, depth, alpha, sdf, contrib, stream);
}

extern "C" int vanerf_composite(const float* rgba, const float* z, const float* mesh_sdf, int R, int S, float beta,
                                float* color, float* depth, float* alpha, float* sdf, float* contrib, void* stream)
{
    return guarded([&] {
        if (!rgba || !z || !mesh_sdf || !color || !depth || !alpha || !sdf) throw_error("vanerf_composite: null argument");
        launch_composite(rgba, z, mesh_sdf, nullptr, nullptr, nullptr, S, 0, R, S, beta, nullptr, color, depth, alpha, sdf, contrib, stream);
    });
}

// Either composite with sigmoid_beta taken from a weight handle's device copy (what vanerf_render_pass does): rgba_n == NULL composites the
// Sa samples of one table, otherwise the merged order of two tables as vanerf_composite_merged.
extern "C" int vanerf_composite_handle(const VanerfWeights* w, const float* rgba, const float* z, const float* mesh_sdf, int Sa, const float* rgba_n,
                                       const float* mesh_sdf_n, int Sn, const int32_t* src, int R, float* color, float* depth, float* alpha,
                                       float* sdf, float* contrib, void* stream)
{
    return guarded([&] {
        if (!w || !rgba || !z || !mesh_sdf || !color || !depth || !alpha || !sdf) throw_error("vanerf_composite_handle: null argument");
        if (rgba_n && (!mesh_sdf_n || !src || Sn <= 0)) throw_error("vanerf_composite_handle: the second table needs mesh_sdf_n, src and Sn > 0");
        vanerf::composite_with_handle(w, rgba, z, mesh_sdf, rgba_n, rgba_n ? mesh_sdf_n : nullptr, rgba_n ? src : nullptr, Sa, rgba_n ? Sn : 0, R, color,
                                      depth, alpha, sdf, contrib, stream);
    });
}

extern "C" int vanerf_composite_merged(const float* rgba_c, const float* mesh_sdf_c, int Sc, const float* rgba_n, const float* mesh_sdf_n,
                                       int Sn, const int32_t* src, const float* z_fine, int R, float beta, float* color, float* depth,
                                       float* alpha, float* sdf, float* contrib, void* stream)
{
    return guarded([&] {
        if (!rgba_c || !mesh_sdf_c || !rgba_n || !mesh_sdf_n || !src || !z_fine || !color || !depth || !alpha || !sdf)
            throw_error("vanerf_composite_merged: null argument");
        if (Sc <= 0 || Sn <= 0) throw_error("vanerf_composite_merged: Sc=%d Sn=%d", Sc, Sn);
        launch_composite(rgba_c, z_fine, mesh_sdf_c, rgba_n, mesh_sdf_n, src, Sc, Sn, R, Sc + Sn, beta, nullptr, color, depth, alpha, sdf, contrib, stream);
    });
}

// importance_sample + sort-merge with one WAVE per ray (Sc, Sf <= 64 * EPL: lane l holds coarse samples and new samples l, l + 64, ...).
// The rows of a ray are read and written coalesced (one thread per ray touched 64 cache lines per access), the pdf total is a wave
// reduction and the cdf a wave scan in fp64 (the sequential fp64 sums they replace differ from them by ~1e-16 relative before the rounding
// to fp32), each lane draws its samples by binary search in the LDS copy of the cdf, and the merge is by rank: coarse sample a goes to
// a + #(new < z_a), new sample b to b + #(coarse <= z_b) -- the stable merge of the serial kernel (coarse first on ties).  Unsorted draws
// (random u, or a last-bit inversion) are rank-sorted first, stably, like the insertion sort they replace.
constexpr int IW_RAYS = 4; // waves (rays) per block
template <int EPL>
__global__ __launch_bounds__(64 * IW_RAYS) void importance_merge_wave_kernel(const float* __restrict__ contrib, const float* __restrict__ z,
                                                                             const float* __restrict__ u, const float* __restrict__ t_lin, int R,
                                                                             int Sc, int Sf, float* __restrict__ z_new, float* __restrict__ z_fine,
                                                                             int32_t* __restrict__ src, int32_t* __restrict__ idx_out)
{
    constexpr int N = 64 * EPL;
    __shared__ float s_f[IW_RAYS][4][N];
    __shared__ unsigned s_k[IW_RAYS][3][N];
    __shared__ int s_p[IW_RAYS][N];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int r = blockIdx.x * IW_RAYS + wv;
    if (r >= R) return; // whole wave
    float* cdf = s_f[wv][0];
    float* zmid = s_f[wv][1];
    float* zc = s_f[wv][2];    // coarse depths (INFINITY beyond Sc)
    float* smp = s_f[wv][3];   // new samples in draw order
    unsigned* kn = s_k[wv][0];   // keys of the draws, draw order
    unsigned* kc = s_k[wv][1];   // keys of the coarse depths
    unsigned* ksrt = s_k[wv][2]; // keys of the draws, sorted
    int* perm = s_p[wv];
    // the wave's own LDS traffic only (no other wave touches this slice): wait for it, and keep the compiler from moving accesses across
    auto lds_sync = [] { __builtin_amdgcn_s_waitcnt(0xc07f); __builtin_amdgcn_wave_barrier(); };
    const int nb = Sc - 2;
    float ci[EPL], zi[EPL];
    double tot = 0.0;
#pragma unroll
    for (int k = 0; k < EPL; ++k) {
        const int i = k * 64 + lane;
        ci[k] = i < Sc ? contrib[(size_t)r * Sc + i] : 0.0f;
        zi[k] = i < Sc ? z[(size_t)r * Sc + i] : INFINITY;
        zc[i] = zi[k];
        if (i >= 1 && i <= nb) tot += (double)(ci[k] + 1e-5f);
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) tot += __shfl_xor(tot, d);
    const float sum = (float)tot;
    lds_sync();
    double carry = 0.0;
#pragma unroll
    for (int k = 0; k < EPL; ++k) {
        const int i = k * 64 + lane;
        double run = (i >= 1 && i <= nb) ? (double)((ci[k] + 1e-5f) / sum) : 0.0;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const double v = __shfl_up(run, d);
            if (lane >= d) run += v;
        }
        run += carry;
        carry = __shfl(run, 63);
        if (i <= nb) { cdf[i] = i == 0 ? 0.0f : (float)run; zmid[i] = 0.5f * (zc[i + 1] + zi[k]); }
    }
    lds_sync();
    // Order: a total order on the bit patterns (key), so that the positions below always form a permutation of 0 .. Sc+Sf-1 whatever the
    // values are -- a camera whose far plane lies in front of the bounding box yields DESCENDING coarse depths (near > far, the reference then
    // sorts everything: th.sort(th.cat([z, z_fine])), src/model.py:1303), and a NaN anywhere must not leave a slot of `src` unwritten (the
    // composite gathers through it).  key(a) < key(b) <=> a < b for ordinary floats; NaNs sort last.
    auto key = [](float f) { const unsigned b = __float_as_uint(f); return b ^ ((b >> 31) ? 0xffffffffu : 0x80000000u); };
    float s[EPL];
    unsigned ks[EPL], kz[EPL];
#pragma unroll
    for (int k = 0; k < EPL; ++k) {
        const int j = k * 64 + lane;
        s[k] = INFINITY;
        if (j < Sf) {
            const float uk = u ? u[(size_t)r * Sf + j] : t_lin[j];
            int lo = 0, hi = nb + 1; // searchsorted(cdf, uk, right=True): first position with cdf > uk
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (cdf[mid] <= uk) lo = mid + 1; else hi = mid;
            }
            const int ip = max(lo - 1, 0), in = min(lo, nb);
            const float cp = cdf[ip], cn = cdf[in], zp = zmid[ip], zq = zmid[in];
            float den = cn - cp;
            if (den < 1e-5f) den = 1.0f;
            s[k] = zp + ((uk - cp) / den) * (zq - zp);
            z_new[(size_t)r * Sf + j] = s[k];
            if (idx_out) idx_out[(size_t)r * Sf + j] = in;
        }
        ks[k] = key(s[k]);
        kz[k] = key(zi[k]);
        smp[j] = s[k];
        kn[j] = ks[k];
        kc[j] = kz[k];
    }
    lds_sync();
    bool new_unsorted = false, coarse_unsorted = false;
#pragma unroll
    for (int k = 0; k < EPL; ++k) {
        const int j = k * 64 + lane;
        new_unsorted = new_unsorted || __ballot(j >= 1 && j < Sf && ks[k] < kn[max(j - 1, 0)]) != 0ull;
        coarse_unsorted = coarse_unsorted || __ballot(j >= 1 && j < Sc && kz[k] < kc[max(j - 1, 0)]) != 0ull;
    }
    float* frow = z_fine + (size_t)r * (Sc + Sf);
    int32_t* srow = src + (size_t)r * (Sc + Sf);
    if (!coarse_unsorted) {
#pragma unroll
        for (int k = 0; k < EPL; ++k) {
            const int j = k * 64 + lane;
            int rank = j;
            if (new_unsorted) { // stable rank of this draw among the draws (the insertion sort of the serial kernel)
                rank = 0;
                for (int jj = 0; jj < Sf; ++jj) { const unsigned o = kn[jj]; rank += (o < ks[k] || (o == ks[k] && jj < j)) ? 1 : 0; }
            }
            if (j < Sf) { ksrt[rank] = ks[k]; perm[rank] = j; }
        }
        lds_sync();
#pragma unroll
        for (int k = 0; k < EPL; ++k) {
            const int j = k * 64 + lane;
            if (j < Sc) { // coarse sample: position = j + #(draws < it)
                int lo = 0, hi = Sf;
                while (lo < hi) { const int mid = (lo + hi) >> 1; if (ksrt[mid] < kz[k]) lo = mid + 1; else hi = mid; }
                frow[j + lo] = zi[k]; srow[j + lo] = j;
            }
            if (j < Sf) { // draw j of the sorted order: position = j + #(coarse <= it)
                const unsigned v = ksrt[j];
                const int pl = perm[j];
                int lo = 0, hi = Sc;
                while (lo < hi) { const int mid = (lo + hi) >> 1; if (kc[mid] <= v) lo = mid + 1; else hi = mid; }
                frow[j + lo] = smp[pl]; srow[j + lo] = ~pl;
            }
        }
    } else { // coarse depths not ascending: full stable sort of [coarse | draws] by counting
#pragma unroll
        for (int k = 0; k < EPL; ++k) {
            const int j = k * 64 + lane;
            int pc = 0, pn = 0;
            for (int jj = 0; jj < Sc; ++jj) {
                const unsigned o = kc[jj];
                pc += (o < kz[k] || (o == kz[k] && jj < j)) ? 1 : 0; // coarse before coarse
                pn += (o <= ks[k]) ? 1 : 0;                           // coarse before a draw (coarse first on ties)
            }
            for (int jj = 0; jj < Sf; ++jj) {
                const unsigned o = kn[jj];
                pc += (o < kz[k]) ? 1 : 0;
                pn += (o < ks[k] || (o == ks[k] && jj < j)) ? 1 : 0;
            }
            if (j < Sc) { frow[pc] = zi[k]; srow[pc] = j; }
            if (j < Sf) { frow[pn] = s[k]; srow[pn] = ~j; }
        }
    }
}


extern "C" int vanerf_importance_merge(const float* contrib, const float* z, const float* u, const float* t_lin, int R, int Sc, int Sf,
                                       float* z_new, float* z_fine, int32_t* src, int32_t* idx, void* stream)
{
    return guarded([&] {
        if (!contrib || !z || !z_new || !z_fine || !src) throw_error("vanerf_importance_merge: null argument");
        if (!u && !t_lin) throw_error("vanerf_importance_merge: need u (random) or t_lin (uniform)");
        if (R <= 0 || Sc < 3 || Sf < 1) throw_error("vanerf_importance_merge: R=%d Sc=%d Sf=%d", R, Sc, Sf);
        const int smax = Sc > Sf ? Sc : Sf;
        if (smax <= 256) { // one wave per ray
            const dim3 wg((R + IW_RAYS - 1) / IW_RAYS), wb(64 * IW_RAYS);
#define VANERF_IMPORTANCE_WAVE(EPL)                                                                                                      \
    hipLaunchKernelGGL(importance_merge_wave_kernel<EPL>, wg, wb, 0, (hipStream_t)stream, contrib, z, u, t_lin, R, Sc, Sf, z_new, z_fine, src, idx)
            if (smax <= 64) VANERF_IMPORTANCE_WAVE(1);
            else if (smax <= 128) VANERF_IMPORTANCE_WAVE(2);
            else VANERF_IMPORTANCE_WAVE(4);
#undef VANERF_IMPORTANCE_WAVE
            HIP_CHECK(hipGetLastError());
            return;
        }
        const size_t lds = (size_t)IM_BLOCK * sizeof(float) * (2 * (size_t)(Sc - 1) + 2 * (size_t)Sf);
        if (lds > 160 * 1024) throw_error("vanerf_importance_merge: %d + %d samples per ray exceed LDS", Sc, Sf);
        if (lds > 64 * 1024)
            HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(importance_merge_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(importance_merge_kernel<false>, dim3((R + IM_BLOCK - 1) / IM_BLOCK), dim3(IM_BLOCK), lds, (hipStream_t)stream,
                           contrib, z, u, t_lin, R, Sc, Sf, z_new, z_fine, src, idx);
        HIP_CHECK(hipGetLastError());
    });
}

extern "C" int vanerf_importance_sample(const float* contrib_inner, const float* z_mid, const float* u, const float* t_lin, int R, int n_bins,
                                        int Sf, float* z_new, int32_t* idx, void* stream)
{
    return guarded([&] {
        if (!contrib_inner || !z_mid || !z_new) throw_error("vanerf_importance_sample: null argument");
        if (!u && !t_lin) throw_error("vanerf_importance_sample: need u (random) or t_lin (uniform)");
        if (R <= 0 || n_bins < 1 || Sf < 1) throw_error("vanerf_importance_sample: R=%d bins=%d Sf=%d", R, n_bins, Sf);
        const int Sc = n_bins + 2;
        const size_t lds = (size_t)IM_BLOCK * sizeof(float) * (2 * (size_t)(Sc - 1) + 2 * (size_t)Sf);
        if (lds > 160 * 1024) throw_error("vanerf_importance_sample: %d bins + %d samples per ray exceed LDS", n_bins, Sf);
        if (lds > 64 * 1024)
            HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(importance_merge_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(importance_merge_kernel<true>, dim3((R + IM_BLOCK - 1) / IM_BLOCK), dim3(IM_BLOCK), lds, (hipStream_t)stream,
                           contrib_inner, z_mid, u, t_lin, R, Sc, Sf, z_new, nullptr, nullptr, idx);
        HIP_CHECK(hipGetLastError());
    });
}

extern "C" int vanerf_ray_bbox(const float* bounds, const float* orig, const float* dirs, int R, float* near, float* far, uint8_t* hit, void* stream)
{
    return guarded([&] {
        if (!bounds || !orig || !dirs || !near || !far || !hit) throw_error("vanerf_ray_bbox: null argument");
        if (R <= 0) throw_error("vanerf_ray_bbox: R=%d", R);
        BboxParams B;
        std::copy_n(bounds, 6, B.bounds);
        std::copy_n(orig, 3, B.o);
        hipLaunchKernelGGL(ray_bbox_kernel, dim3((R + 255) / 256), dim3(256), 0, (hipStream_t)stream, B, dirs, R, near, far, hit);
        HIP_CHECK(hipGetLastError());
    });
}
