/*
 * vanerf_hip.h -- C ABI of the MI355X-native VANeRF volume-rendering hot path (libvanerf_hip.so).
 *
 * The reference (XuanHuang0/VANeRF) has no FFI of its own: its seam is Python methods on
 * `class VANeRF(torch.nn.Module)` (src/model.py:604).  Each entry point below replaces the
 * arithmetic of one of those methods (cited per function); `vanerf_amd/model.py` keeps the
 * Python signatures and calls these through ctypes.  INTEGRATION.md shows the binding a
 * maintainer of the reference would add.
 *
 * Conventions
 *   - every `const float*` / `float*` / `int*` below is a DEVICE pointer borrowed from the caller
 *     (contiguous, fp32 / int32 / uint8); the library allocates nothing that outlives a call
 *     except the packed-weight handle;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); no entry point
 *     synchronises the host, all work is enqueued on `stream`;
 *   - return value: 0 on success, negative on error (message via vanerf_last_error(), thread local);
 *   - B = V = 1 (one target view, one source view): the non-spconv reference path is
 *     structurally single-view (src/networks.py:86,94) and evaluates batch items in Python loops.
 */
#ifndef VANERF_HIP_H
#define VANERF_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VANERF_ABI_VERSION 10

#define VANERF_OK 0
#define VANERF_EINVAL (-22)
#define VANERF_ENOMEM (-12)
#define VANERF_EHIP (-5)

#define VANERF_NV 1558      /* vertices: 2 x (778 + 1 seal vertex), src/networks.py:25 */
#define VANERF_NV_HAND 779
#define VANERF_NKPT 42
#define VANERF_PE_LEVELS 3  /* sp_level, configs/vanerf.json:50 */

/* Host pointers to the per-sample network weights in the reference's state_dict layout
 * (row-major [out][in], Conv1d k=1 weights with the trailing 1 dropped).  Weight-normed layers
 * (src/utils.py:674-675) are passed as (v, g): the packer folds W = g * v / ||v||_row.          */
typedef struct {
    const float* geo_at0_w1;   /* [10][196]  geo_vis_fusion.fconv_at.0.weight   (src/networks.py:47-52) */
    const float* geo_at0_w2;   /* [3][10]    geo_vis_fusion.fconv_at.2.weight */
    const float* geo_ated0_w1; /* [64][196]  geo_vis_fusion.fconv_ated.0.weight (54-58) */
    const float* geo_ated0_w2; /* [64][64]   geo_vis_fusion.fconv_ated.2.weight */
    const float* geo_at1_w1;   /* [10][28]   geo_vis_fusion.fconv_at1.0.weight  (60-65) */
    const float* geo_at1_w2;   /* [3][10] */
    const float* geo_ated1_w1; /* [8][28]    geo_vis_fusion.fconv_ated1.0.weight (67-71) */
    const float* geo_ated1_w2; /* [8][8] */
    /* mlp_geo (src/utils.py:609-649): layers1 = [358->128, 128->128, 136->120, 120->64], layers2 = [128->64, 64->64, 64->2] */
    const float* l1_v[3];      /* weight_v of layers1.{0,1,2}: [128][358], [128][128], [120][136] */
    const float* l1_g[3];      /* weight_g: [128], [128], [120] */
    const float* l1_b[3];      /* bias */
    const float* l1_w3;        /* [64][120] layers1.3 (plain) */
    const float* l1_b3;        /* [64] */
    const float* l2_v[2];      /* layers2.{0,1}: [64][128], [64][64] */
    const float* l2_g[2];
    const float* l2_b[2];
    const float* l2_w2;        /* [2][64] layers2.2 (plain) */
    const float* l2_b2;        /* [2] */
    const float* ibr_w;        /* [24][128] ibr_compress_gfeat (src/model.py:633-636) */
    const float* ibr_b;        /* [24] */
    const float* tex_at_w1;    /* [96][96]  tex_vis_fusion.fconv_at.0.weight (src/networks.py:230-235) */
    const float* tex_at_w2;    /* [6][96] */
    const float* tex_w1;       /* [96][96]  tex_vis_fusion.fconv.0.weight (224-228) */
    const float* tex_w2;       /* [40][96]  (only rows 0..2 reach the image at V = 1: src/model.py:1613,1635) */
    float sigmoid_beta;        /* VANeRF.sigmoid_beta (src/model.py:614); clamped to >= 2e-3 as in sdf_activation (880) */
} VanerfWeightTable;

/* Per-source-frame inputs of the per-sample networks (all device pointers, fp32). */
typedef struct {
    const float* geo0;      /* [h0][w0][64] feat_geo[0], channel-last          (src/model.py:826, 971) */
    const float* geo1;      /* [h1][w1][8]  feat_geo[1], channel-last */
    const float* tex;       /* [ht][wt][8]  feat_tex, channel-last             (src/model.py:919, 972) */
    const float* img;       /* [hi][wi][4]  source image rgb + pad             (src/model.py:906) */
    const float* mask;      /* [hi][wi]     src_foreground_mask                (src/model.py:800) */
    int h0, w0, h1, w1, ht, wt, hi, wi;
    const float* verts;     /* [NV][4] vert_world xyz + pad */
    const float* vfeat0;    /* [NV][64] feat_sample(feat_geo[0], vert_xy) * vert_vis   (src/networks.py:83, 29) */
    const float* vfeat1;    /* [NV][8]  feat_sample(feat_geo[1], vert_xy) * vert_vis   (src/networks.py:96) */
    const float* vfeat_tex; /* [NV][32] [img3 | tex8 | gf18 | 0 0 0] * vert_vis        (src/networks.py:270-279) */
    const float* vert_vis;  /* [NV]     {0,1}                                          (mesh_util.py:284-318) */
    const float* kpt_cam;   /* [42][4]  key points in source-camera coordinates (src/spatial.py:84) */
    float KRT[12];          /* source view K*[R|t], rows 0..2 (src/model.py:780) */
    float extrin[12];       /* source view [R|t]                (src/spatial.py:71-72) */
    float width, height, znear, zfar; /* cam_in (src/model.py:313-317) */
    float invalid_sdf;      /* 0.1 / nml_scale (src/model.py:1144) */
    float pe_scale, pe_inv_2sigma2; /* sp_args.scale, 1/(2 sigma^2) (src/spatial.py:110-113) */
} VanerfFrame;

typedef struct VanerfWeights VanerfWeights; /* opaque device-resident packed weights */

int vanerf_abi_version(void);
const char* vanerf_last_error(void);

/* Packs the weight table into MFMA fragment order on the device.  mode selects the arithmetic of the 20 dense layers that
 * vanerf_query_samples runs with these weights:
 *   0  fp32:   v_mfma_f32_32x32x2_f32 on fp32 weights and activations (summation order is the only difference to the reference)
 *   1  bf16x3: v_mfma_f32_32x32x16_bf16, fp32 accumulate, W = W_hi + W_lo and X = X_hi + X_lo (bf16 each), three products per term;
 *              outputs within 3.3e-5 of mode 0 on the 10.9 M-sample benchmark launch, about twice as fast */
int vanerf_weights_pack(const VanerfWeightTable* w, int mode, VanerfWeights** out);
int vanerf_weights_free(VanerfWeights* w);
/* Re-packs a handle IN PLACE from parameters that live on the DEVICE (the training step: the optimiser has just changed them).  `dev` is a
 * weight table of device pointers to the parameters' own storage (fp32, contiguous); sigmoid_beta_dev points at the parameter on the device
 * (NULL: dev->sigmoid_beta, a host value, is taken).  A few launches on `stream`, no copy to the host, no allocation after the first call, nothing
 * blocks; the streams are bit-identical to a fresh vanerf_weights_pack of the same values.  The caller orders the update after the launches
 * that still read the old weights (same stream, or an event).                                                                                  */
int vanerf_weights_update(VanerfWeights* w, const VanerfWeightTable* dev, const float* sigmoid_beta_dev, void* stream);
/* Diagnostics: number of 32-sample groups (since the pack) for which vanerf_query_samples took its all-invalid short path
 * (only the colour branch evaluated).  Blocking device read; for benchmarks and tests.                                        */
int vanerf_weights_short_groups(const VanerfWeights* w, uint64_t* count);
/* Host-only view of the packed fragment stream (no GPU touched): out[cap] or NULL to query the size; offsets[20]. */
int vanerf_weights_pack_host(const VanerfWeightTable* w, float* out, int64_t cap, int64_t* n_out, unsigned* offsets);
/* Host-only view of any stream a handle can carry: which = 0 the fp32 forward stream, 1 the bf16x3 forward stream (32-bit words of two bf16
 * each, copied raw), 2 the transposed fp32 stream of the fused backward pass.  out[cap] or NULL to query the size.                            */
int vanerf_weights_stream_host(const VanerfWeightTable* w, int which, float* out, int64_t cap, int64_t* n_out);
/* The streams a handle holds on the device, copied back to host memory (blocking; for tests of vanerf_weights_update): which = 0 the forward
 * stream of the handle's mode, 2 the backward stream (fp32 handles only).                                                                       */
int vanerf_weights_download(const VanerfWeights* w, int which, float* out, int64_t cap, int64_t* n_out);

/* a1-a4  Pixel grid, ray generation, bbox clipping, coarse depths (src/model.py:1191-1238, 1496-1570).
 *   grid: x = x0 + ix*step_x, y = y0 + (iy / y_block)*step_y + (iy % y_block)*step_x for iy < ny (outer), ix < nx (inner); R = nx*ny rays
 *         (the reference's strided grid: step_x = step_y = 2^(level-1), y_block = 1, model.py:1194; a multi-GPU shard takes blocks of
 *          y_block = 8 consecutive rows, step_x = 1, step_y = 8 N, y0 = 8 rank: 8x8 pixel tiles stay whole, the load stays balanced)
 *   invK_T[9]: inverse(K[:3,:3]) transposed, row-major (host computed, as th.inverse at model.py:1208)
 *   RT[12]: target [R|t] rows 0..2;  bounds[6] = {min xyz, max xyz} (host values)
 *   t_lin[S]: th.linspace(0, 1, S) (device; computed by the caller so that it is bit-identical to torch's)
 *   outputs: index[R] (int64 pixel index x + y*W), rays_d[R][3], cam_pos[3] (device), near[R], far[R], hit[R] (u8),
 *            z[R][S] coarse depths (uniform=True: t_lin; else stratified with jitter[R][S] in [0,1) drawn by the caller) */
int vanerf_ray_setup(int x0, int y0, int step_x, int step_y, int y_block, int nx, int ny, int width, const float* invK_T, const float* RT,
                     float znear, float zfar, const float* bounds, int S, const float* t_lin, const float* jitter,
                     int64_t* index, float* rays_d, float* cam_pos, float* near, float* far, uint8_t* hit, float* z,
                     void* stream);

/* ... with the rows as a list of blocks: row iy = row_blocks[iy / y_block] + (iy % y_block) * step_x (device table of ny / y_block entries): the
 * shard of a multi-GPU view whose blocks of 8 rows were dealt by cost instead of round robin (vanerf_amd/parallel.py: deal_blocks)          */
int vanerf_ray_setup_blocks(const int32_t* row_blocks, int x0, int step_x, int y_block, int nx, int ny, int width, const float* invK_T,
                            const float* RT, float znear, float zfar, const float* bounds, int S, const float* t_lin, const float* jitter,
                            int64_t* index, float* rays_d, float* cam_pos, float* near, float* far, uint8_t* hit, float* z, void* stream);
/* Same with an explicit pixel list pixels_xy[n_rays][2] (int32, device): the training branch's clamped 64x64 window around a
 * random mask pixel (src/model.py:1172-1189) is not a regular grid.                                                         */
int vanerf_ray_setup_pixels(const int32_t* pixels_xy, int n_rays, int width, const float* invK_T, const float* RT, float znear,
                            float zfar, const float* bounds, int S, const float* t_lin, const float* jitter, int64_t* index,
                            float* rays_d, float* cam_pos, float* near, float* far, uint8_t* hit, float* z, void* stream);

/* eval_pts = cam_pos + dir * z (src/model.py:1234-1235).  pts[R*S][3]. */
int vanerf_sample_points(const float* rays_d, const float* cam_pos, const float* z, int R, int S, float* pts, void* stream);

/* a6  get_visibility (mesh_util.py:284-318): vert_xy01[NV][2], vert_z01[NV], faces[NF][3] int32 -> vert_vis[NV];
 *     pix_to_face[S*S] int32 is scratch (may be NULL only if `scratch` given).                                          */
int vanerf_vertex_visibility(const float* vert_xy01, const float* vert_z01, int nv, const int32_t* faces, int nf,
                             int raster, int32_t* pix_to_face, float* vert_vis, void* stream);

/* a6  cal_vis_sdf_batch (mesh_util.py:498-524): signed distance, closest face, interpolated visibility.
 *     verts[NV][3], faces[NF][3] int32, vert_vis[NV], pts[N][3] -> sdf[N], vis[N] (u8), face[N] (int32, may be NULL)     */
int vanerf_mesh_query(const float* verts, int nv, const int32_t* faces, int nf, const float* vert_vis,
                      const float* pts, int64_t n, float* sdf, uint8_t* vis, int32_t* face, void* stream);

/* Acceleration structure for vanerf_mesh_query_accel, built per source frame by vanerf_mesh_accel_build below.
 * Results are bit-identical to vanerf_mesh_query.                                                                        */
/* Triangles / vertices per cluster the library was built with. */
int vanerf_mesh_cluster_size(void);

typedef struct {
    const float* tri;          /* [nfp][9]  triangle corners, Morton-sorted, padded to a multiple of the cluster size with far-away triangles */
    const float* sphere;       /* [nfp][4]  bounding sphere of each triangle (centroid, radius) */
    const float* tnorm;        /* [nfp][4]  unit normal of each triangle (0 for a degenerate one), w = rounding allowance 1e5 a^2: with `sphere` the triangle lies in the
                                *           disc {centroid + u : u normal to tnorm, |u| <= radius} -- the lower bound the search prunes with */
    const int32_t* orig;       /* [nfp]     original face index (INT32_MAX for padding) */
    const float* cbox;         /* [nc][6]   AABB of each cluster of vanerf_mesh_cluster_size() consecutive triangles (lo xyz, hi xyz) */
    int nfp, nc;
    const int32_t* cell_start; /* [G*G + 1] CSR offsets of the (y,z) grid used by the inside test */
    const int32_t* cell_tri;   /* original face ids per cell */
    const float* cell_rec;     /* [entries][12] the same lists as self-contained records: vertex ids i0 i1 i2 (int bits), corners a b c -- one load per entry in the inside test */
    const float* grid;         /* [5] DEVICE record of that grid: y0, z0, cell size in y, in z, G as int bits.  cell = clamp(floor((c - c0) / size), 0, G-1) */
    const float* vsort;        /* [nvc*CL][4] Morton-sorted vertices (xyz, original index as int bits), padded with far points */
    const float* vbox;         /* [nvc][6]    AABB of each cluster of CL vertices */
    int nvc;
    const float* cdisc;        /* [nc][8]  cylinder around each triangle cluster: (centre xyz, radius), (unit axis xyz, half height); the axis may be 0
                                *           (then radius alone bounds the distance from the centre: a ball).  Every vertex of the cluster's triangles
                                *           lies inside; second lower bound of the tile search of vanerf_mesh_query_accel (ray-grid hint) */
} VanerfMeshAccel;

/* Builds the tables above on the device, on `stream`, without a host synchronisation (seven small launches;
 * no counterpart in the reference: kaolin's point_to_mesh_distance / check_sign behind cal_vis_sdf_batch, src/lib/dataset/mesh_util.py:498-524, and
 * pytorch3d's knn_points, src/networks.py:28, scan all faces / vertices per point -- this is what lets vanerf_mesh_query_accel give their answers without; 0.1 ms for the two-hand MANO mesh):
 *     verts[NV][3], faces[NF][3] int32 (indices inside [0, NV): the caller's responsibility) -> *out, whose pointers point into `tables`,
 *     a caller-owned 16-byte-aligned device block of at least vanerf_mesh_accel_bytes(nv, nf, G, cell_capacity) bytes that must stay alive
 *     (and unmodified) for as long as *out is used.  G x G = cells of the (y,z) grid of the inside test (1..256; 64 for a hand mesh);
 *     cell_capacity >= nf = entries of the cells' triangle lists (64 * nf is ample: at G = 64 a triangle of a hand mesh touches ~20 cells).  Lists
 *     that would not fit make the grid degrade, on the device, to ONE cell listing every triangle: slower inside test, same results.
 *     Meshes of up to 16 384 faces / 4 096 vertices (the LDS-resident tables of vanerf_mesh_query_accel).                                  */
int64_t vanerf_mesh_accel_bytes(int nv, int nf, int G, int cell_capacity);   /* < 0: error code */
int vanerf_mesh_accel_build(const float* verts, int nv, const int32_t* faces, int nf, int G, int cell_capacity, void* tables,
                            int64_t tables_bytes, VanerfMeshAccel* out, void* stream);

/* knn_idx (may be NULL): 1-NN vertex of every point (knn_points K=1, src/networks.py:28), found in the same pass.
 * grid_nx, grid_ny, grid_s: optional layout hint (0,0,0 = none): pts are the samples of a grid_nx x grid_ny ray grid with grid_s
 * samples per ray in the reference's order (sample index = ray * grid_s + depth); lets a wave work on 64 neighbouring points.
 * The results do not depend on the hint.                                                                                      */
/* queue_word: 8 bytes of device memory owned by the caller for the duration of the launch (8-byte aligned; any contents): the head of the
 * kernel's work queue.  The library zeroes it on `stream` ahead of the launch and keeps NO device-side state of its own, so any number
 * of launches may be in flight on any number of streams as long as each has its own word (vanerf_render_pass takes them from its scratch). */
int vanerf_mesh_query_accel(const VanerfMeshAccel* accel, const float* verts, int nv, const int32_t* faces, int nf,
                            const float* vert_vis, const float* pts, int64_t n, float* sdf, uint8_t* vis, int32_t* face,
                            int32_t* knn_idx, int grid_nx, int grid_ny, int grid_s, void* queue_word, void* stream);

/* a10 knn_points K=1 (src/networks.py:28): verts[NV][4], pts[N][3] -> idx[N] int32 (first minimum). */
int vanerf_knn1(const float* verts4, int nv, const float* pts, int64_t n, int32_t* idx, void* stream);

/* a7-a15  VANeRF.query + eval_func for N samples (src/model.py:748-957, 1140-1160), n_views = 1:
 *     pts[N][3], query_sdf[N], query_vis[N] (u8), knn_idx[N] (1-NN vertex, from vanerf_mesh_query_accel or vanerf_knn1),
 *     noise[N] or NULL (rand_noise_std draws, model.py:1156)
 *     order[N] or NULL: a permutation of 0..N-1 from vanerf_query_order; slot k of the launch then works on sample order[k].  Inputs are
 *         read and outputs written at the sample's own index either way: the results do not depend on `order`, only the time does.
 *     raw = 0 -> out[N][5] = [alpha, sdf, r, g, b] (eval_func applied);  raw = 1 -> [sdf_pred, rad, r, g, b] as VANeRF.query returns
 *     valid[N] (u8, may be NULL)                                                                                          */
int vanerf_query_samples(const VanerfWeights* w, const VanerfFrame* frame, const float* pts, const float* query_sdf,
                         const uint8_t* query_vis, const int32_t* knn_idx, const float* noise, const int32_t* order, int raw, int64_t n,
                         float* out, uint8_t* valid, void* queue_word /* as for vanerf_mesh_query_accel */, void* stream);

/* Validity partition for vanerf_query_samples: order[N] = the samples whose projection hits the source view and its foreground mask
 * (src/model.py:780-803), in their order, then the others, in their order.  A 32-sample group of vanerf_query_samples in which no sample
 * is valid skips the geometry networks; with this order only one group per launch is mixed.  scratch: device memory of at least
 * vanerf_query_order_scratch(n) bytes.                                                                                          */
int vanerf_query_order(const VanerfFrame* frame, const float* pts, int64_t n, int32_t* order, void* scratch, int64_t scratch_bytes,
                       void* stream);
int64_t vanerf_query_order_scratch(int64_t n);

/* a16  sdf_activation + rgba2out (src/model.py:879-882, 1464-1494):
 *     rgba[R][S][5], z[R][S], mesh_sdf[R][S] -> color[R][3], depth[R], alpha[R], sdf[R], contrib[R][S] (may be NULL)      */
int vanerf_composite(const float* rgba, const float* z, const float* mesh_sdf, int R, int S, float beta,
                     float* color, float* depth, float* alpha, float* sdf, float* contrib, void* stream);

/* a16 + a18  Fine composite that re-uses the coarse evaluations: sample i of ray r is coarse sample src[r][i] (>= 0) or
 *     importance sample ~src[r][i] (< 0), in the merged depth order z_fine[R][Sc+Sn] produced by vanerf_importance_merge.
 *     Identical bits to vanerf_composite on a re-evaluated (R, Sc+Sn) batch (the networks are per-sample pure functions). */
int vanerf_composite_merged(const float* rgba_c, const float* mesh_sdf_c, int Sc, const float* rgba_n, const float* mesh_sdf_n,
                            int Sn, const int32_t* src, const float* z_fine, int R, float beta, float* color, float* depth,
                            float* alpha, float* sdf, float* contrib, void* stream);
/* Either composite with sigmoid_beta read from the weight handle's device copy (what vanerf_render_pass does; after vanerf_weights_update the host
 * never sees the value): rgba_n == NULL composites the Sa samples per ray of one table, otherwise [table | rgba_n table] in merged order (src).   */
/* eval_func (src/model.py:1140-1160) on raw outputs of vanerf_query_samples(raw = 1): [sdf_pred, rad, r, g, b] -> [alpha, sdf, r, g, b] with the validity
 * flags and optional per-sample noise, the arithmetic of the kernel's own epilogue (same bits).  src == NULL: R x Sa entries of table a, noise[i] per entry;
 * otherwise noise[r][p] belongs to position p of ray r's merged order and src names the entry (>= 0 table a, < 0 entry ~src of table b), each written once.
 * The outputs may be the raw tables themselves.  With training noise a pass evaluates the networks once per point and runs this once per set of draws.     */
int vanerf_eval_func(const float* raw_a, const uint8_t* valid_a, const float* raw_b, const uint8_t* valid_b, const int32_t* src, const float* noise,
                     int Sa, int Sb, int R, float invalid_sdf, float* rgba_a, float* rgba_b, void* stream);
int vanerf_composite_handle(const VanerfWeights* w, const float* rgba, const float* z, const float* mesh_sdf, int Sa, const float* rgba_n,
                            const float* mesh_sdf_n, int Sn, const int32_t* src, int R, float* color, float* depth, float* alpha, float* sdf,
                            float* contrib, void* stream);
/* Training: the backward of either composite (reference: autograd through rgba2out / sdf_activation, src/model.py:1464-1494, 879-882), tables and src as
 * vanerf_composite_handle.  g_color [R][3], g_depth, g_alpha, g_sdf [R]: gradients with respect to the composite's outputs (any may be NULL = 0);
 * d_rgba [R][Sa][5] (and d_rgba_n [R][Sn][5]): gradient with respect to every table entry (each is written once), d_beta [R] or NULL: the rays' shares
 * of the gradient with respect to the (clamped) sigmoid_beta.  At most 256 samples per ray.                                                            */
int vanerf_composite_backward(const VanerfWeights* w, const float* rgba, const float* z, const float* mesh_sdf, int Sa, const float* rgba_n,
                              const float* mesh_sdf_n, int Sn, const int32_t* src, int R, const float* g_color, const float* g_depth,
                              const float* g_alpha, const float* g_sdf, float* d_rgba, float* d_rgba_n, float* d_beta, void* stream);

/* a17  importance_sample + sort-merge (src/model.py:1424-1462, 1301-1307):
 *     contrib[R][Sc], z[R][Sc], u[R][Sf] (random draws) or NULL with t_lin[Sf] = th.linspace(0, 1, Sf) (uniform=True) ->
 *     z_new[R][Sf], z_fine[R][Sc+Sf] (sorted), src[R][Sc+Sf] int32 (>= 0: coarse sample id, < 0: ~(new sample id)),
 *     idx[R][Sf] int32 searchsorted index after clamping (may be NULL)                                                   */
int vanerf_importance_merge(const float* contrib, const float* z, const float* u, const float* t_lin, int R, int Sc, int Sf,
                            float* z_new, float* z_fine, int32_t* src, int32_t* idx, void* stream);

/* a17 in the reference's own call shape (src/model.py:1304, 1424-1462): contrib_inner[R][n_bins] = contrib[..., 1:-1],
 *     z_mid[R][n_bins+1] -> z_new[R][Sf]; idx[R][Sf] (searchsorted index after clamping, may be NULL).                     */
int vanerf_importance_sample(const float* contrib_inner, const float* z_mid, const float* u, const float* t_lin, int R, int n_bins,
                             int Sf, float* z_new, int32_t* idx, void* stream);

/* One whole pass -- VANeRF.batch_render_pifu_nerf without its GT-gather tail (src/model.py:1102-1360): rays, bbox clip, coarse depths,
 * mesh queries, validity partition, per-sample networks, composite, importance sampling + merge, fine march, fine composite -- as ONE call that
 * enqueues the kernels above on `stream` in the order vanerf_amd/renderer.py:render_pass does (same kernels, same arguments: same bits).
 * No allocation, no host synchronisation: all temporaries live in `scratch` (device memory of at least vanerf_render_pass_scratch(...) bytes,
 * caller-owned; two passes may run concurrently on different streams with different scratch blocks and the same weights handle).            */
typedef struct {
    int x0, y0, step_x, step_y, y_block, nx, ny; /* pixel grid as in vanerf_ray_setup (ignored when pixels_xy is given; then n_rays = nx * ny) */
    const int32_t* pixels_xy;  /* optional explicit pixel list [nx*ny][2] (device), as in vanerf_ray_setup_pixels */
    const int32_t* row_blocks; /* optional first rows of the ny / y_block blocks of rows (device), as in vanerf_ray_setup_blocks (y0, step_y ignored) */
    int width;                 /* target image width (pixel index = x + y * width) */
    float invK_T[9], RT[12];   /* target camera: inverse(K[:3,:3]) transposed; [R|t] rows 0..2 */
    float znear, zfar;
    float bounds[6];           /* {min xyz, max xyz} of the mesh bounding box (config['bounds']) */
    int Sc, Sf;                /* sample_per_ray_c, sample_per_ray_f */
    int fine;                  /* config['fine'] */
    int reuse_coarse;          /* 1: the fine composite re-uses the coarse evaluations (only the Sf new samples are evaluated; identical bits).  With
                                  noise_c / noise_f (training) the networks still run once per point: raw outputs, then eval_func with the coarse draws
                                  and again with the fine batch's draws (vanerf_eval_func) -- the noise is added to the networks' OUTPUT
                                  (src/model.py:1155-1156), so the bits are those of re-evaluating all Sc + Sf samples */
    const float* t_lin_c;      /* th.linspace(0, 1, Sc) (device) */
    const float* t_lin_f;      /* th.linspace(0, 1, Sf) (device); used when u == NULL (uniform=True) */
    const float* jitter;       /* [R][Sc] stratification draws or NULL (uniform=True) */
    const float* u;            /* [R][Sf] importance draws or NULL */
    const float* noise_c;      /* [R*Sc] rand_noise_std * randn draws for the coarse march, or NULL */
    const float* noise_f;      /* [R*(Sc+Sf)] the same for the fine march (required with noise_c when fine) */
} VanerfPassDesc;

typedef struct {               /* all device pointers; R = nx * ny */
    int64_t* index;            /* [R]    pixel index */
    uint8_t* hit;              /* [R]    ray crosses the bounding box */
    float* z;                  /* [R][Sc] coarse depths */
    float* color;              /* [R][3] coarse colour   (tex_fg)   */
    float* depth;              /* [R]                    (depth)    */
    float* alpha;              /* [R]                    (alpha)    */
    float* color_fine;         /* [R][3] (tex_fg_fine); the four fine outputs may be NULL when fine == 0 */
    float* depth_fine;         /* [R] */
    float* alpha_fine;         /* [R] */
    float* sdf;                /* [R] */
    float* z_fine;             /* [R][Sc+Sf] merged depths, or NULL */
} VanerfPassOut;

/* reuse_coarse: 0, 1 as VanerfPassDesc.reuse_coarse; 2 = re-use under per-sample noise (desc->reuse_coarse with noise_c given: more temporaries) */
int64_t vanerf_render_pass_scratch(int n_rays, int Sc, int Sf, int fine, int reuse_coarse);
int vanerf_render_pass(const VanerfWeights* w, const VanerfFrame* frame, const VanerfMeshAccel* accel, const float* verts, int nv,
                       const int32_t* faces, int nf, const VanerfPassDesc* desc, const VanerfPassOut* out, void* scratch, int64_t scratch_bytes,
                       void* stream);

/* Training step, backward of the row gathers (bilinear taps of feat_sample, src/utils.py:136-151; nearest / twin vertex rows of KNN_vis,
 * src/networks.py:27-33):  table[idx[i]][0..C) += w[i] * g[i][0..C)  for i < n  (w may be NULL = 1; rows outside [0, R) are ignored).
 * All device pointers, fp32 / int32; `table` is accumulated into.  Samples outnumber rows by hundreds: the adds go through LDS-resident
 * slices of the table instead of contended global atomics.                                                                               */
/* g_ld: floats between consecutive rows of g (0 = C: packed rows).                                                                        */
int vanerf_scatter_add_rows(const int32_t* idx, const float* w, const float* g, int64_t g_ld, int64_t n, int C, float* table, int R, void* stream);
/* the four bilinear taps of feat_sample (src/utils.py:136-151; border padding, align_corners) at xy[n][2] in [-1, 1] on an H x W map: idx4 / w4 = [4][n], the
 * row indices of the channel-last map and the weights vanerf_scatter_add_taps takes                                                              */
int vanerf_bilinear_taps(const float* xy, int64_t n, int H, int W, int32_t* idx4, float* w4, void* stream);
/* two sources into one table in one launch (the nearest and the twin vertex row of a sample): table[idx[i]] += w[i] g[i], table[idx2[i]] += w2[i] g2[i] */
int vanerf_scatter_add_rows2(const int32_t* idx, const float* w, const float* g, const int32_t* idx2, const float* w2, const float* g2, int64_t g_ld,
                             int64_t n, int C, float* table, int R, void* stream);
/* the same with FOUR (index, weight) pairs per sample, idx4 / w4 = [4][ld] (ld >= n): the backward of a bilinear tap gather (src/utils.py:136-151) */
int vanerf_scatter_add_taps(const int32_t* idx4, const float* w4, int64_t ld, const float* g, int64_t g_ld, int64_t n, int C, float* table, int R,
                            void* stream);

/* f-4, the fused backward pass of the per-sample networks (training; reference: autograd through VANeRF.query, src/model.py:748-957, driven by
 * training_step, src/model.py:381-459).  Two launches per block of n samples, on one stream, with an fp32 weight handle (vanerf_weights_pack mode 0):
 *   vanerf_query_forward_spill  the forward pass once more (raw outputs [sdf_pred, rad, r, g, b], validity), which also writes every layer's operands
 *                               X and the gates / pixel weight / pre-pooling latent the backward needs:  xs[x_rows][npad], aux[aux_rows][npad]
 *   vanerf_query_backward       d[n][5]: gradient with respect to eval_func's outputs [alpha, sdf, r, g, b] (src/model.py:1140-1160; d2 / noise /
 *                               noise2 may be NULL: a second set of draws on the same points, the density noise) with raw / valid of the spill pass
 *                               -> ys[y_rows][npad]: every layer's output gradient, and
 *                               ig[ig_rows * npad]: the gradients of the gathered inputs (pixel taps, nearest / twin vertex rows) as nine ROW-MAJOR
 *                               tensors [npad][stride] one after the other (vanerf_ig_tensor), which vanerf_scatter_add_rows / _taps read as they stand
 * xs, aux, ys: channel-major fp32 spills, npad = n rounded up to a multiple of 32 (columns >= n carry zero gradients); row counts: vanerf_spill_rows.
 * The weight gradient of layer l is then one matrix product over the samples, dW'[out][slot] = ys_l xs_l^T with the rows of
 * vanerf_layer_rows and the slot -> input-channel table of vanerf_layer_slots (slot 2 t + h; -1 unused, -2 bias: that column is the bias
 * gradient); vanerf_amd/hip_backward.py does this with torch.bmm and hands ig to vanerf_scatter_add_rows.                                */
int vanerf_query_forward_spill(const VanerfWeights* w, const VanerfFrame* frame, const float* pts, const float* query_sdf,
                               const uint8_t* query_vis, const int32_t* knn_idx, int64_t n, int64_t npad, float* out_raw, uint8_t* valid,
                               float* xs, float* aux, void* queue_word, void* stream);
int vanerf_query_backward(const VanerfWeights* w, const float* d, const float* d2, const float* noise, const float* noise2, const float* raw,
                          const uint8_t* valid, int64_t n, int64_t npad, const float* xs, const float* aux, float* ys, float* ig, void* stream);
int vanerf_spill_rows(int* x_rows, int* y_rows, int* aux_rows, int* ig_rows);
/* The weight gradients of a block: dw[slice][layer l at (sum of n_out x n_slots over the layers before l)][n_out][n_slots] += ys_l xs_l^T over the slice's
 * samples, every layer in one launch (slices: the block's samples are cut into this many parts that accumulate separately -- the caller sums dw over its
 * first dimension once per step; npad must be a multiple of 32 x slices; slices <= layout_slices, the number of parts dw has room for;
 * vanerf_weight_products_size: floats per part).  One wave per (tile group, slice): no atomics, reproducible.                                       */
int vanerf_weight_products(const float* xs, const float* ys, int64_t npad, int slices, int layout_slices, float* dw, void* stream);
int vanerf_weight_products_size(int64_t* floats_per_slice);
/* Tensor `which` of the ig spill: 0..2 pixel feature / nearest / twin vertex row of GeoVisFusion's scale 0 (64 channels), 3..5 the same of scale 1
 * (8), 6, 7 TexVisFusion's nearest / twin vertex row [img3 | tex8 | global18] (29 of 32), 8 the texture map's pixel feature (8).  The tensor
 * starts at ig + offset * npad, has `channels` channels and `stride` floats per row.  Returns the number of tensors (9), < 0: error.          */
int vanerf_ig_tensor(int which, int* offset, int* channels, int* stride);
int vanerf_layer_slots(int layer, int32_t* k_of_slot, int cap);              /* returns 2 T (k-pairs x lane halves), < 0: error */
int vanerf_layer_rows(int layer, int* x_row, int* y_row, int* n_out);

/* a3  ray_bbox_intersection (src/model.py:1496-1570) alone: bounds[6], orig[3] (host values), dirs[R][3] (device)
 *     -> near[R], far[R] (1.0 when the ray does not cross the box exactly twice), hit[R] (u8).                               */
int vanerf_ray_bbox(const float* bounds, const float* orig, const float* dirs, int R, float* near, float* far, uint8_t* hit, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* VANERF_HIP_H */
