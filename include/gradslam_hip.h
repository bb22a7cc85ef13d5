/* gradslam_hip.h -- C ABI of libgradslam_hip.so: the MI355X (gfx950) hot path of gradslam's
 * point-to-plane ICP odometry and PointFusion map update.
 *
 * The reference (EdwardjkFeng/gradslam) has no FFI for this path: it is pure Python on torch ops
 * (SURVEY.md section 8b).  Each entry point below therefore replaces a *chain of torch ops* in the
 * reference, cited as file:line relative to the reference root; INTEGRATION.md shows the ctypes
 * stub a gradslam maintainer would add at that call site.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer (HBM) unless its name starts with h_;
 *  - float tensors are fp32, row-major, densely packed; index tables are int64 rows [b,n,h,w]
 *    exactly as the reference's pc2im_bnhw; counts are int32;
 *  - images are channels-last: depth (B,L,H,W), maps (B,L,H,W,3); clouds are zero-padded
 *    (B,Nmax,C) with per-batch counts[B] (reference structures/pointclouds.py "padded" form);
 *  - 4x4 matrices are 16 contiguous floats, row-major;
 *  - `stream` is a hipStream_t passed as void*; all work is enqueued on it, nothing synchronises
 *    the host, nothing allocates (capturable in a hipGraph).  Scratch comes from the caller
 *    (`ws`, sized by the matching *_ws_bytes function);
 *  - return value: 0 on success, GS_ERR_* (<0) for an invalid argument, a positive hipError_t if
 *    a launch failed.  gs_last_error() describes the last failure of the calling thread.
 */
#ifndef GRADSLAM_HIP_H
#define GRADSLAM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GS_ABI_VERSION 3
#define GS_OK 0
#define GS_ERR_INVALID_ARG (-1)
#define GS_ERR_WORKSPACE_TOO_SMALL (-2)
#define GS_ERR_UNSUPPORTED (-3)

typedef void *gs_stream_t;

int gs_abi_version(void);
const char *gs_last_error(void);

/* ---------------------------------------------------------------- V: depth -> vertex/normal maps
 * Replaces RGBDImages._compute_vertex_map / _compute_normal_map / _compute_global_vertex_map /
 * _compute_global_normal_map (structures/rgbdimages.py:643-762) and inverse_intrinsics
 * (geometry/projutils.py:437-450), fused into one pass.  Any output pointer may be NULL.
 * poses == NULL reproduces the "poses is None -> clone" branch (global = local). */
int gs_vertex_normal_maps(const float *depth, const float *intrinsics /* B x 16 */,
                          const float *poses /* B*L x 16 or NULL */, int B, int L, int H, int W,
                          float *vertex, float *normal, float *gvertex, float *gnormal,
                          gs_stream_t stream);

/* Adjoint of the above: given d(loss)/d(map) for any subset of the four maps (NULL = zero),
 * accumulates d/d(depth) (B,L,H,W), d/d(intrinsics) (B x 16: fx,fy,cx,cy slots) and
 * d/d(poses) (B*L x 16: R and t slots).  Outputs must be zero-initialised by the caller. */
size_t gs_vertex_normal_maps_backward_ws_bytes(int B, int L, int H, int W);
int gs_vertex_normal_maps_backward(const float *depth, const float *intrinsics, const float *poses,
                                   int B, int L, int H, int W, const float *g_vertex,
                                   const float *g_normal, const float *g_gvertex,
                                   const float *g_gnormal, float *g_depth, float *g_intrinsics,
                                   float *g_poses, void *ws, size_t ws_bytes, gs_stream_t stream);

/* get_alpha (slam/fusionutils.py:69-73) on an (n,3) block: clamp(exp(-|p|^2/(2 sigma^2)), eps, 1.01) */
int gs_get_alpha(const float *points, int64_t n, float sigma, float eps, float *alpha,
                 gs_stream_t stream);
int gs_get_alpha_backward(const float *points, int64_t n, float sigma, float eps,
                          const float *g_alpha, float *g_points, gs_stream_t stream);

/* ---------------------------------------------------------------- dataset front-end: raw frames -> float
 * What the reference's dataset loaders do per frame on the host (datasets/tum.py:346, :455-499; icl.py:387;
 * scannet.py:189), done on the device from the raw integer frames so that 5 instead of 16 bytes per pixel
 * cross PCIe: depth (B,Hd,Wd) = uint16 (B,Hs,Ws) / depth_scale, nearest-neighbour resize; rgb (B,Hd,Wd,3) =
 * uint8 (B,Hs,Ws,3), bilinear resize (pixel centres at +0.5), optionally / 255.  Either input may be NULL. */
int gs_frames_from_raw(const uint16_t *depth_raw, const uint8_t *rgb_raw, int B, int Hs, int Ws, int Hd,
                       int Wd, float depth_scale, int normalise_color, float *depth, float *rgb,
                       gs_stream_t stream);

/* ---------------------------------------------------------------- generic stable compaction
 * out = rows of `src` (n_rows x row_floats fp32) whose mask byte is non-zero, order preserved;
 * *out_count = number kept.  Replaces the boolean-mask indexing x[mask] the reference uses at
 * odometry/icputils.py:654-668, slam/fusionutils.py:282,401,710-713, structures/utils.py:47-50. */
size_t gs_compact_ws_bytes(int64_t n_rows);
int gs_compact_rows(const float *src, const uint8_t *mask, int64_t n_rows, int row_floats,
                    float *out, int32_t *out_count, void *ws, size_t ws_bytes, gs_stream_t stream);

/* Same, for up to 4 row arrays sharing one mask and one scan (h_src / h_row_floats / h_out are HOST
 * arrays of n_arrays device pointers / widths): the append of fuse_with_map compacts vertex, normal,
 * colour and alpha rows of the new pixels in one go (slam/fusionutils.py:710-713). */
int gs_compact_multi(int n_arrays, const float *const *h_src, const int *h_row_floats,
                     float *const *h_out, const uint8_t *mask, int64_t n_rows, int32_t *out_count,
                     void *ws, size_t ws_bytes, gs_stream_t stream);
/* Append form: the selected rows of every h_src[a] go behind the *d_count rows h_dst[a] (capacity `cap` rows)
 * already holds; *d_count advances on the device (no host round trip), d_appended / d_overflow (optional)
 * receive the number of rows appended and a flag set when rows had to be dropped for lack of capacity.
 * This is Pointclouds.append_points (structures/pointclouds.py:1203-1235) for an arena-backed map. */
size_t gs_append_rows_ws_bytes(int64_t n_rows);
int gs_append_rows(int n_arrays, const float *const *h_src, const int *h_row_floats, float *const *h_dst,
                   const uint8_t *mask, int64_t n_rows, int32_t *d_count, int cap, int32_t *d_appended,
                   int32_t *d_overflow, void *ws, size_t ws_bytes, gs_stream_t stream);
/* Adjoint of gs_compact_multi (what autograd does for x[mask] in the reference): h_out[a] (n_rows, w_a)
 * receives the compacted adjoint row of every selected row and zeros everywhere else. */
int gs_expand_multi(int n_arrays, const float *const *h_grad, const int *h_row_floats,
                    float *const *h_out, const uint8_t *mask, int64_t n_rows, void *ws, size_t ws_bytes,
                    gs_stream_t stream);

/* ---------------------------------------------------------------- D: live frame -> ICP source cloud
 * downsample_rgbdimages (odometry/icputils.py:651-669): [::ds, ::ds] sub-grid of the global
 * vertex / normal maps and the rgb image of ONE frame per batch element (L == 1), valid-depth
 * pixels only, row-major order.  Outputs are padded (B, cap, 3) with cap >= ceil(H/ds)*ceil(W/ds);
 * counts[b] receives the number of rows written for batch b.  Any of the three outputs may be NULL. */
size_t gs_downsample_frame_ws_bytes(int H, int W, int ds);
int gs_downsample_frame(const float *depth, const float *gvertex, const float *gnormal,
                        const float *rgb, int B, int H, int W, int ds, int cap, float *out_points,
                        float *out_normals, float *out_colors, int32_t *out_pix /* (B,cap) ds-grid pixel
                        id r*ceil(W/ds)+c of every kept row, or NULL */, int32_t *counts, void *ws,
                        size_t ws_bytes, gs_stream_t stream);

/* ---------------------------------------------------------------- P: active map points
 * find_active_map_points (slam/fusionutils.py:247-282): inverse pose, transform every map point,
 * pinhole projection, in-frame test, round-half-to-even, rows [b,n,h,w] kept in (b,n) order.
 * points: padded (B,Nmax,3); counts[B]; poses/intrinsics: B x 16.  out_rows capacity: B*Nmax rows.
 * ds > 0 additionally keeps only rows with h%ds==0 && w%ds==0 (downsample_pointclouds' filter,
 * odometry/icputils.py:596-597); ds <= 0 keeps all. */
size_t gs_project_active_ws_bytes(int B, int Nmax);
int gs_project_active(const float *points, const int32_t *counts, int B, int Nmax,
                      const float *poses, const float *intrinsics, int H, int W, int ds,
                      int64_t *out_rows, int32_t *out_count, void *ws, size_t ws_bytes,
                      gs_stream_t stream);

/* S: downsample_pointclouds' gather (odometry/icputils.py:600-619): for batch element b, gather
 * attribute rows `attr[b, n]` for every table row with row.b == b, order preserved, into a padded
 * (B, cap, C) output; counts[b] = rows written.  n_rows is read from d_n_rows (device). */
size_t gs_gather_table_rows_ws_bytes(int B);
int gs_gather_table_rows(const int64_t *rows, const int32_t *d_n_rows, int64_t max_rows,
                         const float *attr, int B, int Nmax, int C, int cap, float *out,
                         int32_t *counts, void *ws, size_t ws_bytes, gs_stream_t stream);

/* Target cloud in pixel order: buckets the table rows (all on the ds-grid: gs_project_active with ds > 0)
 * of each batch element by their ds-grid pixel.  scan_points (B,cap,3) = map points in pixel order,
 * scan_orig (B,cap) = rank of each of them among the rows of its batch element (the index the reference's
 * downsample_pointclouds order gives it), pix_start (B, npix+1 with npix = ceil(H/ds)*ceil(W/ds)) = first
 * scan slot of every ds-grid pixel (pix_start[npix] = number of targets); tgt_pix (B,cap; optional, NULL to skip) =
 * ds-grid pixel of every row in the reference's order.  Feeds gs_icp_hints; changes no result, only the order in
 * which the exact search visits the target. */
size_t gs_bucket_by_pixel_ws_bytes(int B, int H, int W, int ds);
int gs_bucket_by_pixel(const int64_t *rows, const int32_t *d_n_rows, int64_t max_rows, int B, int H,
                       int W, int ds, const float *map_points, int Nmax, int cap, float *scan_points,
                       int32_t *scan_orig, int32_t *pix_start, int32_t *tgt_pix, void *ws, size_t ws_bytes,
                       gs_stream_t stream);

/* gs_gather_table_rows (points and normals) + gs_bucket_by_pixel fused into five launches: everything
 * gs_icp_point_to_plane needs of its target -- tgt / tgt_normals (B,cap,3) and counts (B) in the reference's
 * order, plus the search hints.  tgt_index (B,cap; optional, NULL to skip) receives the map index n of every
 * target slot (what the reverse pass scatters the target adjoints back with), tgt_pix (B,cap; optional) its
 * ds-grid pixel (gs_icp_hints.tgt_pix). */
size_t gs_build_icp_target_ws_bytes(int B, int H, int W, int ds);
int gs_build_icp_target(const int64_t *rows, const int32_t *d_n_rows, int64_t max_rows, int B, int H,
                        int W, int ds, const float *map_points, const float *map_normals, int Nmax,
                        int cap, float *tgt, float *tgt_normals, int32_t *counts, float *scan_points,
                        int32_t *scan_orig, int32_t *pix_start, int32_t *tgt_index, int32_t *tgt_pix,
                        void *ws, size_t ws_bytes, gs_stream_t stream);

/* keep mask of downsample_pointclouds' row filter for an arbitrary table
 * (odometry/icputils.py:596-597): mask[i] = rows[i].h % ds == 0 && rows[i].w % ds == 0 */
int gs_table_ds_mask(const int64_t *rows, int64_t n_rows, int ds, uint8_t *mask, gs_stream_t stream);

/* ---------------------------------------------------------------- K: exact 1-nearest neighbour
 * Replaces chamferdist.knn_points(src, tgt) with K=1 (call site odometry/icputils.py:200-201):
 * for each source point the squared L2 distance ((dx^2+dy^2)+dz^2, fp32, no FMA) to, and index of,
 * the nearest target point; the lowest index wins ties.  ns / nt are read from device memory
 * (d_ns, d_nt) so the call needs no host round trip; max_ns / max_nt bound the launch.
 * best: packed (dist_bits << 32 | idx) per source point, the form the J kernels consume.
 * gs_knn1 prunes target chunks by an exact fp32 AABB lower bound (same result, far fewer pairs);
 * gs_knn1_bruteforce evaluates every pair (the verifier). */
size_t gs_knn1_ws_bytes(int max_nt);
int gs_knn1(const float *src, const int32_t *d_ns, int max_ns, const float *tgt,
            const int32_t *d_nt, int max_nt, uint64_t *best, void *ws, size_t ws_bytes,
            gs_stream_t stream);
int gs_knn1_bruteforce(const float *src, const int32_t *d_ns, int max_ns, const float *tgt,
                       const int32_t *d_nt, int max_nt, uint64_t *best, gs_stream_t stream);
/* unpack to the reference's output types: dist2 fp32 (ns), idx int64 (ns) */
int gs_knn1_unpack(const uint64_t *best, const int32_t *d_ns, int max_ns, float *dist2,
                   int64_t *idx, gs_stream_t stream);

/* ---------------------------------------------------------------- J: linearise + reduce
 * gauss_newton_solve's algebra (odometry/icputils.py:203-232) fused with the normal-equation
 * products of solve_linear_system (:85-87) and the error dot product (:340): from src, tgt,
 * tgt normals and the packed nearest neighbours, accumulate
 *    H = sum a a^T (6x6), g = sum a b (6), e = sum b^2, cnt = #rows,
 * a = [n ; s x n], b = n.(d - s), over rows with dist2 < dist_thresh (dist_thresh < 0: all rows;
 * NB the reference compares the threshold with the SQUARED distance).
 * out: 44 floats = H (36, row-major) | g (6) | e | cnt (as float).  Deterministic (fixed-order
 * two-level reduction, no float atomics). */
size_t gs_icp_linearize_ws_bytes(int max_ns);
int gs_icp_linearize(const float *src, const int32_t *d_ns, int max_ns, const float *tgt,
                     const float *tgt_normals, const uint64_t *best, float dist_thresh,
                     float *out44, void *ws, size_t ws_bytes, gs_stream_t stream);

/* Rows form of the same algebra for API parity with gauss_newton_solve: A (ns,6), b (ns), and a
 * keep mask (ns) u8; callers compact with gs_compact_rows. */
int gs_icp_rows(const float *src, const int32_t *d_ns, int max_ns, const float *tgt,
                const float *tgt_normals, const uint64_t *best, float dist_thresh, float *A,
                float *b, uint8_t *keep, gs_stream_t stream);

/* Adjoint of gs_icp_linearize: given d/dH (36), d/dg (6), d/de (1) in g_out43, accumulate
 * d/dsrc (ns,3) (plain stores) and d/dtgt, d/dnormals (nt,3) (atomic scatter-add; zero-init). */
int gs_icp_linearize_backward(const float *src, const int32_t *d_ns, int max_ns, const float *tgt,
                              const float *tgt_normals, const uint64_t *best, float dist_thresh,
                              const float *g_out43, float *g_src, float *g_tgt, float *g_normals,
                              gs_stream_t stream);

/* transform_pointcloud (geometry/geometryutils.py:780-792): out = R p + t, T is a DEVICE 4x4. */
int gs_transform_points(const float *pts, const int32_t *d_n, int max_n, const float *T,
                        float *out, gs_stream_t stream);

/* ---------------------------------------------------------------- X: whole ICP loops on device
 * point_to_plane_ICP (odometry/icputils.py:310-367): LM loop with the accept/reject decision kept
 * on the device.  src (ns,3), tgt/normals (nt,3), init_T (device 4x4; NULL = identity).  Outputs: T (device 4x4),
 * optional best_last (packed NN of the last iteration's first solve) and optional trace
 * (numiters x 48 floats: H36|g6|err|new_err|damp|accept|cnt|pad).  dist_thresh < 0 == None. */
/* Optional search hints (NULL, or any member NULL, = none).  They never change a result: the
 * association stays the exact nearest neighbour with the reference's tie-break, indices are reported in
 * the reference order of `tgt`. */
typedef struct gs_icp_hints {
    const float *scan_points;   /* (nt,3) the target points in a spatially coherent scan order */
    const int32_t *scan_orig;   /* (nt) reference index (into tgt) of every scan slot; required with scan_points */
    const int32_t *src_pix;     /* (ns) ds-grid pixel id r*grid_w+c of every source point */
    const int32_t *pix_start;   /* (grid_h*grid_w+1) first scan slot of every pixel (scan order = pixel order) */
    const int32_t *tgt_pix;     /* (nt) ds-grid pixel of every target, reference order (optional; unused since ABI 3) */
    int32_t grid_w, grid_h;
    /* ABI 3: the camera the targets were bucketed with -- pose (4x4, camera -> world) and intrinsics (4x4) on the
     * device, and the grid step: a target sits in pixel (r, c) iff its projection under (cam_pose, cam_K), rounded
     * half-to-even as find_active_map_points does (slam/fusionutils.py:247-282), is the image pixel (r ds, c ds).
     * gs_build_icp_target / gs_bucket_by_pixel produce exactly that from the pose and intrinsics they are given. */
    const float *cam_pose, *cam_K;
    int32_t ds;
    /* With ALL of the above given (tgt_pix excepted) the loops associate by GRID SEARCH WITH A GEOMETRIC PROOF
     * (gs_set_grid_search): every source point examines all targets of the 3x3 grid pixels around the pixel it projects
     * to; every other target projects at least 2 ds - 0.5 image pixels away from that pixel's centre, i.e. lies outside a
     * pyramid through the camera centre, and the point's distance to the pyramid's faces bounds its distance to all of
     * them from below -- a window best strictly inside that bound IS the nearest neighbour.  Points whose proof fails
     * (no map point within centimetres) take the exact chunk-box search.  Hints never change a result (tested against
     * the brute-force scan with consistent and with scrambled hints). */
} gs_icp_hints;

/* point_to_plane_ICP (odometry/icputils.py:310-367) as one call on the device (section comment above). */
size_t gs_icp_ws_bytes(int max_ns, int max_nt);
int gs_icp_point_to_plane(const float *src, const int32_t *d_ns, int max_ns, const float *tgt,
                          const float *tgt_normals, const int32_t *d_nt, int max_nt,
                          const float *init_T, int numiters, float damp, float dist_thresh,
                          const gs_icp_hints *hints, float *out_T, uint64_t *best_last,
                          float *trace, void *ws, size_t ws_bytes, gs_stream_t stream);

/* point_to_plane_gradICP (odometry/icputils.py:479-545): the smooth gradLM variant. */
int gs_icp_point_to_plane_grad(const float *src, const int32_t *d_ns, int max_ns, const float *tgt,
                               const float *tgt_normals, const int32_t *d_nt, int max_nt,
                               const float *init_T, int numiters, float damp, float dist_thresh,
                               float lambda_max, float B, float B2, float nu,
                               const gs_icp_hints *hints, float *out_T, uint64_t *best_last,
                               float *trace, void *ws, size_t ws_bytes, gs_stream_t stream);

/* ---------------------------------------------------------------- X with autograd: taped loops + reverse pass
 * The differentiable form of point_to_plane_ICP / point_to_plane_gradICP (odometry/icputils.py:310-367,
 * :479-545; gradients as torch autograd derives them for the reference: through the rigid transforms,
 * the linearisation, the damped solve, se3_exp and -- gradLM only -- the damping / step gates; the
 * association indices and the LM accept/reject decisions are constants).
 * Forward: same loop and results as gs_icp_point_to_plane[_grad], but every association launch keeps its
 * cloud and neighbour array, and every step its state, in the caller's `tape` (gs_icp_tape_bytes).
 * Backward: given grad_T (device 4x4, adjoint of out_T) walks the tape in reverse on the device with no
 * host synchronisation and writes grad_src (max_ns,3), grad_tgt / grad_normals (max_nt,3, rows < *d_nt;
 * optional, NULL to skip) and grad_init_T (4x4).  grad_lm selects the gradLM variant (0: LM, parameters ignored). */
size_t gs_icp_tape_bytes(int max_ns, int numiters, int grad_lm);
int gs_icp_point_to_plane_taped(const float *src, const int32_t *d_ns, int max_ns, const float *tgt,
                                const float *tgt_normals, const int32_t *d_nt, int max_nt,
                                const float *init_T, int numiters, float damp, float dist_thresh,
                                int grad_lm, float lambda_max, float B, float B2, float nu,
                                const gs_icp_hints *hints, float *out_T, uint64_t *best_last,
                                void *tape, size_t tape_bytes, void *ws, size_t ws_bytes,
                                gs_stream_t stream);
size_t gs_icp_backward_ws_bytes(int max_ns);
int gs_icp_point_to_plane_backward(const float *src, const int32_t *d_ns, int max_ns, const float *tgt,
                                   const float *tgt_normals, const int32_t *d_nt, int max_nt, const float *init_T,
                                   int numiters, float dist_thresh, int grad_lm, float lambda_max,
                                   float B, float B2, float nu, const void *tape, size_t tape_bytes,
                                   const float *grad_T, float *grad_src, float *grad_tgt,
                                   float *grad_normals, float *grad_init_T, void *ws, size_t ws_bytes,
                                   gs_stream_t stream);

/* ---------------------------------------------------------------- whole PointFusion map update
 * update_map_fusion(pointclouds, live_frame, dist_th, dot_th, sigma, inplace=True)
 * (slam/fusionutils.py:761-789 = find_correspondences :549-577 + fuse_with_map :580-722) as ONE call with no
 * host synchronisation, on a map kept in caller-owned arena arrays: map_* are (B, Nmax, C) with the rows
 * n >= map_counts[b] zero; matched points are merged in place, the unmatched valid pixels of the live frame are
 * appended behind them in (h, w) order and map_counts advances ON THE DEVICE.  Nmax is both the row stride
 * and the capacity: the caller guarantees map_counts[b] + H*W <= Nmax (stats[2] flags a violation; rows that
 * do not fit are dropped).  depth (B,H,W), rgb (B,H,W,3), intrinsics / poses (B,16; poses = the live frame's
 * pose).  stats (optional, 4 + B int32): active rows, unique correspondences, overflow flag, max normal
 * dot product (float bits), appended rows per batch element -- what the reference's warnings are raised from. */
size_t gs_pointfusion_update_ws_bytes(int B, int H, int W, int Nmax);
int gs_pointfusion_update(const float *depth, const float *rgb, const float *intrinsics, const float *poses,
                          int B, int H, int W, float *map_points, float *map_normals, float *map_colors,
                          float *map_ccounts, int32_t *map_counts, int Nmax, float dist_th, float dot_th,
                          float sigma, int32_t *stats, void *ws, size_t ws_bytes, gs_stream_t stream);

/* update_map_fusion with gradients, as ONE node per sequence (reference: torch autograd through
 * slam/fusionutils.py:654-720 and structures/pointclouds.py:1203-1235, frame after frame).
 * Forward = gs_pointfusion_update (same arguments, same in-place result) that also fills `tape`
 * (gs_pointfusion_update_tape_bytes): per pixel the map point it merged into, the ten attribute floats that point held
 * before the merge, and the row counts before the append.
 * Backward (frames in reverse order; G_* = running adjoint of the WHOLE map, (B,Nmax,C) like the map arrays, holding
 * the adjoint of the map AFTER this frame on entry and of the map BEFORE it on return): pulls G back through the merge
 * at the matched rows only (an unmatched point passes its adjoint through: x' = (c x + 0)/c), reads the appended rows'
 * adjoints out behind the previous count, writes the frame's adjoints g_vertex (local vertex map, through alpha),
 * g_gvertex, g_gnormal, g_rgb (B,H,W,3 each, fully written) and RESTORES map_* / map_counts to the previous frame's,
 * so that the localisation's reverse pass of the same frame finds the map it ran against.  O(pixels) per frame,
 * no host synchronisation. */
size_t gs_pointfusion_update_tape_bytes(int B, int H, int W);
int gs_pointfusion_update_taped(const float *depth, const float *rgb, const float *intrinsics, const float *poses,
                                int B, int H, int W, float *map_points, float *map_normals, float *map_colors,
                                float *map_ccounts, int32_t *map_counts, int Nmax, float dist_th, float dot_th,
                                float sigma, int32_t *stats, void *tape, size_t tape_bytes, void *ws,
                                size_t ws_bytes, gs_stream_t stream);
size_t gs_pointfusion_update_backward_ws_bytes(int B, int H, int W);
int gs_pointfusion_update_backward(const float *depth, const float *rgb, const float *intrinsics, const float *poses,
                                   int B, int H, int W, float *map_points, float *map_normals, float *map_colors,
                                   float *map_ccounts, int32_t *map_counts, int Nmax, float sigma, const void *tape,
                                   size_t tape_bytes, float *G_points, float *G_normals, float *G_colors,
                                   float *G_ccounts, float *g_vertex, float *g_gvertex, float *g_gnormal,
                                   float *g_rgb, void *ws, size_t ws_bytes, gs_stream_t stream);

/* The same for ICPSLAM's aggregate map (update_map_aggregate, slam/fusionutils.py:725-758 with inplace=True): the
 * global vertices, normals and colours of every valid live-frame pixel are appended, unmerged, in (h, w) order
 * behind the rows the arena holds; map_counts advances on the device.  stats (optional, 4 + B int32): [2] = overflow
 * flag (set in all of 0..3 when rows had to be dropped), [4 + b] = rows appended. */
size_t gs_aggregate_update_ws_bytes(int B, int H, int W);
int gs_aggregate_update(const float *depth, const float *rgb, const float *intrinsics, const float *poses, int B,
                        int H, int W, float *map_points, float *map_normals, float *map_colors,
                        int32_t *map_counts, int Nmax, int32_t *stats, void *ws, size_t ws_bytes,
                        gs_stream_t stream);

/* ---------------------------------------------------------------- differentiable localisation step
 * gs_slam_localize with autograd (the same stages; gvertex = the live frame's global vertex map under the
 * PREVIOUS pose is an input here, so that its own adjoint chains into gs_vertex_normal_maps_backward).
 * Gradients flow to gvertex (the ICP source cloud), to the map points / normals that were ICP targets, and
 * to prev_poses through the final composition -- exactly the paths torch autograd follows in the reference
 * (icpslam.py:238-247): projection / ds-grid selection / association indices are constants.
 * Backward outputs are dense and fully written: grad_gvertex (B,H,W,3), grad_map_points / grad_map_normals
 * (B,Nmax,3; optional), grad_prev_poses (B,16). */
size_t gs_slam_localize_tape_bytes(int B, int H, int W, int ds, int Nmax, int numiters, int use_grad_lm);
int gs_slam_localize_taped(const float *depth, const float *gvertex, const float *intrinsics,
                           const float *prev_poses, int B, int H, int W, int ds, const float *map_points,
                           const float *map_normals, const int32_t *map_counts, int Nmax, int use_grad_lm,
                           int numiters, float damp, float dist_thresh, float lambda_max, float Bp,
                           float B2, float nu, float *out_poses, void *tape, size_t tape_bytes, void *ws,
                           size_t ws_bytes, gs_stream_t stream);
size_t gs_slam_localize_backward_ws_bytes(int B, int H, int W, int ds, int Nmax);
int gs_slam_localize_backward(const float *prev_poses, int B, int H, int W, int ds, const float *map_points,
                              const float *map_normals, int Nmax, int use_grad_lm, int numiters,
                              float dist_thresh, float lambda_max, float Bp, float B2, float nu,
                              const void *tape, size_t tape_bytes, const float *grad_out_poses,
                              float *grad_gvertex, float *grad_map_points, float *grad_map_normals,
                              float *grad_prev_poses, int accumulate_map_grads /* 1: add into grad_map_* (a running
                              adjoint of the whole map) instead of overwriting them */,
                              void *ws, size_t ws_bytes, gs_stream_t stream);

/* gs_slam_localize can replay its ICP loops as a cached hipGraph once a configuration repeats (all loop
 * arguments live in the caller's workspace).  mode: 1 on, 0 off (eager launches), -1 automatic: the library
 * times its own eager launches on the host and switches to graph replay only on hosts where a launch costs
 * more than ~8 us (environment: GS_NO_GRAPH=1 / GS_GRAPH=1 force either).  Results are identical either way. */
void gs_set_graph_mode(int mode);
/* Diagnostics of that policy: out4 = {eager enqueues timed, their minimum host cost per launch in us, graphs
 * captured, graph replays}. */
int gs_graph_stats(double *out4);
/* out = T . P for B pairs of 4x4 (compose_transformations as slam/icpslam.py:245-247 uses it). */
int gs_compose_poses(const float *T, const float *P, int B, float *out, gs_stream_t stream);
/* ---------------------------------------------------------------- whole localisation step
 * ICPSLAM._localize for odom in {icp, gradicp} (slam/icpslam.py:238-247) as ONE call with no host
 * synchronisation: live-frame maps posed with the previous pose (rgbdimages.py:643-762), ds-grid
 * source cloud (icputils.py:651-669), active map points on the ds-grid of the previous frame
 * (fusionutils.py:247-282 + icputils.py:596-619), the (grad)ICP loop, and the pose composition
 * T . prev_pose (kornia compose_transformations semantics).  depth (B,H,W) is ONE frame per batch
 * element; prev_poses / out_poses are B x 16.  vertex / normal / gnormal (B,H,W,3) are optional outputs
 * (NULL to skip), gvertex is required scratch/output.  use_grad_lm selects the gradLM variant. */
size_t gs_slam_localize_ws_bytes(int B, int H, int W, int ds, int Nmax);
int gs_slam_localize(const float *depth, const float *intrinsics, const float *prev_poses, int B,
                     int H, int W, int ds, const float *map_points, const float *map_normals,
                     const int32_t *map_counts, int Nmax, int use_grad_lm, int numiters, float damp,
                     float dist_thresh, float lambda_max, float Bp, float B2, float nu,
                     float *vertex, float *normal, float *gvertex, float *gnormal, float *out_poses,
                     void *ws, size_t ws_bytes, gs_stream_t stream);

/* Optional timing of the two hot kernels of the loops above with HIP events recorded on the launch
 * stream (used by bench.py for the roofline line; off by default).  gs_profile_read folds the events
 * recorded so far (caller synchronises first) and returns launches / total ms for tag 0 = association
 * kernel, 1 = linearise kernel. */
void gs_profile_enable(int on);
int gs_profile_read(int tag, long *launches, double *total_ms);

/* How the loops above associate when ALL gs_icp_hints (camera included) are given.  1 (default; 2 is accepted as the
 * same) = grid search with its geometric proof (see gs_icp_hints), at every target density; 0 = chunk-box search always.
 * Same results bit for bit (both are the brute-force scan's); only the cost differs.  Replaces nothing in the
 * reference (chamferdist.knn_points has no such switch); for measurements and tests. */
void gs_set_grid_search(int on);

/* Source points per 1024-thread block of the loops' association kernel: 0 (default) = 64; 32 .. 64 = that many (tests:
 * the tile size fixes the summation order of the 6x6 system, so results of different settings agree to rounding, not bit
 * for bit; nearest neighbours are the brute-force scan's under every setting).  Replaces nothing in the reference. */
void gs_set_tile_points(int n);
/* Launch geometry of one association launch of the loops above for a source capacity (host-side query, no device
 * work): blocks launched, source points per block, and the number of partial rows the workspace (gs_icp_ws_bytes)
 * holds per buffer (>= blocks for every tile-size setting).  The rows are padded to at least 513: the association
 * kernel's prologue sums 512 rows unmasked (rows no block writes are kept at zero) and reads up to twelve bytes past a
 * row.  have_hints is ignored since ABI 3. */
int gs_icp_launch_geometry(int max_ns, int have_hints, int *blocks, int *tile_points_dense, int *partial_rows);
/* Counters kept on the DEVICE for the loops run so far: out4 = {loops, loops associated by grid search, loops with a
 * forced tile size (gs_set_tile_points), tiles whose point-serial straggler search overflowed its pair list and was
 * redone by the tile-level search}.  Synchronises with the device; reset != 0 zeroes the counters afterwards.  Replaces
 * nothing in the reference; lets a test assert which paths a run really exercised. */
int gs_loop_counts(unsigned int *out4, int reset);

/* ---------------------------------------------------------------- C+U: fusion correspondences
 * find_similar_map_points (slam/fusionutils.py:381-401): keep[i] = |Vg(b,h,w) - p(b,n)| < dist_th
 * (Euclidean) && Ng(b,h,w).nrm(b,n) > dot_th for every table row; max_dot (device float, may be
 * NULL) receives the maximum dot product when that exceeds 1 and 0 otherwise (the reference warns when
 * it exceeds 1.001: un-normalised normals). */
int gs_fusion_similar(const int64_t *rows, const int32_t *d_n_rows, int64_t max_rows,
                      const float *gvertex, const float *gnormal, int H, int W,
                      const float *map_points, const float *map_normals, int Nmax, float dist_th,
                      float dot_th, uint8_t *keep, float *max_dot, gs_stream_t stream);

/* find_best_unique_correspondences (slam/fusionutils.py:489-546): per (b,h,w) keep the row
 * minimising (1/(ccount+1e-20), squared ray distance, n) compared as fp32; output rows [b,n,h,w]
 * sorted by (b,h,w).  `keep` (may be NULL = all) pre-filters rows, which fuses the compaction of
 * find_similar_map_points.  Replaces the reference's torch.unique(dim=0) row sort. */
size_t gs_fusion_unique_ws_bytes(int B, int H, int W);
int gs_fusion_unique(const int64_t *rows, const uint8_t *keep, const int32_t *d_n_rows,
                     int64_t max_rows, const float *gvertex, int B, int H, int W,
                     const float *map_points, const float *map_ccounts, int Nmax,
                     int64_t *out_rows, int32_t *out_count, void *ws, size_t ws_bytes,
                     gs_stream_t stream);

/* ---------------------------------------------------------------- F: merge matched points
 * fuse_with_map's weighted running average (slam/fusionutils.py:654-699).  EVERY map point goes
 * through the reference's formula  c' = c + alpha, x' = (c x + alpha x_f) * (1 / (c'==0 ? 1 : c'))
 * for points, normals and colours, with alpha = x_f = 0 for points that are not in `rows` (the
 * reference evaluates it on the whole padded tensors, which perturbs unmatched points by rounding;
 * that is part of its observable result).  Reads the in_* arrays (B,Nmax,C), writes the out_*
 * arrays (may alias in_* for an in-place update).  alpha (B,H,W) = get_alpha(local vertex map). */
size_t gs_fusion_merge_ws_bytes(int B, int Nmax);
int gs_fusion_merge(const int64_t *rows, const int32_t *d_n_rows, int64_t max_rows,
                    const float *gvertex, const float *gnormal, const float *rgb,
                    const float *alpha, int B, int H, int W, int Nmax, const int32_t *counts,
                    const float *in_points, const float *in_normals, const float *in_colors,
                    const float *in_ccounts, float *out_points, float *out_normals,
                    float *out_colors, float *out_ccounts, void *ws, size_t ws_bytes,
                    gs_stream_t stream);
/* Adjoint.  g_in_* (B,Nmax,C) are overwritten; g_gvertex/g_gnormal/g_rgb (B,H,W,3) and g_alpha
 * (B,H,W) receive plain stores at matched pixels only (zero-init by the caller).  Any g_* may be
 * NULL. */
int gs_fusion_merge_backward(const int64_t *rows, const int32_t *d_n_rows, int64_t max_rows,
                             const float *gvertex, const float *gnormal, const float *rgb,
                             const float *alpha, int B, int H, int W, int Nmax,
                             const int32_t *counts, const float *in_points,
                             const float *in_normals, const float *in_colors,
                             const float *in_ccounts, const float *g_out_points,
                             const float *g_out_normals, const float *g_out_colors,
                             const float *g_out_ccounts, float *g_in_points, float *g_in_normals,
                             float *g_in_colors, float *g_in_ccounts, float *g_gvertex,
                             float *g_gnormal, float *g_rgb, float *g_alpha, void *ws,
                             size_t ws_bytes, gs_stream_t stream);

/* In-place form of gs_fusion_merge for an arena-backed map: the rows that exist (n < counts[b]) are merged
 * where they are, padding is not touched, and nothing happens at all when *d_n_rows == 0 (fuse_with_map
 * skips the merge when there is no correspondence, slam/fusionutils.py:654). */
size_t gs_fusion_merge_inplace_ws_bytes(int B, int Nmax);
int gs_fusion_merge_inplace(const int64_t *rows, const int32_t *d_n_rows, int64_t max_rows,
                            const float *gvertex, const float *gnormal, const float *rgb,
                            const float *alpha, int B, int H, int W, int Nmax, const int32_t *counts,
                            float *points, float *normals, float *colors, float *ccounts, void *ws,
                            size_t ws_bytes, gs_stream_t stream);

/* ---------------------------------------------------------------- A: new-point mask
 * fuse_with_map's append mask (slam/fusionutils.py:702-707): mask = valid_depth && pixel not in
 * `rows` (B,H,W) u8.  Callers then compact gvertex/gnormal/rgb/alpha with gs_compact_rows per b. */
int gs_fusion_new_mask(const float *depth, const int64_t *rows, const int32_t *d_n_rows,
                       int64_t max_rows, int B, int H, int W, uint8_t *mask, gs_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* GRADSLAM_HIP_H */
