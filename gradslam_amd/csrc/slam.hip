// slam.hip -- whole-step drivers: chains of the kernels of this library with every data-dependent size
// kept on the device, so one C call enqueues a complete ICPSLAM._localize (reference
// slam/icpslam.py:238-247) without a single host round trip.
#include <chrono>
#include <mutex>
#include <stdlib.h>
#include <vector>

#include "gs_common.hpp"
#include "gs_project.hpp"

namespace gs {

bool profiling_enabled();  // icp.hip

// ------------------------------------------------------------------ hipGraph cache for the ICP loops
// Every argument of the (grad)ICP loop launched by gs_slam_localize lives in the caller's workspace, so
// for a given workspace address and configuration the ~35 launches are the same bytes every frame:
// capture them once (on a private stream: the user's may be the legacy default stream, which cannot be
// captured) and replay the instantiated graph on the user's stream.  Replay costs one launch on the host
// instead of ~35, which is what keeps the step GPU-bound on slow hosts.  GS_NO_GRAPH=1 disables it.
struct GraphKey {
    void *ws;
    int B, capS, capT, numiters, use_grad;
    float damp, thresh, lmax, Bp, B2, nu;
    int icp_cfg;             // what else decides WHICH kernels a loop launches: gs_set_grid_search / gs_set_tile_points
    int H, W, ds;            // the loop's constants hold the ds-grid's dimensions
    bool operator==(const GraphKey &o) const {
        return H == o.H && W == o.W && ds == o.ds && ws == o.ws && B == o.B && capS == o.capS && capT == o.capT && numiters == o.numiters && use_grad == o.use_grad &&
               damp == o.damp && thresh == o.thresh && lmax == o.lmax && Bp == o.Bp && B2 == o.B2 && nu == o.nu &&
               icp_cfg == o.icp_cfg;
    }
};
struct GraphEntry {
    GraphKey key;
    int device;
    hipGraphExec_t exec;
    unsigned long long last_use;
};
static std::mutex g_graph_mu;
static std::vector<GraphEntry> g_graphs;
static std::vector<GraphKey> g_seen_once;  // a configuration is captured the second time it shows up
static unsigned long long g_graph_clock = 0;
static int g_graph_mode = -1;  // -1: automatic, 0: off, 1: on
// Automatic mode.  With the step folded into the association a loop is 13 launches, and on a fast host
// launching them eagerly is as fast as replaying a graph (and lets the loop's last launch write the composed
// pose); on a slow host the ~2x lower host cost of a replay is what keeps the step GPU-bound.  So the library
// times its own eager enqueues (host clock around the loop's launches, minimum over the first calls -- the
// minimum, because a full queue makes a launch block) and turns graphs on only if a launch costs the host
// more than kSlowLaunchUs -- about where the ~24 launches of a step would take the host as long as they take
// the GPU (a fast host needs ~3 us per launch).  Capturing costs several milliseconds once, so a borderline
// host is better off eager.  GS_NO_GRAPH=1 / GS_GRAPH=1 in the environment, or gs_set_graph_mode, override.
constexpr double kSlowLaunchUs = 8.0;
static int g_auto_samples = 0;
static double g_auto_min_us = 1e30;
static int g_captures = 0, g_replays = 0;
static bool graphs_allowed() {
    static const bool env_off = getenv("GS_NO_GRAPH") != nullptr, env_on = getenv("GS_GRAPH") != nullptr;
    if (profiling_enabled()) return false;
    if (g_graph_mode >= 0) return g_graph_mode != 0;
    if (env_off) return false;
    if (env_on) return true;
    return g_auto_samples >= 4 && g_auto_min_us > kSlowLaunchUs;
}

// out[b] = T[b] . P[b]   (compose44, gs_common.hpp)
__global__ void compose_k(const float *__restrict__ T, const float *__restrict__ P, int B, float *__restrict__ out) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    compose44(T + 16 * b, P + 16 * b, out + 16 * b);
}

// icp.hip: gs_icp_point_to_plane[_grad|_taped] with the pose composition folded into the loop's last launch
// (compose_out = out_T . compose_right; both optional)
int icp_localize_run(int grad_lm, const float *src, const int32_t *d_ns, int max_ns, const float *tgt, const float *nrm,
                     const int32_t *d_nt, int max_nt, int numiters, float damp, float thresh, float lambda_max, float Bp,
                     float B2, float nu, const gs_icp_hints *hints, float *out_T, void *ws, size_t ws_bytes, hipStream_t st,
                     void *tape, size_t tape_bytes, const float *compose_right, float *compose_out);
int icp_config_stamp();  // icp.hip: the process-wide search / tiling switches, as one number
// maps.hip: the maps of one frame per batch element + (pose | intrinsics) copied to cam_out (B, 32)
int vertex_normal_maps_cam(const float *depth, const float *intrinsics, const float *poses, int B, int H, int W, float *vertex,
                           float *normal, float *gvertex, float *gnormal, float *cam_out, hipStream_t st);
__global__ void eye4_k(float *__restrict__ T, int B) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < 16 * B) T[i] = ((i % 16) % 5 == 0) ? 1.0f : 0.0f;
}

// project.hip: active-point projection on the ds grid + ICP target build for one sequence, 4 launches
size_t project_target1_ws_bytes(int H, int W, int ds, int Nmax);
int project_target1(const float *points, const int32_t *counts, int Nmax, const float *poses, const float *intrinsics, int H,
                    int W, int ds, const float *map_normals, int cap, int64_t *rows, int32_t *nrows, float *tgt, float *tnrm,
                    int32_t *nt, float *scan_points, int32_t *scan_orig, int32_t *pix_start, int32_t *tgt_index, int32_t *tgt_pix,
                    void *ws, size_t ws_bytes, hipStream_t st, const DsJob *frame);

struct LocWs {
    float *src;        // (B, capS, 3)
    int32_t *src_pix;  // (B, capS) ds-grid pixel of every source point
    float *scan;       // (B, capT, 3) target points in pixel order
    int32_t *scan_orig;// (B, capT)
    int32_t *pix_start;// (B, capS + 1) first scan slot of every ds-grid pixel
    int32_t *ns;       // (B)
    int64_t *rows;     // (B*Nmax, 4)
    int32_t *nrows;    // (1)
    float *tgt, *tnrm; // (B, capT, 3)
    int32_t *nt;       // (B)
    float *T;          // (B, 16) ICP result; eye (B,16) follows
    float *eye;
    float *cam;        // (B, 32) previous pose | intrinsics: the bucketing camera at a workspace address (graph replay)
    void *sub;         // scratch shared by the sub-calls (they run one after the other on one stream)
    size_t sub_bytes;
};

// target capacity: the map size rounded up to a power of two, so that the workspace layout (and with it the
// captured graph of the ICP loops) stays the same while a map grows frame by frame
static inline int target_cap(int Nmax) {
    int c = 1024;
    while (c < Nmax && c < (1 << 30)) c <<= 1;
    return c;
}

static size_t loc_layout(int B, int H, int W, int ds, int Nmax, void *ws, LocWs *out) {
    const int capS = cdiv(H, ds) * cdiv(W, ds), capT = target_cap(Nmax);
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += align_up(bytes, 256); return o; };
    const size_t o_src = take((size_t)B * capS * 12), o_ns = take((size_t)B * 4);
    const size_t o_spix = take((size_t)B * capS * 4), o_scan = take((size_t)B * capT * 12);
    const size_t o_sorig = take((size_t)B * capT * 4), o_pseed = take((size_t)B * (capS + 1) * 4);
    // NB every size below depends on (B, H, W, ds, capT) only -- never on Nmax itself -- so that pointers baked
    // into a captured graph stay valid while the map grows inside one capacity bucket
    const size_t o_rows = take((size_t)B * capT * 32), o_nrows = take(4);
    const size_t o_tgt = take((size_t)B * capT * 12), o_tnrm = take((size_t)B * capT * 12), o_nt = take((size_t)B * 4);
    const size_t o_T = take((size_t)B * 64), o_eye = take((size_t)B * 64), o_cam = take((size_t)B * 128);
    size_t sub = gs_downsample_frame_ws_bytes(H, W, ds);
    sub = std::max(sub, gs_project_active_ws_bytes(B, Nmax));
    sub = std::max(sub, gs_gather_table_rows_ws_bytes(B));
    sub = std::max(sub, gs_bucket_by_pixel_ws_bytes(B, H, W, ds));
    sub = std::max(sub, gs_icp_ws_bytes(capS, capT));
    sub = std::max(sub, project_target1_ws_bytes(H, W, ds, Nmax));
    const size_t o_sub = take(sub);
    if (ws && out) {
        char *p = (char *)ws;
        out->src = (float *)(p + o_src); out->ns = (int32_t *)(p + o_ns);
        out->src_pix = (int32_t *)(p + o_spix); out->scan = (float *)(p + o_scan);
        out->scan_orig = (int32_t *)(p + o_sorig); out->pix_start = (int32_t *)(p + o_pseed);
        out->rows = (int64_t *)(p + o_rows); out->nrows = (int32_t *)(p + o_nrows);
        out->tgt = (float *)(p + o_tgt); out->tnrm = (float *)(p + o_tnrm); out->nt = (int32_t *)(p + o_nt);
        out->T = (float *)(p + o_T); out->eye = (float *)(p + o_eye); out->cam = (float *)(p + o_cam);
        out->sub = p + o_sub; out->sub_bytes = sub;
    }
    return off;
}

// fusion.hip: the fused correspondence chain of the PointFusion update (no tables: 4 bytes per map point), the merge that
// reads it, the append of the unmatched pixels and the update's last launch; maps.hip: the maps kernel with its riders
size_t fusion_corr_state_bytes(int B, int H, int W, int Nmax);
void fusion_corr_init_ptrs(void *state, int B, int H, int W, int Nmax, unsigned long long **pix_key, unsigned int **pix_n);
int fusion_correspond(void *state, const float *map_points, const float *map_normals, const float *map_ccounts, const int32_t *counts,
                      int B, int Nmax, const float *poses, const float *intrinsics, int H, int W, const float *gvertex,
                      const float *gnormal, float dist_th, float dot_th, int32_t *ctr, hipStream_t st);
int fusion_merge_corr(void *state, const int32_t *ctr, const float *gvertex, const float *gnormal, const float *rgb, const float *alpha,
                      int B, int H, int W, int Nmax, const int32_t *counts, float *points, float *normals, float *colors,
                      float *ccounts, hipStream_t st);
int fusion_append_corr(void *state, int B, int H, int W, int Nmax, int b, const float *depth, const float *const *h_src,
                       const int *h_row_floats, float *const *h_dst, const int32_t *d_count, int cap, int *d_total, void *cws,
                       hipStream_t st);
int fusion_finish(void *state, int B, int H, int W, int Nmax, int32_t *ctr, int32_t *counts, const int *totals, int cap,
                  int32_t *appended, int32_t *stats, hipStream_t st);
int vertex_normal_maps_fusion(const float *depth, const float *intrinsics, const float *poses, int B, int H, int W, float *gvertex,
                              float *gnormal, float *alpha, float sigma, float eps, unsigned long long *pix_key, unsigned int *pix_n,
                              int32_t *zero, int n_zero, hipStream_t st);

size_t fusion_tape_bytes(int B, int H, int W);
int fusion_tape_record(const void *state, void *tape, int B, int H, int W, int Nmax, const int32_t *counts, const float *points,
                       const float *normals, const float *colors, const float *ccounts, hipStream_t st);
int fusion_tape_appended(void *tape, int B, int H, int W, const int32_t *appended, hipStream_t st);
int fusion_update_reverse(const void *tape, int B, int H, int W, int Nmax, const float *depth, const float *gvertex,
                          const float *gnormal, const float *rgb, const float *alpha, float *points, float *normals, float *colors,
                          float *ccounts, int32_t *counts, float *Gp, float *Gn, float *Gc, float *Gcc, float *g_gvertex,
                          float *g_gnormal, float *g_rgb, float *g_alpha, void *cws, hipStream_t st);

int append_valid_pixels(int n_arrays, const float *depth_b, int64_t HW, const float *const *h_src, const int *h_row_floats,
                        float *const *h_dst, int32_t *d_count, int cap, int32_t *d_appended, int32_t *d_overflow, void *cws,
                        hipStream_t st);

// ------------------------------------------------------------------ PointFusion map update on an arena
struct FuseWs {
    float *gV, *gN, *alpha;          // (B,H,W,3) x2, (B,H,W)
    void *state;                     // correspondence stage: per-pixel keys / winners, per-point pixel, per-block counts
    int32_t *ctr;                    // counter block: active, unique, overflow, max_dot bits, any-similar flag
    int32_t *appended;               // (B)
    int *totals;                     // (B) rows each sequence's append pass selected
    void *sub;
    size_t sub_bytes;
};
constexpr int kCtrWords = 64;  // counter block (zeroed by the maps kernel): ctr[0..], appended at +64, totals at +128
static size_t fuse_layout(int B, int H, int W, int Nmax, void *ws, FuseWs *out) {
    const size_t npix = (size_t)B * H * W;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += align_up(bytes, 256); return o; };
    const size_t ogV = take(npix * 12), ogN = take(npix * 12), oA = take(npix * 4);
    const size_t oM = take(fusion_corr_state_bytes(B, H, W, Nmax));
    const size_t oC = take((size_t)(kCtrWords + 2 * 64) * 4);
    const size_t sub = gs_compact_ws_bytes((int64_t)H * W) + 256;
    const size_t oS = take(sub);
    if (ws && out) {
        char *p = (char *)ws;
        out->gV = (float *)(p + ogV); out->gN = (float *)(p + ogN);
        out->alpha = (float *)(p + oA); out->state = p + oM;
        int32_t *c = (int32_t *)(p + oC);
        out->ctr = c; out->appended = c + kCtrWords; out->totals = (int *)(c + kCtrWords + 64);
        out->sub = p + oS; out->sub_bytes = sub;
    }
    return off;
}

__global__ void fuse_stats_k(const int32_t *__restrict__ nrows, const int32_t *__restrict__ ucnt, const int32_t *__restrict__ overflow,
                             const float *__restrict__ max_dot, const int32_t *__restrict__ appended, int B,
                             int32_t *__restrict__ stats) {
    if (threadIdx.x == 0) {
        stats[0] = *nrows; stats[1] = *ucnt; stats[2] = *overflow; stats[3] = __float_as_int(*max_dot);
    }
    if ((int)threadIdx.x < B) stats[4 + threadIdx.x] = appended[threadIdx.x];
}

// ------------------------------------------------------------------ differentiable localisation
// Tape of one gs_slam_localize_taped call: what the reverse pass cannot cheaply recompute.
struct LocTape {
    float *src;         // (B, capS, 3) ds-grid source cloud
    int32_t *src_pix;   // (B, capS)   ds-grid pixel of every source point
    int32_t *ns, *nt;   // (B)
    int32_t *tgt_index; // (B, capT)   map index of every target slot
    float *T;           // (B, 16)     ICP result (before composition with the previous pose)
    char *icp;          // B x icp_tape bytes
    size_t icp_bytes;
};
static size_t loc_tape_layout(int B, int H, int W, int ds, int Nmax, int numiters, int grad_lm, void *tape, LocTape *out) {
    const int capS = cdiv(H, ds) * cdiv(W, ds), capT = target_cap(Nmax);
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += align_up(bytes, 256); return o; };
    const size_t o_src = take((size_t)B * capS * 12), o_pix = take((size_t)B * capS * 4), o_ns = take((size_t)B * 4);
    const size_t o_nt = take((size_t)B * 4), o_idx = take((size_t)B * capT * 4), o_T = take((size_t)B * 64);
    const size_t icp_b = align_up(gs_icp_tape_bytes(capS, numiters, grad_lm), 256);
    const size_t o_icp = take((size_t)B * icp_b);
    if (tape && out) {
        char *p = (char *)tape;
        out->src = (float *)(p + o_src); out->src_pix = (int32_t *)(p + o_pix);
        out->ns = (int32_t *)(p + o_ns); out->nt = (int32_t *)(p + o_nt);
        out->tgt_index = (int32_t *)(p + o_idx); out->T = (float *)(p + o_T);
        out->icp = p + o_icp; out->icp_bytes = icp_b;
    }
    return off;
}

// adjoint of compose_k: out = T . P
__global__ void compose_bwd_k(const float *__restrict__ T, const float *__restrict__ P, const float *__restrict__ gO, int B,
                              float *__restrict__ gT, float *__restrict__ gP) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const float *a = T + 16 * b, *p = P + 16 * b, *g = gO + 16 * b;
    float *ga = gT + 16 * b, *gp = gP + 16 * b;
    for (int i = 0; i < 3; ++i) {
        for (int k = 0; k < 3; ++k) ga[4 * i + k] = ((g[4 * i] * p[4 * k] + g[4 * i + 1] * p[4 * k + 1]) + g[4 * i + 2] * p[4 * k + 2]) + g[4 * i + 3] * p[4 * k + 3];
        ga[4 * i + 3] = g[4 * i + 3];
    }
    for (int k = 0; k < 3; ++k)
        for (int j = 0; j < 4; ++j) gp[4 * k + j] = (a[k] * g[j] + a[4 + k] * g[4 + j]) + a[8 + k] * g[8 + j];
    for (int j = 0; j < 4; ++j) { ga[12 + j] = 0.0f; gp[12 + j] = 0.0f; }
}

// target / normals of every slot again from the map (what gs_build_icp_target gathered in the forward pass)
__global__ void regather_k(const int32_t *__restrict__ tgt_index, const int32_t *__restrict__ nt, int capT,
                           const float *__restrict__ map_points, const float *__restrict__ map_normals, int Nmax,
                           float *__restrict__ tgt, float *__restrict__ tnrm) {
    const int b = blockIdx.y, n = min(nt[b], capT);
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x) {
        const int64_t srcI = (int64_t)b * Nmax + tgt_index[(int64_t)b * capT + k], dst = (int64_t)b * capT + k;
        st3(tgt, dst, ld3(map_points, srcI));
        st3(tnrm, dst, ld3(map_normals, srcI));
    }
}

// adjoints back to where the clouds came from: source slot i -> its ds-grid pixel of the (zeroed) global vertex
// map adjoint; target slot k -> its map point (every map point is at most one slot: plain stores)
__global__ void scatter_grads_k(const float *__restrict__ g_src, const int32_t *__restrict__ src_pix, const int32_t *__restrict__ ns,
                                int capS, int Wd, int ds, int H, int W, float *__restrict__ g_gvertex /* this b */,
                                const float *__restrict__ g_tgt, const float *__restrict__ g_nrm,
                                const int32_t *__restrict__ tgt_index, const int32_t *__restrict__ nt, int capT,
                                float *__restrict__ g_map_points, float *__restrict__ g_map_normals /* this b */, int accumulate) {
    const int n_s = min(*ns, capS), n_t = min(*nt, capT);
    const int stride = gridDim.x * blockDim.x;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n_s; i += stride) {
        const int pix = src_pix[i];
        const int64_t at = (int64_t)(pix / Wd) * ds * W + (int64_t)(pix % Wd) * ds;
        st3(g_gvertex, at, ld3(g_src, i));
    }
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < n_t; k += stride) {
        const int64_t at = tgt_index[k];
        f3 gp = g_map_points ? ld3(g_tgt, k) : f3{0, 0, 0}, gq = g_map_normals ? ld3(g_nrm, k) : f3{0, 0, 0};
        if (accumulate) {  // a running adjoint of the whole map (one slot per map point: no atomics needed)
            if (g_map_points) { const f3 o = ld3(g_map_points, at); gp = f3{o.x + gp.x, o.y + gp.y, o.z + gp.z}; }
            if (g_map_normals) { const f3 o = ld3(g_map_normals, at); gq = f3{o.x + gq.x, o.y + gq.y, o.z + gq.z}; }
        }
        if (g_map_points) st3(g_map_points, at, gp);
        if (g_map_normals) st3(g_map_normals, at, gq);
    }
}

}  // namespace gs

using namespace gs;

extern "C" {

void gs_set_graph_mode(int mode) { g_graph_mode = mode; }

int gs_graph_stats(double *out4) {
    GS_REQUIRE(out4, "gs_graph_stats: NULL argument");
    std::lock_guard<std::mutex> lock(g_graph_mu);
    out4[0] = g_auto_samples; out4[1] = g_auto_min_us; out4[2] = g_captures; out4[3] = g_replays;
    return GS_OK;
}

int gs_compose_poses(const float *T, const float *P, int B, float *out, gs_stream_t stream) {
    GS_REQUIRE(T && P && out && B > 0, "gs_compose_poses: bad arguments");
    hipLaunchKernelGGL(compose_k, dim3(cdiv(B, 64)), dim3(64), 0, (hipStream_t)stream, T, P, B, out);
    GS_LAUNCH_CHECK("gs_compose_poses");
    return GS_OK;
}

size_t gs_slam_localize_ws_bytes(int B, int H, int W, int ds, int Nmax) {
    if (B <= 0 || H <= 0 || W <= 0 || ds <= 0 || Nmax <= 0) return 0;
    return loc_layout(B, H, W, ds, Nmax, nullptr, nullptr);
}

int gs_slam_localize(const float *depth, const float *intrinsics, const float *prev_poses, int B, int H, int W, int ds,
                     const float *map_points, const float *map_normals, const int32_t *map_counts, int Nmax,
                     int use_grad_lm, int numiters, float damp, float dist_thresh, float lambda_max, float Bp, float B2,
                     float nu, float *vertex, float *normal, float *gvertex, float *gnormal, float *out_poses, void *ws,
                     size_t ws_bytes, gs_stream_t stream) {
    GS_REQUIRE(depth && intrinsics && prev_poses && map_points && map_normals && map_counts && gvertex && out_poses,
               "gs_slam_localize: NULL argument (gvertex is required, the other maps are optional)");
    GS_REQUIRE(B > 0 && H >= 2 && W >= 2 && ds > 0 && Nmax > 0 && numiters >= 0, "gs_slam_localize: bad shape");
    if (!ws || ws_bytes < gs_slam_localize_ws_bytes(B, H, W, ds, Nmax)) {
        set_error("gs_slam_localize: workspace too small (%zu < %zu)", ws_bytes, gs_slam_localize_ws_bytes(B, H, W, ds, Nmax));
        return GS_ERR_WORKSPACE_TOO_SMALL;
    }
    hipStream_t st = (hipStream_t)stream;
    LocWs w;
    loc_layout(B, H, W, ds, Nmax, ws, &w);
    const int capS = cdiv(H, ds) * cdiv(W, ds), capT = target_cap(Nmax);
    int rc;
    // live frame posed with the previous pose: maps, then the ds-grid source cloud
    // (the maps kernel also leaves the bucketing camera -- previous pose and intrinsics -- in the workspace: the loops read it
    // from there, an address a captured graph may keep, never from the caller's tensors)
    if ((rc = vertex_normal_maps_cam(depth, intrinsics, prev_poses, B, H, W, vertex, normal, gvertex, gnormal, w.cam, st))) return rc;
    // map points that land on the ds-grid of the previous frame: the ICP target
    // reference-order target (points, normals, counts) + the same points in pixel order and the first scan
    // slot of every ds-grid pixel (search hints only); one sequence takes the 4-launch fused form, and the frame's ds-grid
    // source cloud rides on its first two launches (two launches of ~5 us less on the step's chain)
    if (B == 1) {
        const DsJob frame{depth, gvertex, w.src, w.src_pix, w.ns};
        if ((rc = project_target1(map_points, map_counts, Nmax, prev_poses, intrinsics, H, W, ds, map_normals, capT, w.rows, w.nrows,
                                  w.tgt, w.tnrm, w.nt, w.scan, w.scan_orig, w.pix_start, nullptr, nullptr, w.sub, w.sub_bytes, st, &frame))) return rc;
    } else {
        if ((rc = gs_downsample_frame(depth, gvertex, nullptr, nullptr, B, H, W, ds, capS, w.src, nullptr, nullptr, w.src_pix, w.ns,
                                      w.sub, w.sub_bytes, stream))) return rc;
        if ((rc = gs_project_active(map_points, map_counts, B, Nmax, prev_poses, intrinsics, H, W, ds, w.rows, w.nrows, w.sub,
                                    w.sub_bytes, stream))) return rc;
        if ((rc = gs_build_icp_target(w.rows, w.nrows, (int64_t)B * Nmax, B, H, W, ds, map_points, map_normals, Nmax, capT, w.tgt,
                                      w.tnrm, w.nt, w.scan, w.scan_orig, w.pix_start, nullptr, nullptr, w.sub, w.sub_bytes, stream))) return rc;
    }
    // fold_compose: the loop's last launch also writes out_poses = T . prev_poses.  Only for eager launches: a
    // captured graph must not bake the caller's prev_poses / out_poses addresses in (they change every call).
    auto enqueue_loops = [&](gs_stream_t s, bool fold_compose) -> int {
        for (int b = 0; b < B; ++b) {  // sequences are independent; one device-resident loop each
            const float *src = w.src + (size_t)b * capS * 3;
            const float *tgt = w.tgt + (size_t)b * capT * 3, *nrm = w.tnrm + (size_t)b * capT * 3;
            const gs_icp_hints hints{w.scan + (size_t)b * capT * 3, w.scan_orig + (size_t)b * capT, w.src_pix + (size_t)b * capS,
                                     w.pix_start + (size_t)b * (capS + 1), nullptr, cdiv(W, ds), cdiv(H, ds),
                                     w.cam + 32 * b, w.cam + 32 * b + 16, ds};
            // the loop's last launch also writes out_poses[b] = T . prev_poses[b]
            const int r = icp_localize_run(use_grad_lm, src, w.ns + b, capS, tgt, nrm, w.nt + b, capT, numiters, damp, dist_thresh,
                                           lambda_max, Bp, B2, nu, &hints, w.T + 16 * b, w.sub, w.sub_bytes, (hipStream_t)s, nullptr,
                                           0, fold_compose ? prev_poses + 16 * b : nullptr,
                                           fold_compose ? out_poses + 16 * b : nullptr);
            if (r) return r;
        }
        return GS_OK;
    };
    bool launched = false;
    if (graphs_allowed() && numiters > 0) {
        std::lock_guard<std::mutex> lock(g_graph_mu);
        int device = 0;
        (void)hipGetDevice(&device);
        const GraphKey key{ws, B, capS, capT, numiters, use_grad_lm, damp, dist_thresh, lambda_max, Bp, B2, nu,
                           icp_config_stamp(), H, W, ds};
        GraphEntry *hit = nullptr;
        for (auto &e : g_graphs)
            if (e.device == device && e.key == key) hit = &e;
        bool seen = false;
        if (!hit) {
            for (auto &k : g_seen_once) seen = seen || (k == key);
            if (!seen) {
                if (g_seen_once.size() >= 16) g_seen_once.erase(g_seen_once.begin());
                g_seen_once.push_back(key);
            }
        }
        if (!hit && seen) {
            hipStream_t cs = nullptr;
            hipGraph_t graph = nullptr;
            hipGraphExec_t exec = nullptr;
            bool ok = hipStreamCreateWithFlags(&cs, hipStreamNonBlocking) == hipSuccess;
            ok = ok && hipStreamBeginCapture(cs, hipStreamCaptureModeThreadLocal) == hipSuccess;
            if (ok) {
                const int r = enqueue_loops((gs_stream_t)cs, false);
                const hipError_t e = hipStreamEndCapture(cs, &graph);
                ok = (r == GS_OK) && e == hipSuccess && graph != nullptr;
            }
            ok = ok && hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) == hipSuccess;
            if (graph) (void)hipGraphDestroy(graph);
            if (cs) (void)hipStreamDestroy(cs);
            if (ok) {
                if (g_graphs.size() >= 8) {  // evict the least recently used
                    size_t v = 0;
                    for (size_t i = 1; i < g_graphs.size(); ++i)
                        if (g_graphs[i].last_use < g_graphs[v].last_use) v = i;
                    (void)hipGraphExecDestroy(g_graphs[v].exec);
                    g_graphs.erase(g_graphs.begin() + v);
                }
                g_graphs.push_back(GraphEntry{key, device, exec, 0});
                hit = &g_graphs.back();
                ++g_captures;
            } else {
                (void)hipGetLastError();  // fall back to eager launches below
            }
        }
        if (hit) {
            hit->last_use = ++g_graph_clock;
            ++g_replays;
            GS_HIP(hipGraphLaunch(hit->exec, st), "gs_slam_localize/graph");
            launched = true;
        }
    }
    if (!launched) {
        const auto t0 = std::chrono::steady_clock::now();
        if ((rc = enqueue_loops(stream, numiters > 0))) return rc;
        if (numiters > 0) {
            const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
            const int launches = B * ((use_grad_lm ? 2 * numiters : numiters + 1) + 2);
            std::lock_guard<std::mutex> lock(g_graph_mu);
            if (g_auto_samples < 64) {
                ++g_auto_samples;
                g_auto_min_us = std::min(g_auto_min_us, us / launches);
            }
            return GS_OK;  // composed by the loop's last launch
        }
    }
    return gs_compose_poses(w.T, prev_poses, B, out_poses, stream);
}

size_t gs_pointfusion_update_ws_bytes(int B, int H, int W, int Nmax) {
    if (B <= 0 || H <= 0 || W <= 0 || Nmax <= 0) return 0;
    return fuse_layout(B, H, W, Nmax, nullptr, nullptr);
}

static int pointfusion_update_impl(const float *depth, const float *rgb, const float *intrinsics, const float *poses, int B, int H, int W,
                                   float *map_points, float *map_normals, float *map_colors, float *map_ccounts, int32_t *map_counts,
                                   int Nmax, float dist_th, float dot_th, float sigma, int32_t *stats, void *ws, size_t ws_bytes,
                                   gs_stream_t stream, void *tape, const char *name) {
    GS_REQUIRE(depth && rgb && intrinsics && poses && map_points && map_normals && map_colors && map_ccounts && map_counts,
               "%s: NULL argument", name);
    GS_REQUIRE(B > 0 && B <= 60 && H >= 2 && W >= 2 && Nmax > 0, "%s: bad shape", name);
    if (!ws || ws_bytes < gs_pointfusion_update_ws_bytes(B, H, W, Nmax)) {
        set_error("%s: workspace too small (%zu < %zu)", name, ws_bytes, gs_pointfusion_update_ws_bytes(B, H, W, Nmax));
        return GS_ERR_WORKSPACE_TOO_SMALL;
    }
    hipStream_t st = (hipStream_t)stream;
    FuseWs w;
    fuse_layout(B, H, W, Nmax, ws, &w);
    int rc;
    // live frame under its final pose: global maps, the sample confidence from the LOCAL vertex map
    // (fusionutils.py:650-652), and -- riding on the same pass over the pixels -- the correspondence stage's per-pixel
    // state and the counter block initialised (no memset, no alpha launch)
    unsigned long long *pix_key; unsigned int *pix_n;
    fusion_corr_init_ptrs(w.state, B, H, W, Nmax, &pix_key, &pix_n);
    if ((rc = vertex_normal_maps_fusion(depth, intrinsics, poses, B, H, W, w.gV, w.gN, w.alpha, sigma, 1e-7f, pix_key, pix_n, w.ctr,
                                        kCtrWords + 2 * 64, st))) return rc;
    // find_correspondences (fusionutils.py:549-577): active -> similar -> best unique per pixel, without the tables
    if ((rc = fusion_correspond(w.state, map_points, map_normals, map_ccounts, map_counts, B, Nmax, poses, intrinsics, H, W, w.gV, w.gN,
                                dist_th, dot_th, w.ctr, st))) return rc;
    // differentiable form: the winners and the matched rows' values before the merge go to the tape
    if (tape && (rc = fusion_tape_record(w.state, tape, B, H, W, Nmax, map_counts, map_points, map_normals, map_colors, map_ccounts, st)))
        return rc;
    // fuse_with_map (fusionutils.py:654-720): merge in place, then append the unmatched valid pixels
    if ((rc = fusion_merge_corr(w.state, w.ctr, w.gV, w.gN, rgb, w.alpha, B, H, W, Nmax, map_counts, map_points, map_normals, map_colors,
                                map_ccounts, st))) return rc;
    const int64_t HW = (int64_t)H * W;
    for (int b = 0; b < B; ++b) {
        const float *src[4] = {w.gV + b * HW * 3, w.gN + b * HW * 3, rgb + b * HW * 3, w.alpha + b * HW};
        float *dst[4] = {map_points + (size_t)b * Nmax * 3, map_normals + (size_t)b * Nmax * 3, map_colors + (size_t)b * Nmax * 3,
                         map_ccounts + (size_t)b * Nmax};
        const int widths[4] = {3, 3, 3, 1};
        if ((rc = fusion_append_corr(w.state, B, H, W, Nmax, b, depth, src, widths, dst, map_counts + b, Nmax, w.totals + b, w.sub, st)))
            return rc;
    }
    if ((rc = fusion_finish(w.state, B, H, W, Nmax, w.ctr, map_counts, w.totals, Nmax, w.appended, stats, st))) return rc;
    if (tape && (rc = fusion_tape_appended(tape, B, H, W, w.appended, st))) return rc;
    return GS_OK;
}

int gs_pointfusion_update(const float *depth, const float *rgb, const float *intrinsics, const float *poses, int B, int H, int W,
                          float *map_points, float *map_normals, float *map_colors, float *map_ccounts, int32_t *map_counts,
                          int Nmax, float dist_th, float dot_th, float sigma, int32_t *stats, void *ws, size_t ws_bytes,
                          gs_stream_t stream) {
    return pointfusion_update_impl(depth, rgb, intrinsics, poses, B, H, W, map_points, map_normals, map_colors, map_ccounts, map_counts,
                                   Nmax, dist_th, dot_th, sigma, stats, ws, ws_bytes, stream, nullptr, "gs_pointfusion_update");
}

size_t gs_pointfusion_update_tape_bytes(int B, int H, int W) {
    if (B <= 0 || H <= 0 || W <= 0) return 0;
    return fusion_tape_bytes(B, H, W);
}

int gs_pointfusion_update_taped(const float *depth, const float *rgb, const float *intrinsics, const float *poses, int B, int H, int W,
                                float *map_points, float *map_normals, float *map_colors, float *map_ccounts, int32_t *map_counts,
                                int Nmax, float dist_th, float dot_th, float sigma, int32_t *stats, void *tape, size_t tape_bytes,
                                void *ws, size_t ws_bytes, gs_stream_t stream) {
    GS_REQUIRE(tape && tape_bytes >= gs_pointfusion_update_tape_bytes(B, H, W), "gs_pointfusion_update_taped: tape missing or too small");
    return pointfusion_update_impl(depth, rgb, intrinsics, poses, B, H, W, map_points, map_normals, map_colors, map_ccounts, map_counts,
                                   Nmax, dist_th, dot_th, sigma, stats, ws, ws_bytes, stream, tape, "gs_pointfusion_update_taped");
}

size_t gs_pointfusion_update_backward_ws_bytes(int B, int H, int W) {
    if (B <= 0 || H <= 0 || W <= 0) return 0;
    const size_t npix = (size_t)B * H * W;
    // V, N, gV, gN (recomputed), alpha, g_alpha + the compaction's scratch
    return 4 * align_up(npix * 12, 256) + 2 * align_up(npix * 4, 256) + gs_compact_ws_bytes((int64_t)H * W) + 512;
}

int gs_pointfusion_update_backward(const float *depth, const float *rgb, const float *intrinsics, const float *poses, int B, int H,
                                   int W, float *map_points, float *map_normals, float *map_colors, float *map_ccounts,
                                   int32_t *map_counts, int Nmax, float sigma, const void *tape, size_t tape_bytes, float *G_points,
                                   float *G_normals, float *G_colors, float *G_ccounts, float *g_vertex, float *g_gvertex,
                                   float *g_gnormal, float *g_rgb, void *ws, size_t ws_bytes, gs_stream_t stream) {
    const char *name = "gs_pointfusion_update_backward";
    GS_REQUIRE(depth && rgb && intrinsics && poses && map_points && map_normals && map_colors && map_ccounts && map_counts && tape &&
                   G_points && G_normals && G_colors && G_ccounts && g_vertex && g_gvertex && g_gnormal && g_rgb, "%s: NULL argument", name);
    GS_REQUIRE(B > 0 && B <= 60 && H >= 2 && W >= 2 && Nmax > 0, "%s: bad shape", name);
    GS_REQUIRE(tape_bytes >= gs_pointfusion_update_tape_bytes(B, H, W), "%s: tape too small", name);
    if (!ws || ws_bytes < gs_pointfusion_update_backward_ws_bytes(B, H, W)) {
        set_error("%s: workspace too small (%zu < %zu)", name, ws_bytes, gs_pointfusion_update_backward_ws_bytes(B, H, W));
        return GS_ERR_WORKSPACE_TOO_SMALL;
    }
    hipStream_t st = (hipStream_t)stream;
    const size_t npix = (size_t)B * H * W;
    char *p = (char *)ws;
    auto take = [&](size_t bytes) { char *o = p; p += align_up(bytes, 256); return o; };
    float *V = (float *)take(npix * 12), *N = (float *)take(npix * 12), *gV = (float *)take(npix * 12), *gN = (float *)take(npix * 12);
    float *alpha = (float *)take(npix * 4), *g_alpha = (float *)take(npix * 4);
    void *cws = p;
    int rc;
    // the frame's maps and sample confidences again (cheaper to recompute than to keep per frame)
    if ((rc = gs_vertex_normal_maps(depth, intrinsics, poses, B, 1, H, W, V, N, gV, gN, stream))) return rc;
    if ((rc = gs_get_alpha(V, (int64_t)npix, sigma, 1e-7f, alpha, stream))) return rc;
    if ((rc = fusion_update_reverse(tape, B, H, W, Nmax, depth, gV, gN, rgb, alpha, map_points, map_normals, map_colors, map_ccounts,
                                    map_counts, G_points, G_normals, G_colors, G_ccounts, g_gvertex, g_gnormal, g_rgb, g_alpha, cws, st)))
        return rc;
    // alpha = f(local vertex map): its adjoint is the only one the local vertex map receives from the update
    // (gs_get_alpha_backward adds into its output)
    GS_HIP(hipMemsetAsync(g_vertex, 0, npix * 12, st), name);
    return gs_get_alpha_backward(V, (int64_t)npix, sigma, 1e-7f, g_alpha, g_vertex, stream);
}



size_t gs_aggregate_update_ws_bytes(int B, int H, int W) {
    if (B <= 0 || H <= 0 || W <= 0) return 0;
    const size_t npix = (size_t)B * H * W;
    return 2 * align_up(npix * 12, 256) + 256 + align_up((size_t)B * 4, 256) + gs_compact_ws_bytes((int64_t)H * W) + 256;
}

int gs_aggregate_update(const float *depth, const float *rgb, const float *intrinsics, const float *poses, int B, int H, int W,
                        float *map_points, float *map_normals, float *map_colors, int32_t *map_counts, int Nmax, int32_t *stats,
                        void *ws, size_t ws_bytes, gs_stream_t stream) {
    const char *name = "gs_aggregate_update";
    GS_REQUIRE(depth && rgb && intrinsics && poses && map_points && map_normals && map_colors && map_counts, "%s: NULL argument", name);
    GS_REQUIRE(B > 0 && B <= 60 && H >= 2 && W >= 2 && Nmax > 0, "%s: bad shape", name);
    if (!ws || ws_bytes < gs_aggregate_update_ws_bytes(B, H, W)) {
        set_error("%s: workspace too small (%zu < %zu)", name, ws_bytes, gs_aggregate_update_ws_bytes(B, H, W));
        return GS_ERR_WORKSPACE_TOO_SMALL;
    }
    hipStream_t st = (hipStream_t)stream;
    const size_t npix = (size_t)B * H * W;
    char *p = (char *)ws;
    float *gV = (float *)p; p += align_up(npix * 12, 256);
    float *gN = (float *)p; p += align_up(npix * 12, 256);
    int32_t *overflow = (int32_t *)p; p += 256;
    int32_t *appended = (int32_t *)p; p += align_up((size_t)B * 4, 256);
    void *cws = p;
    GS_HIP(hipMemsetAsync(overflow, 0, 256 + align_up((size_t)B * 4, 256), st), name);
    int rc;
    if ((rc = gs_vertex_normal_maps(depth, intrinsics, poses, B, 1, H, W, nullptr, nullptr, gV, gN, stream))) return rc;
    const int64_t HW = (int64_t)H * W;
    for (int b = 0; b < B; ++b) {
        const float *src[3] = {gV + b * HW * 3, gN + b * HW * 3, rgb + b * HW * 3};
        float *dst[3] = {map_points + (size_t)b * Nmax * 3, map_normals + (size_t)b * Nmax * 3, map_colors + (size_t)b * Nmax * 3};
        const int widths[3] = {3, 3, 3};
        if ((rc = append_valid_pixels(3, depth + b * HW, HW, src, widths, dst, map_counts + b, Nmax, appended + b, overflow, cws, st)))
            return rc;
    }
    if (stats) {
        hipLaunchKernelGGL(fuse_stats_k, dim3(1), dim3(64), 0, st, overflow, overflow, overflow, (const float *)overflow, appended, B, stats);
        GS_LAUNCH_CHECK(name);
    }
    return GS_OK;
}

size_t gs_slam_localize_tape_bytes(int B, int H, int W, int ds, int Nmax, int numiters, int use_grad_lm) {
    if (B <= 0 || H <= 0 || W <= 0 || ds <= 0 || Nmax <= 0 || numiters < 0) return 0;
    return loc_tape_layout(B, H, W, ds, Nmax, numiters, use_grad_lm, nullptr, nullptr);
}

int gs_slam_localize_taped(const float *depth, const float *gvertex, const float *intrinsics, const float *prev_poses, int B, int H,
                           int W, int ds, const float *map_points, const float *map_normals, const int32_t *map_counts, int Nmax,
                           int use_grad_lm, int numiters, float damp, float dist_thresh, float lambda_max, float Bp, float B2,
                           float nu, float *out_poses, void *tape, size_t tape_bytes, void *ws, size_t ws_bytes,
                           gs_stream_t stream) {
    GS_REQUIRE(depth && gvertex && intrinsics && prev_poses && map_points && map_normals && map_counts && out_poses && tape,
               "gs_slam_localize_taped: NULL argument");
    GS_REQUIRE(B > 0 && H >= 2 && W >= 2 && ds > 0 && Nmax > 0 && numiters >= 0, "gs_slam_localize_taped: bad shape");
    if (!ws || ws_bytes < gs_slam_localize_ws_bytes(B, H, W, ds, Nmax)) {
        set_error("gs_slam_localize_taped: workspace too small (%zu < %zu)", ws_bytes, gs_slam_localize_ws_bytes(B, H, W, ds, Nmax));
        return GS_ERR_WORKSPACE_TOO_SMALL;
    }
    GS_REQUIRE(tape_bytes >= gs_slam_localize_tape_bytes(B, H, W, ds, Nmax, numiters, use_grad_lm),
               "gs_slam_localize_taped: tape too small");
    LocWs w;
    loc_layout(B, H, W, ds, Nmax, ws, &w);
    LocTape tp;
    loc_tape_layout(B, H, W, ds, Nmax, numiters, use_grad_lm, tape, &tp);
    const int capS = cdiv(H, ds) * cdiv(W, ds), capT = target_cap(Nmax);
    int rc;
    if (B == 1) {
        const DsJob frame{depth, gvertex, tp.src, tp.src_pix, tp.ns};
        if ((rc = project_target1(map_points, map_counts, Nmax, prev_poses, intrinsics, H, W, ds, map_normals, capT, w.rows, w.nrows,
                                  w.tgt, w.tnrm, tp.nt, w.scan, w.scan_orig, w.pix_start, tp.tgt_index, nullptr, w.sub, w.sub_bytes,
                                  (hipStream_t)stream, &frame))) return rc;
    } else {
        if ((rc = gs_downsample_frame(depth, gvertex, nullptr, nullptr, B, H, W, ds, capS, tp.src, nullptr, nullptr, tp.src_pix, tp.ns,
                                      w.sub, w.sub_bytes, stream))) return rc;
        if ((rc = gs_project_active(map_points, map_counts, B, Nmax, prev_poses, intrinsics, H, W, ds, w.rows, w.nrows, w.sub,
                                    w.sub_bytes, stream))) return rc;
        if ((rc = gs_build_icp_target(w.rows, w.nrows, (int64_t)B * Nmax, B, H, W, ds, map_points, map_normals, Nmax, capT, w.tgt,
                                      w.tnrm, tp.nt, w.scan, w.scan_orig, w.pix_start, tp.tgt_index, nullptr, w.sub, w.sub_bytes, stream))) return rc;
    }
    for (int b = 0; b < B; ++b) {
        const gs_icp_hints hints{w.scan + (size_t)b * capT * 3, w.scan_orig + (size_t)b * capT, tp.src_pix + (size_t)b * capS,
                                 w.pix_start + (size_t)b * (capS + 1), nullptr, cdiv(W, ds), cdiv(H, ds),
                                 prev_poses + 16 * b, intrinsics + 16 * b, ds};
        if ((rc = icp_localize_run(use_grad_lm, tp.src + (size_t)b * capS * 3, tp.ns + b, capS, w.tgt + (size_t)b * capT * 3,
                                   w.tnrm + (size_t)b * capT * 3, tp.nt + b, capT, numiters, damp, dist_thresh, lambda_max, Bp, B2, nu,
                                   &hints, tp.T + 16 * b, w.sub, w.sub_bytes, (hipStream_t)stream, tp.icp + (size_t)b * tp.icp_bytes,
                                   tp.icp_bytes, prev_poses + 16 * b, out_poses + 16 * b)))
            return rc;
    }
    if (numiters == 0) return gs_compose_poses(tp.T, prev_poses, B, out_poses, stream);
    return GS_OK;
}

size_t gs_slam_localize_backward_ws_bytes(int B, int H, int W, int ds, int Nmax) {
    if (B <= 0 || H <= 0 || W <= 0 || ds <= 0 || Nmax <= 0) return 0;
    const int capS = cdiv(H, ds) * cdiv(W, ds), capT = target_cap(Nmax);
    // tgt, tnrm, g_tgt, g_nrm (one batch element at a time) + g_src + g_T, g_init, eye + the ICP reverse pass's own
    return 4 * align_up((size_t)capT * 12, 256) + align_up((size_t)capS * 12, 256) + 3 * align_up((size_t)B * 64, 256) +
           align_up(gs_icp_backward_ws_bytes(capS), 256);
}

int gs_slam_localize_backward(const float *prev_poses, int B, int H, int W, int ds, const float *map_points,
                              const float *map_normals, int Nmax, int use_grad_lm, int numiters, float dist_thresh,
                              float lambda_max, float Bp, float B2, float nu, const void *tape, size_t tape_bytes,
                              const float *grad_out_poses, float *grad_gvertex, float *grad_map_points, float *grad_map_normals,
                              float *grad_prev_poses, int accumulate_map_grads, void *ws, size_t ws_bytes, gs_stream_t stream) {
    const char *name = "gs_slam_localize_backward";
    GS_REQUIRE(prev_poses && map_points && map_normals && tape && grad_out_poses && grad_gvertex && grad_prev_poses,
               "%s: NULL argument", name);
    GS_REQUIRE(B > 0 && H >= 2 && W >= 2 && ds > 0 && Nmax > 0 && numiters >= 0, "%s: bad shape", name);
    if (!ws || ws_bytes < gs_slam_localize_backward_ws_bytes(B, H, W, ds, Nmax)) {
        set_error("%s: workspace too small", name);
        return GS_ERR_WORKSPACE_TOO_SMALL;
    }
    GS_REQUIRE(tape_bytes >= gs_slam_localize_tape_bytes(B, H, W, ds, Nmax, numiters, use_grad_lm), "%s: tape too small", name);
    hipStream_t st = (hipStream_t)stream;
    LocTape tp;
    loc_tape_layout(B, H, W, ds, Nmax, numiters, use_grad_lm, (void *)tape, &tp);
    const int capS = cdiv(H, ds) * cdiv(W, ds), capT = target_cap(Nmax);
    char *p = (char *)ws;
    auto take = [&](size_t bytes) { char *o = p; p += align_up(bytes, 256); return o; };
    float *tgt = (float *)take((size_t)capT * 12), *tnrm = (float *)take((size_t)capT * 12);
    float *g_tgt = (float *)take((size_t)capT * 12), *g_nrm = (float *)take((size_t)capT * 12);
    float *g_src = (float *)take((size_t)capS * 12);
    float *g_T = (float *)take((size_t)B * 64), *g_init = (float *)take((size_t)B * 64), *eye = (float *)take((size_t)B * 64);
    void *sub = p;
    const size_t sub_bytes = gs_icp_backward_ws_bytes(capS);

    GS_HIP(hipMemsetAsync(grad_gvertex, 0, (size_t)B * H * W * 12, st), name);
    if (grad_map_points && !accumulate_map_grads) GS_HIP(hipMemsetAsync(grad_map_points, 0, (size_t)B * Nmax * 12, st), name);
    if (grad_map_normals && !accumulate_map_grads) GS_HIP(hipMemsetAsync(grad_map_normals, 0, (size_t)B * Nmax * 12, st), name);
    hipLaunchKernelGGL(eye4_k, dim3(cdiv(16 * B, 64)), dim3(64), 0, st, eye, B);
    hipLaunchKernelGGL(compose_bwd_k, dim3(cdiv(B, 64)), dim3(64), 0, st, tp.T, prev_poses, grad_out_poses, B, g_T, grad_prev_poses);
    GS_LAUNCH_CHECK(name);
    const int gb = min(cdiv(capS, 256), 512);
    for (int b = 0; b < B; ++b) {
        // NB the gathered arrays hold one batch element at a time: index them with b = 0
        hipLaunchKernelGGL(regather_k, dim3(gb, 1), dim3(256), 0, st, tp.tgt_index + (size_t)b * capT, tp.nt + b, capT,
                           map_points + (size_t)b * Nmax * 3, map_normals + (size_t)b * Nmax * 3, Nmax, tgt, tnrm);
        GS_LAUNCH_CHECK(name);
        int rc;
        if ((rc = gs_icp_point_to_plane_backward(tp.src + (size_t)b * capS * 3, tp.ns + b, capS, tgt, tnrm, tp.nt + b, capT, eye + 16 * b, numiters,
                                                 dist_thresh, use_grad_lm, lambda_max, Bp, B2, nu, tp.icp + (size_t)b * tp.icp_bytes,
                                                 tp.icp_bytes, g_T + 16 * b, g_src, grad_map_points ? g_tgt : nullptr,
                                                 grad_map_normals ? g_nrm : nullptr, g_init + 16 * b, sub, sub_bytes, stream)))
            return rc;
        hipLaunchKernelGGL(scatter_grads_k, dim3(gb), dim3(256), 0, st, g_src, tp.src_pix + (size_t)b * capS, tp.ns + b, capS,
                           cdiv(W, ds), ds, H, W, grad_gvertex + (size_t)b * H * W * 3, g_tgt, g_nrm, tp.tgt_index + (size_t)b * capT,
                           tp.nt + b, capT, grad_map_points ? grad_map_points + (size_t)b * Nmax * 3 : nullptr,
                           grad_map_normals ? grad_map_normals + (size_t)b * Nmax * 3 : nullptr, accumulate_map_grads);
        GS_LAUNCH_CHECK(name);
    }
    return GS_OK;
}

}  // extern "C"
