// project.hip -- active-map-point search (P) and the map-side downsample gather (S).
//
// HBM-bound stream over the map: 12 B read per map point, 32 B written per active row.  The
// projection is evaluated twice (count pass + write pass) instead of staging flags: the second
// read of the points is served by L2 / Infinity Cache and costs less than a 4 B/pt flag round trip.
#include "gs_common.hpp"
#include "gs_compact.hpp"
#include "gs_project.hpp"

namespace gs {

// per-block LDS copy of the cameras (up to kCamB batch elements): filled by ActivePred::block_init in both
// compaction passes, so the projection needs no preparation launch
constexpr int kCamB = 32;
__device__ __forceinline__ Cam *cam_cache() {
    __shared__ Cam cams[kCamB];
    return cams;
}

__global__ void build_cams_k(const float *__restrict__ poses, const float *__restrict__ Ks, int B, Cam *__restrict__ cams) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const float *T = poses + 16 * b;
    Cam c;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) c.R[3 * i + j] = T[4 * i + j];
    const float t0 = T[3], t1 = T[7], t2 = T[11];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        // row j of -R^T is (-R[0][j], -R[1][j], -R[2][j])
        c.tinv[j] = ((-T[j]) * t0 + (-T[4 + j]) * t1) + (-T[8 + j]) * t2;
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) c.K[i] = Ks[16 * b + i];
    cams[b] = c;
}

struct ActivePred {
    const float *points;
    const int32_t *counts;
    const Cam *cams;
    int Nmax, H, W, ds;
    float umax, vmax;
    __device__ bool operator()(int64_t i) const {
        const int b = (int)(i / Nmax), n = (int)(i - (int64_t)b * Nmax);
        if (n >= counts[b]) return false;
        int h, w;
        if (!project_point(cams[b], ld3(points, i), H, W, umax, vmax, h, w)) return false;
        return ds <= 0 || ((h % ds == 0) && (w % ds == 0));
    }
};
struct ActiveWriter {
    const float *points;
    const Cam *cams;
    int64_t *rows;
    int Nmax, H, W;
    float umax, vmax;
    __device__ void operator()(int64_t i, int64_t pos) const {
        const int b = (int)(i / Nmax), n = (int)(i - (int64_t)b * Nmax);
        int h, w;
        project_point(cams[b], ld3(points, i), H, W, umax, vmax, h, w);
        longlong4 r;  // one 32-byte store per row
        r.x = b; r.y = n; r.z = h; r.w = w;
        *reinterpret_cast<longlong4 *>(rows + 4 * pos) = r;
    }
};

// the same pair with the cameras built per block in LDS (B <= kCamB)
struct ActivePredL {
    const float *points;
    const int32_t *counts;
    const float *poses, *Ks;
    int B, Nmax, H, W, ds;
    float umax, vmax;
    __device__ void block_init(int /*pass*/, int /*bid*/, int /*nb*/) const {
        for (int b = threadIdx.x; b < B; b += blockDim.x) cam_cache()[b] = make_cam(poses + 16 * b, Ks + 16 * b);
    }
    struct Item { f3 p; };
    __device__ Item fetch(int64_t i) const { return Item{ld3(points, i)}; }
    __device__ bool test(const Item &it, int64_t i) const {
        const int b = (int)(i / Nmax), n = (int)(i - (int64_t)b * Nmax);
        if (n >= counts[b]) return false;
        int h, w;
        if (!project_point(cam_cache()[b], it.p, H, W, umax, vmax, h, w)) return false;
        return ds <= 0 || ((h % ds == 0) && (w % ds == 0));
    }
    __device__ bool operator()(int64_t i) const { return test(fetch(i), i); }
};
struct ActiveWriterL {
    const float *points;
    int64_t *rows;
    int Nmax, H, W;
    float umax, vmax;
    __device__ void put(const ActivePredL::Item &it, int64_t i, int64_t pos) const {
        const int b = (int)(i / Nmax), n = (int)(i - (int64_t)b * Nmax);
        int h, w;
        project_point(cam_cache()[b], it.p, H, W, umax, vmax, h, w);
        longlong4 r;  // one 32-byte store per row
        r.x = b; r.y = n; r.z = h; r.w = w;
        *reinterpret_cast<longlong4 *>(rows + 4 * pos) = r;
    }
    __device__ void operator()(int64_t i, int64_t pos) const { put(ActivePredL::Item{ld3(points, i)}, i, pos); }
};

// The same projection for ONE sequence when its result feeds the ICP target build: the count pass also zeroes
// the per-pixel counters, the write pass also builds the per-pixel histogram of the ds-grid rows (what
// tgt_init_k and tgt_gather_count_k's counting would do in two more launches).
struct ActivePredH {
    const float *points;
    const int32_t *counts;
    const float *poses, *Ks;
    int Nmax, H, W, ds;
    float umax, vmax;
    int *cnt, *fill;  // (npix) each
    int npix;
    __device__ void block_init(int pass, int bid, int nb) const {
        if (threadIdx.x == 0) cam_cache()[0] = make_cam(poses, Ks);
        if (pass == 0)
            for (int i = bid * blockDim.x + threadIdx.x; i < npix; i += nb * blockDim.x) { cnt[i] = 0; fill[i] = 0; }
    }
    struct Item { f3 p; };
    __device__ Item fetch(int64_t i) const { return Item{ld3(points, i)}; }
    __device__ bool test(const Item &it, int64_t i) const {
        if (i >= counts[0]) return false;
        int h, w;
        if (!project_point(cam_cache()[0], it.p, H, W, umax, vmax, h, w)) return false;
        return (h % ds == 0) && (w % ds == 0);
    }
    __device__ bool operator()(int64_t i) const { return test(fetch(i), i); }
};
struct ActiveWriterH {
    const float *points;
    int64_t *rows;
    int H, W, ds, Wd;
    float umax, vmax;
    int *cnt;
    __device__ void put(const ActivePredH::Item &it, int64_t i, int64_t pos) const {
        int h, w;
        project_point(cam_cache()[0], it.p, H, W, umax, vmax, h, w);
        longlong4 r;
        r.x = 0; r.y = i; r.z = h; r.w = w;
        *reinterpret_cast<longlong4 *>(rows + 4 * pos) = r;
        atomicAdd(cnt + (h / ds) * Wd + (w / ds), 1);
    }
    __device__ void operator()(int64_t i, int64_t pos) const { put(ActivePredH::Item{ld3(points, i)}, i, pos); }
};

// starts[b] = first table row with row.b >= b (rows sorted by b); starts[B] = n_rows
__global__ void table_starts_k(const int64_t *__restrict__ rows, const int32_t *__restrict__ d_n, int B, int32_t *__restrict__ starts) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b > B) return;
    const int n = *d_n;
    int lo = 0, hi = n;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (rows[4 * (int64_t)mid] < b) lo = mid + 1; else hi = mid;
    }
    starts[b] = lo;
}

__global__ void gather_table_k(const int64_t *__restrict__ rows, const int32_t *__restrict__ d_n,
                               const int32_t *__restrict__ starts, const float *__restrict__ attr, int B, int Nmax, int C,
                               int cap, float *__restrict__ out, int32_t *__restrict__ counts) {
    const int n = *d_n;
    if (blockIdx.x == 0 && threadIdx.x < B && counts) counts[threadIdx.x] = starts[threadIdx.x + 1] - starts[threadIdx.x];
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int b = (int)rows[4 * i], p = (int)rows[4 * i + 1];
        const int k = (int)(i - starts[b]);
        if (k >= cap) continue;
        const float *s = attr + ((int64_t)b * Nmax + p) * C;
        float *d = out + ((int64_t)b * cap + k) * C;
        for (int c = 0; c < C; ++c) d[c] = s[c];
    }
}

__global__ void table_ds_mask_k(const int64_t *__restrict__ rows, int64_t n, int ds, uint8_t *__restrict__ mask) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        mask[i] = ((rows[4 * i + 2] % ds) == 0 && (rows[4 * i + 3] % ds) == 0) ? 1 : 0;
}

// ---------------------------------------------------------------- target cloud in pixel order
// The ICP target (map points on the ds-grid of the previous frame) arrives in MAP order, which stops
// being spatially coherent once the map holds points of many frames.  Bucketing the rows by their
// ds-grid pixel gives a scan order in which consecutive points are image neighbours again (tight AABBs
// for the nearest-neighbour pruning) and, per pixel, one target index to seed the first association
// with (a projective guess used only as a SEED: the search stays exact).  Indices reported to the caller
// stay in the reference's order through `orig`.
__global__ void pix_count_k(const int64_t *__restrict__ rows, const int32_t *__restrict__ d_n,
                            const int32_t *__restrict__ starts, int Wd, int npix, int ds, int *__restrict__ cnt) {
    const int n = *d_n;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const longlong4 r = *reinterpret_cast<const longlong4 *>(rows + 4 * i);
        const int b = (int)r.x;
        const int pix = (int)(r.z / ds) * Wd + (int)(r.w / ds);
        atomicAdd(cnt + (int64_t)b * npix + pix, 1);
    }
}
// one block per batch element: exclusive scan of the pixel counts -> start[0..npix] (start[npix] = total).
// Bins are staged through LDS in tiles (coalesced loads / stores); inside a tile every thread owns a
// contiguous run, so the block-level part is ONE scan of 1024 partials per tile.
constexpr int PIX_TILE = 24 * 1024;  // bins per LDS tile (96 KiB)
__global__ __launch_bounds__(1024) void pix_scan_k(const int *__restrict__ cnt, int npix, int *__restrict__ start) {
    __shared__ int sm[1024 / 64 + 1];
    __shared__ int bins[PIX_TILE];
    __shared__ int carry;
    const int b = blockIdx.x;
    cnt += (int64_t)b * npix; start += (int64_t)b * (npix + 1);
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int t0 = 0; t0 < npix; t0 += PIX_TILE) {
        const int nb = min(PIX_TILE, npix - t0);
        for (int i = threadIdx.x; i < nb; i += 1024) bins[i] = cnt[t0 + i];
        __syncthreads();
        const int per = (nb + 1023) / 1024;
        const int i0 = min(nb, (int)threadIdx.x * per), i1 = min(nb, i0 + per);
        int sum = 0;
        for (int i = i0; i < i1; ++i) sum += bins[i];
        int total;
        int run = carry + block_excl_scan<1024>(sum, sm, &total);
        for (int i = i0; i < i1; ++i) { const int c = bins[i]; bins[i] = run; run += c; }
        __syncthreads();
        for (int i = threadIdx.x; i < nb; i += 1024) start[t0 + i] = bins[i];
        if (threadIdx.x == 0) carry += total;
        __syncthreads();
    }
    if (threadIdx.x == 0) start[npix] = carry;
}
__global__ void pix_scatter_k(const int64_t *__restrict__ rows, const int32_t *__restrict__ d_n,
                              const int32_t *__restrict__ starts, int Wd, int npix, int ds,
                              const int *__restrict__ start, int *__restrict__ fill, const float *__restrict__ map_points,
                              int Nmax, int cap, float *__restrict__ scan_pts, int32_t *__restrict__ scan_orig,
                              int32_t *__restrict__ tgt_pix /* or NULL */) {
    const int n = *d_n;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const longlong4 r = *reinterpret_cast<const longlong4 *>(rows + 4 * i);
        const int b = (int)r.x;
        const int pix = (int)(r.z / ds) * Wd + (int)(r.w / ds);
        const int k = (int)(i - starts[b]);
        if (tgt_pix && k < cap) tgt_pix[(int64_t)b * cap + k] = pix;
        const int slot = start[(int64_t)b * (npix + 1) + pix] + atomicAdd(fill + (int64_t)b * npix + pix, 1);
        if (slot >= cap) continue;
        st3(scan_pts, (int64_t)b * cap + slot, ld3(map_points, (int64_t)b * Nmax + r.y));
        scan_orig[(int64_t)b * cap + slot] = (int32_t)k;
    }
}
// fused front of the ICP target build: reference-order gather of points + normals AND the per-pixel
// histogram / seed in one pass over the table rows
// zeroes the per-pixel counters AND finds the per-batch row ranges (two independent preparations, one launch)
__global__ void tgt_init_k(int *__restrict__ cnt, int *__restrict__ fill, int64_t n, const int64_t *__restrict__ rows,
                           const int32_t *__restrict__ d_n, int B, int32_t *__restrict__ starts) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        cnt[i] = 0; fill[i] = 0;
    }
    if (blockIdx.x == 0) {
        for (int b = threadIdx.x; b <= B; b += blockDim.x) {
            int lo = 0, hi = *d_n;
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (rows[4 * (int64_t)mid] < b) lo = mid + 1; else hi = mid;
            }
            starts[b] = lo;
        }
    }
}
__global__ void tgt_gather_count_k(const int64_t *__restrict__ rows, const int32_t *__restrict__ d_n,
                                   const int32_t *__restrict__ starts, const float *__restrict__ map_points,
                                   const float *__restrict__ map_normals, int B, int Nmax, int cap,
                                   float *__restrict__ tgt, float *__restrict__ tnrm, int32_t *__restrict__ counts, int Wd,
                                   int npix, int ds, int *__restrict__ cnt, int32_t *__restrict__ tgt_index) {
    const int n = *d_n;
    if (blockIdx.x == 0 && threadIdx.x < B) counts[threadIdx.x] = starts[threadIdx.x + 1] - starts[threadIdx.x];
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const longlong4 r = *reinterpret_cast<const longlong4 *>(rows + 4 * i);
        const int b = (int)r.x;
        const int k = (int)(i - starts[b]);
        const int pix = (int)(r.z / ds) * Wd + (int)(r.w / ds);
        atomicAdd(cnt + (int64_t)b * npix + pix, 1);
        if (k >= cap) continue;
        const int64_t src = (int64_t)b * Nmax + r.y, dst = (int64_t)b * cap + k;
        st3(tgt, dst, ld3(map_points, src));
        st3(tnrm, dst, ld3(map_normals, src));
        if (tgt_index) tgt_index[dst] = (int32_t)r.y;
    }
}

// one sequence: reference-order target arrays (gather) and the pixel-ordered copy (scatter) in one pass
__global__ void tgt_scatter_gather1_k(const int64_t *__restrict__ rows, const int32_t *__restrict__ d_n,
                                      const float *__restrict__ map_points, const float *__restrict__ map_normals, int cap,
                                      float *__restrict__ tgt, float *__restrict__ tnrm, int32_t *__restrict__ counts,
                                      int32_t *__restrict__ tgt_index, int Wd, int ds, const int *__restrict__ start,
                                      int *__restrict__ fill, float *__restrict__ scan_pts, int32_t *__restrict__ scan_orig,
                                      int32_t *__restrict__ tgt_pix /* or NULL */) {
    const int n = *d_n;
    if (blockIdx.x == 0 && threadIdx.x == 0) counts[0] = n;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const longlong4 r = *reinterpret_cast<const longlong4 *>(rows + 4 * i);
        const f3 p = ld3(map_points, r.y);
        if (i < cap) {
            st3(tgt, i, p);
            st3(tnrm, i, ld3(map_normals, r.y));
            if (tgt_index) tgt_index[i] = (int32_t)r.y;
        }
        const int pix = (int)(r.z / ds) * Wd + (int)(r.w / ds);
        if (tgt_pix && i < cap) tgt_pix[i] = pix;
        const int slot = start[pix] + atomicAdd(fill + pix, 1);
        if (slot >= cap) continue;
        st3(scan_pts, slot, p);
        scan_orig[slot] = (int32_t)i;
    }
}

__global__ void fill_i32b_k(int *__restrict__ p, int64_t n, int v) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] = v;
}

}  // namespace gs

using namespace gs;

extern "C" {

size_t gs_project_active_ws_bytes(int B, int Nmax) {
    return align_up(sizeof(Cam) * (size_t)(B > 0 ? B : 1), 256) + compact_ws_bytes((int64_t)B * Nmax);
}

int gs_project_active(const float *points, const int32_t *counts, int B, int Nmax, const float *poses,
                      const float *intrinsics, int H, int W, int ds, int64_t *out_rows, int32_t *out_count, void *ws,
                      size_t ws_bytes, gs_stream_t stream) {
    GS_REQUIRE(points && counts && poses && intrinsics && out_rows && out_count, "gs_project_active: NULL argument");
    GS_REQUIRE(B > 0 && Nmax > 0 && H > 0 && W > 0, "gs_project_active: bad shape B=%d Nmax=%d H=%d W=%d", B, Nmax, H, W);
    if (!ws || ws_bytes < gs_project_active_ws_bytes(B, Nmax)) {
        set_error("gs_project_active: workspace too small (%zu < %zu)", ws_bytes, gs_project_active_ws_bytes(B, Nmax));
        return GS_ERR_WORKSPACE_TOO_SMALL;
    }
    hipStream_t st = (hipStream_t)stream;
    Cam *cams = (Cam *)ws;
    void *cws = (char *)ws + align_up(sizeof(Cam) * (size_t)B, 256);
    const float umax = (float)((double)W - 0.999), vmax = (float)((double)H - 0.999);
    if (B <= kCamB) {  // cameras built per block in LDS: no preparation launch
        ActivePredL pred{points, counts, poses, intrinsics, B, Nmax, H, W, ds, umax, vmax};
        ActiveWriterL wr{points, out_rows, Nmax, H, W, umax, vmax};
        return compact_launch((int64_t)B * Nmax, pred, wr, out_count, cws, st, "gs_project_active");
    }
    hipLaunchKernelGGL(build_cams_k, dim3(cdiv(B, 64)), dim3(64), 0, st, poses, intrinsics, B, cams);
    GS_LAUNCH_CHECK("gs_project_active/cams");
    ActivePred pred{points, counts, cams, Nmax, H, W, ds, umax, vmax};
    ActiveWriter wr{points, cams, out_rows, Nmax, H, W, umax, vmax};
    return compact_launch((int64_t)B * Nmax, pred, wr, out_count, cws, st, "gs_project_active");
}

}  // extern "C"

namespace gs {
// gs_project_active (ds-grid) + gs_build_icp_target for one sequence in 4 launches instead of 6 (see ActivePredH)
size_t project_target1_ws_bytes(int H, int W, int ds, int Nmax) {
    const size_t npix = (size_t)cdiv(H, ds) * cdiv(W, ds);
    return compact_ws_bytes(Nmax) + 2 * align_up(npix * 4, 256) + compact_flags_bytes(Nmax) + compact_ws_bytes((int64_t)npix);
}
// `frame` (optional): the ds-grid source cloud of the live frame (gs_downsample_frame for one sequence) rides on the same two
// launches -- its inputs (the maps kernel's output) are ready when the projection's are
int project_target1(const float *points, const int32_t *counts, int Nmax, const float *poses, const float *intrinsics, int H,
                    int W, int ds, const float *map_normals, int cap, int64_t *rows, int32_t *nrows, float *tgt, float *tnrm,
                    int32_t *nt, float *scan_points, int32_t *scan_orig, int32_t *pix_start, int32_t *tgt_index, int32_t *tgt_pix,
                    void *ws, size_t ws_bytes, hipStream_t st, const DsJob *frame) {
    const char *name = "gs_slam_localize/target";
    if (!ws || ws_bytes < project_target1_ws_bytes(H, W, ds, Nmax)) {
        set_error("%s: workspace too small", name);
        return GS_ERR_WORKSPACE_TOO_SMALL;
    }
    const int Wd = cdiv(W, ds), npix = cdiv(H, ds) * Wd;
    char *p = (char *)ws;
    void *cws = p; p += compact_ws_bytes(Nmax);
    int *cnt = (int *)p, *fill = (int *)(p + align_up((size_t)npix * 4, 256));
    unsigned char *flags = (unsigned char *)(p + 2 * align_up((size_t)npix * 4, 256));  // the count pass's verdicts for the write pass
    const float umax = (float)((double)W - 0.999), vmax = (float)((double)H - 0.999);
    ActivePredH pred{points, counts, poses, intrinsics, Nmax, H, W, ds, umax, vmax, cnt, fill, npix};
    ActiveWriterH wr{points, rows, H, W, ds, Wd, umax, vmax, cnt};
    int rc;
    if (frame) {
        const int Hd = cdiv(H, ds);
        DsPred dp{frame->depth, W, Wd, ds};
        DsWriter dw{frame->gvertex, nullptr, nullptr, frame->out_points, nullptr, nullptr, frame->out_pix, W, Wd, ds};
        void *dws = (char *)flags + compact_flags_bytes(Nmax);  // (the frame's block counts: behind everything the projection uses)
        if (compact_blocks((int64_t)Hd * Wd) <= kSelfScanBlocks) {
            rc = compact_launch2((int64_t)Nmax, pred, wr, nrows, cws, flags, (int64_t)Hd * Wd, dp, dw, frame->count, dws, st, name);
        } else {  // a ds-grid of more than a million pixels: its write pass needs the scan launch, so it goes by itself
            rc = compact_launch((int64_t)Hd * Wd, dp, dw, frame->count, dws, st, name);
            if (rc == GS_OK) rc = compact_launch((int64_t)Nmax, pred, wr, nrows, cws, st, name, flags);
        }
    } else {
        rc = compact_launch((int64_t)Nmax, pred, wr, nrows, cws, st, name, flags);
    }
    if (rc) return rc;
    hipLaunchKernelGGL(pix_scan_k, dim3(1), dim3(1024), 0, st, cnt, npix, pix_start);
    hipLaunchKernelGGL(tgt_scatter_gather1_k, dim3(min(cdiv(Nmax, 256), 1024)), dim3(256), 0, st, rows, nrows, points, map_normals, cap,
                       tgt, tnrm, nt, tgt_index, Wd, ds, pix_start, fill, scan_points, scan_orig, tgt_pix);
    GS_LAUNCH_CHECK(name);
    return GS_OK;
}
}  // namespace gs

extern "C" {

size_t gs_gather_table_rows_ws_bytes(int B) { return align_up(sizeof(int32_t) * (size_t)(B + 1), 256); }

int gs_gather_table_rows(const int64_t *rows, const int32_t *d_n_rows, int64_t max_rows, const float *attr, int B,
                         int Nmax, int C, int cap, float *out, int32_t *counts, void *ws, size_t ws_bytes,
                         gs_stream_t stream) {
    GS_REQUIRE(rows && d_n_rows && attr && out, "gs_gather_table_rows: NULL argument");
    GS_REQUIRE(B > 0 && B <= 256 && Nmax > 0 && C > 0 && cap >= 0 && max_rows >= 0, "gs_gather_table_rows: bad shape (B <= 256)");
    if (!ws || ws_bytes < gs_gather_table_rows_ws_bytes(B)) {
        set_error("gs_gather_table_rows: workspace too small");
        return GS_ERR_WORKSPACE_TOO_SMALL;
    }
    hipStream_t st = (hipStream_t)stream;
    int32_t *starts = (int32_t *)ws;
    hipLaunchKernelGGL(table_starts_k, dim3(cdiv(B + 1, 64)), dim3(64), 0, st, rows, d_n_rows, B, starts);
    GS_LAUNCH_CHECK("gs_gather_table_rows/starts");
    const int nb = max_rows > 0 ? min(cdiv(max_rows, 256), 2048) : 1;
    hipLaunchKernelGGL(gather_table_k, dim3(nb), dim3(256), 0, st, rows, d_n_rows, starts, attr, B, Nmax,
                       C, cap, out, counts);
    GS_LAUNCH_CHECK("gs_gather_table_rows");
    return GS_OK;
}

size_t gs_bucket_by_pixel_ws_bytes(int B, int H, int W, int ds) {
    const size_t npix = (size_t)cdiv(H, ds) * cdiv(W, ds);
    return align_up(sizeof(int32_t) * (size_t)(B + 1), 256) + 2 * align_up((size_t)B * npix * 4, 256);
}

int gs_bucket_by_pixel(const int64_t *rows, const int32_t *d_n_rows, int64_t max_rows, int B, int H, int W, int ds,
                       const float *map_points, int Nmax, int cap, float *scan_points, int32_t *scan_orig,
                       int32_t *pix_start, int32_t *tgt_pix, void *ws, size_t ws_bytes, gs_stream_t stream) {
    GS_REQUIRE(rows && d_n_rows && map_points && scan_points && scan_orig && pix_start, "gs_bucket_by_pixel: NULL argument");
    GS_REQUIRE(B > 0 && B <= 256 && H > 0 && W > 0 && ds > 0 && Nmax > 0 && cap > 0 && max_rows >= 0, "gs_bucket_by_pixel: bad shape");
    if (!ws || ws_bytes < gs_bucket_by_pixel_ws_bytes(B, H, W, ds)) {
        set_error("gs_bucket_by_pixel: workspace too small");
        return GS_ERR_WORKSPACE_TOO_SMALL;
    }
    hipStream_t st = (hipStream_t)stream;
    const int Wd = cdiv(W, ds), npix = cdiv(H, ds) * Wd;
    char *p = (char *)ws;
    int32_t *starts = (int32_t *)p; p += align_up(sizeof(int32_t) * (size_t)(B + 1), 256);
    const size_t seg = align_up((size_t)B * npix * 4, 256);
    int *cnt = (int *)p, *fill = (int *)(p + seg);
    const int64_t nbins = (int64_t)B * npix;
    hipLaunchKernelGGL(tgt_init_k, dim3(min(cdiv(nbins, 256), 1024)), dim3(256), 0, st, cnt, fill, nbins, rows, d_n_rows, B, starts);
    const int nb = max_rows > 0 ? min(cdiv(max_rows, 256), 1024) : 1;
    hipLaunchKernelGGL(pix_count_k, dim3(nb), dim3(256), 0, st, rows, d_n_rows, starts, Wd, npix, ds, cnt);
    hipLaunchKernelGGL(pix_scan_k, dim3(B), dim3(1024), 0, st, cnt, npix, pix_start);
    hipLaunchKernelGGL(pix_scatter_k, dim3(nb), dim3(256), 0, st, rows, d_n_rows, starts, Wd, npix, ds, pix_start, fill, map_points,
                       Nmax, cap, scan_points, scan_orig, tgt_pix);
    GS_LAUNCH_CHECK("gs_bucket_by_pixel");
    return GS_OK;
}

size_t gs_build_icp_target_ws_bytes(int B, int H, int W, int ds) { return gs_bucket_by_pixel_ws_bytes(B, H, W, ds); }

int gs_build_icp_target(const int64_t *rows, const int32_t *d_n_rows, int64_t max_rows, int B, int H, int W, int ds,
                        const float *map_points, const float *map_normals, int Nmax, int cap, float *tgt, float *tgt_normals,
                        int32_t *counts, float *scan_points, int32_t *scan_orig, int32_t *pix_start, int32_t *tgt_index,
                        int32_t *tgt_pix, void *ws, size_t ws_bytes, gs_stream_t stream) {
    GS_REQUIRE(rows && d_n_rows && map_points && map_normals && tgt && tgt_normals && counts && scan_points && scan_orig && pix_start,
               "gs_build_icp_target: NULL argument");
    GS_REQUIRE(B > 0 && B <= 256 && H > 0 && W > 0 && ds > 0 && Nmax > 0 && cap > 0 && max_rows >= 0, "gs_build_icp_target: bad shape");
    if (!ws || ws_bytes < gs_build_icp_target_ws_bytes(B, H, W, ds)) {
        set_error("gs_build_icp_target: workspace too small");
        return GS_ERR_WORKSPACE_TOO_SMALL;
    }
    hipStream_t st = (hipStream_t)stream;
    const int Wd = cdiv(W, ds), npix = cdiv(H, ds) * Wd;
    char *p = (char *)ws;
    int32_t *starts = (int32_t *)p; p += align_up(sizeof(int32_t) * (size_t)(B + 1), 256);
    const size_t seg = align_up((size_t)B * npix * 4, 256);
    int *cnt = (int *)p, *fill = (int *)(p + seg);
    const int64_t nbins = (int64_t)B * npix;
    hipLaunchKernelGGL(tgt_init_k, dim3(min(cdiv(nbins, 256), 1024)), dim3(256), 0, st, cnt, fill, nbins, rows, d_n_rows, B, starts);
    const int nb = max_rows > 0 ? min(cdiv(max_rows, 256), 1024) : 1;
    hipLaunchKernelGGL(tgt_gather_count_k, dim3(nb), dim3(256), 0, st, rows, d_n_rows, starts, map_points, map_normals, B, Nmax,
                       cap, tgt, tgt_normals, counts, Wd, npix, ds, cnt, tgt_index);
    hipLaunchKernelGGL(pix_scan_k, dim3(B), dim3(1024), 0, st, cnt, npix, pix_start);
    hipLaunchKernelGGL(pix_scatter_k, dim3(nb), dim3(256), 0, st, rows, d_n_rows, starts, Wd, npix, ds, pix_start, fill, map_points,
                       Nmax, cap, scan_points, scan_orig, tgt_pix);
    GS_LAUNCH_CHECK("gs_build_icp_target");
    return GS_OK;
}

int gs_table_ds_mask(const int64_t *rows, int64_t n_rows, int ds, uint8_t *mask, gs_stream_t stream) {
    GS_REQUIRE(rows && mask && n_rows >= 0 && ds > 0, "gs_table_ds_mask: bad arguments");
    if (n_rows == 0) return GS_OK;
    hipLaunchKernelGGL(table_ds_mask_k, dim3(min(cdiv(n_rows, 256), 2048)), dim3(256), 0, (hipStream_t)stream, rows,
                       n_rows, ds, mask);
    GS_LAUNCH_CHECK("gs_table_ds_mask");
    return GS_OK;
}

}  // extern "C"
