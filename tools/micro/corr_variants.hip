// corr_variants.hip -- what bounds the map-wide kernels of gs_pointfusion_update?  (tools/corr_variants.py drives it)
//
// The product's corr_pass1_k (33 us at 1.7 M map points) and merge_corr_k (35 us) move 54 MB / 136 MB -- 7 / 17 us at
// 8 TB/s -- and issue ~9 us of VALU work.  This file holds the same arithmetic (the shared gs_project.hpp /
// gs_common.hpp, so the decisions are the product's) with the pieces switchable, so one run under rocprofv3 tells which
// piece the time belongs to:
//   I  items per thread (1, 2, 4, 8), in load-batched phases like the product's
//   F  bit 0: no atomics;  bit 1: no gathers from the frame (the point's own values stand in);  bit 2: read the pixel's
//      key first and skip an atomic that cannot lower it;  bit 3: gathers under EXEC masks instead of clamped indices;
//      bit 4 (with 3): the frame's normal, the point's normal and confidence only after the distance test
//      (merge: bit 0 no gathers, bit 1 gathers under EXEC masks)
//   grid_cap  0: one block per chunk (the product's form);  > 0: that many blocks walk the chunks with a stride
//      (the camera is set up once per block, and the blocks drift apart so that their phases overlap)
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fPIC -shared -I../../gradslam_amd/csrc
//        -I../../include corr_variants.hip -o libcorr_variants.so
#include "gs_project.hpp"

namespace gs {
void set_error(const char *, ...) {}
}
using namespace gs;

constexpr int T = 256;

template <int I, int F>
__global__ __launch_bounds__(T) void v_pass1_k(const float *__restrict__ mp, const float *__restrict__ mn, const float *__restrict__ cc,
                                               const int32_t *__restrict__ counts, int nchunks, const float *__restrict__ poses,
                                               const float *__restrict__ Ks, int H, int W, float umax, float vmax,
                                               const float *__restrict__ gv, const float *__restrict__ gn, float dist_th,
                                               float dot_th, unsigned long long *__restrict__ pix_key, int *__restrict__ pt_pix,
                                               int32_t *__restrict__ part_active, int32_t *__restrict__ part_similar) {
    __shared__ Cam cam;
    __shared__ int red[2][T / 64];
    __shared__ float raw[32];
    if (threadIdx.x < 32) {
        raw[threadIdx.x] = threadIdx.x < 16 ? poses[threadIdx.x] : Ks[threadIdx.x - 16];
        __builtin_amdgcn_wave_barrier();
        if (threadIdx.x == 0) cam = make_cam(raw, raw + 16);
    }
    const int cnt = counts[0];
    __syncthreads();
    for (int chunk = blockIdx.x; chunk < nchunks; chunk += gridDim.x) {
        const int n0 = chunk * (T * I) + threadIdx.x;
        f3 p[I];
        bool live[I];
#pragma unroll
        for (int k = 0; k < I; ++k) {
            const int n = n0 + k * T;
            live[k] = n < cnt;
            p[k] = ld3(mp, live[k] ? n : 0);
        }
        int px[I];
        bool act[I];
#pragma unroll
        for (int k = 0; k < I; ++k) {
            int h, w;
            act[k] = project_point(cam, p[k], H, W, umax, vmax, h, w) && live[k];
            px[k] = act[k] ? h * W + w : 0;
        }
        f3 fv[I], fn[I], q[I];
        float c[I];
        unsigned long long cur[I];
        if (F & 8) {  // loads under EXEC masks instead of clamped indices; with bit 4 the normals / confidence only after the distance test
#pragma unroll
            for (int k = 0; k < I; ++k) {
                fv[k] = f3{0, 0, 0}; fn[k] = f3{0, 0, 0}; q[k] = f3{0, 0, 0}; c[k] = 0.0f;
                if (act[k]) {
                    fv[k] = ld3(gv, px[k]);
                    if (!(F & 16)) { fn[k] = ld3(gn, px[k]); q[k] = ld3(mn, n0 + k * T); c[k] = cc[n0 + k * T]; }
                }
            }
            if (F & 16) {
#pragma unroll
                for (int k = 0; k < I; ++k) {
                    if (act[k]) {
                        const float dx = fv[k].x - p[k].x, dy = fv[k].y - p[k].y, dz = fv[k].z - p[k].z;
                        const float dist = sqrtf(__fmaf_rn(dz, dz, __fmaf_rn(dy, dy, dx * dx)));
                        if (dist < dist_th) { fn[k] = ld3(gn, px[k]); q[k] = ld3(mn, n0 + k * T); c[k] = cc[n0 + k * T]; }
                    }
                }
            }
        } else {
#pragma unroll
        for (int k = 0; k < I; ++k) {
            const int64_t pt = (live[k] && act[k]) ? n0 + k * T : 0, pix = px[k];
            q[k] = ld3(mn, pt); c[k] = cc[pt];
            if (F & 2) { fv[k] = p[k]; fn[k] = q[k]; } else { fv[k] = ld3(gv, pix); fn[k] = ld3(gn, pix); }
            if (F & 4) cur[k] = pix_key[pix];
        }
        }
        int n_act = 0, n_sim = 0;
#pragma unroll
        for (int k = 0; k < I; ++k) {
            if (!live[k]) continue;
            int out = -1;
            if (act[k]) {
                ++n_act;
                const float dx = fv[k].x - p[k].x, dy = fv[k].y - p[k].y, dz = fv[k].z - p[k].z;
                const float dist = sqrtf(__fmaf_rn(dz, dz, __fmaf_rn(dy, dy, dx * dx)));
                const float dot = (fn[k].x * q[k].x + fn[k].y * q[k].y) + fn[k].z * q[k].z;
                if (dist < dist_th && dot > dot_th) {
                    out = px[k];
                    ++n_sim;
                    const float inv_c = 1.0f / (c[k] + 1e-20f);
                    const float ex = p[k].x - fv[k].x, ey = p[k].y - fv[k].y, ez = p[k].z - fv[k].z;
                    const float ray = (ex * ex + ey * ey) + ez * ez;
                    const unsigned long long key = ((unsigned long long)fbits(inv_c) << 32) | fbits(ray);
                    if (!(F & 1)) {
                        if (!(F & 4) || key < cur[k]) atomicMin(pix_key + px[k], key);
                    }
                }
            }
            pt_pix[n0 + k * T] = out;
        }
        n_act = wave_sum_i(n_act);
        n_sim = wave_sum_i(n_sim);
        if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = n_act; red[1][threadIdx.x >> 6] = n_sim; }
        __syncthreads();
        if (threadIdx.x < 2) {
            int sum = 0;
            for (int wv = 0; wv < T / 64; ++wv) sum += red[threadIdx.x][wv];
            (threadIdx.x == 0 ? part_active : part_similar)[chunk] = sum;
        }
        __syncthreads();
    }
}

template <int I, int F>
__global__ __launch_bounds__(T) void v_merge_k(const int *__restrict__ pt_pix, const unsigned int *__restrict__ pix_n,
                                               const int32_t *__restrict__ counts, int nchunks, const float *__restrict__ gv,
                                               const float *__restrict__ gn, const float *__restrict__ rgb,
                                               const float *__restrict__ alpha, float *p, float *nn, float *cl, float *cc,
                                               int32_t *__restrict__ part_u) {
    __shared__ int red[T / 64];
    const int cnt = counts[0];
    for (int chunk = blockIdx.x; chunk < nchunks; chunk += gridDim.x) {
        int n_u = 0;
        const int n0 = chunk * (T * I) + threadIdx.x;
        f3 x[I], y[I], z[I];
        float c[I];
        int px[I];
        bool live[I];
#pragma unroll
        for (int k = 0; k < I; ++k) {
            const int n = n0 + k * T;
            live[k] = n < cnt;
            const int64_t pt = live[k] ? n : 0;
            px[k] = pt_pix[pt];
            x[k] = ld3(p, pt); y[k] = ld3(nn, pt); z[k] = ld3(cl, pt); c[k] = cc[pt];
        }
        unsigned int win[I];
        f3 fp[I], fn[I], fc[I];
        float a[I];
        if (F & 2) {
#pragma unroll
            for (int k = 0; k < I; ++k) {
                if (!live[k]) px[k] = -1;
                win[k] = 0xffffffffu;
                if (px[k] >= 0) win[k] = pix_n[px[k]];
            }
#pragma unroll
            for (int k = 0; k < I; ++k) {
                const bool m = px[k] >= 0 && win[k] == (unsigned int)(n0 + k * T);
                if (!m) px[k] = -1;
                a[k] = 0.0f; fp[k] = f3{0, 0, 0}; fn[k] = f3{0, 0, 0}; fc[k] = f3{0, 0, 0};
                if (m) { a[k] = alpha[px[k]]; fp[k] = ld3(gv, px[k]); fn[k] = ld3(gn, px[k]); fc[k] = ld3(rgb, px[k]); }
            }
        } else {
#pragma unroll
        for (int k = 0; k < I; ++k) {
            if (!live[k]) px[k] = -1;
            win[k] = (F & 1) ? (unsigned int)(n0 + k * T) : pix_n[px[k] >= 0 ? px[k] : 0];
        }
#pragma unroll
        for (int k = 0; k < I; ++k) {
            const bool m = px[k] >= 0 && win[k] == (unsigned int)(n0 + k * T);
            if (!m) px[k] = -1;
            const int64_t pix = m ? px[k] : 0;
            if (F & 1) { a[k] = c[k]; fp[k] = x[k]; fn[k] = y[k]; fc[k] = z[k]; }
            else { a[k] = alpha[pix]; fp[k] = ld3(gv, pix); fn[k] = ld3(gn, pix); fc[k] = ld3(rgb, pix); }
        }
        }
#pragma unroll
        for (int k = 0; k < I; ++k) {
            if (!live[k]) continue;
            const bool m = px[k] >= 0;
            n_u += m ? 1 : 0;
            const float ak = m ? a[k] : 0.0f;
            const f3 vp = m ? fp[k] : f3{0, 0, 0}, vn = m ? fn[k] : f3{0, 0, 0}, vc = m ? fc[k] : f3{0, 0, 0};
            const float c2 = c[k] + ak;
            const float inv = 1.0f / (c2 == 0.0f ? 1.0f : c2);
            const int64_t pt = n0 + k * T;
            st3(p, pt, f3{((c[k] * x[k].x) + (ak * vp.x)) * inv, ((c[k] * x[k].y) + (ak * vp.y)) * inv, ((c[k] * x[k].z) + (ak * vp.z)) * inv});
            st3(nn, pt, f3{((c[k] * y[k].x) + (ak * vn.x)) * inv, ((c[k] * y[k].y) + (ak * vn.y)) * inv, ((c[k] * y[k].z) + (ak * vn.z)) * inv});
            st3(cl, pt, f3{((c[k] * z[k].x) + (ak * vc.x)) * inv, ((c[k] * z[k].y) + (ak * vc.y)) * inv, ((c[k] * z[k].z) + (ak * vc.z)) * inv});
            cc[pt] = c2;
        }
        n_u = wave_sum_i(n_u);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = n_u;
        __syncthreads();
        if (threadIdx.x == 0) {
            int sum = 0;
            for (int wv = 0; wv < T / 64; ++wv) sum += red[wv];
            part_u[chunk] = sum;
        }
        __syncthreads();
    }
}

struct Args {
    const float *mp, *mn, *cc;
    const int32_t *counts;
    const float *poses, *Ks, *gv, *gn, *rgb, *alpha;
    float *p, *nn, *cl, *ccw;  // the merge's in-place arrays (scratch copies)
    unsigned long long *pix_key;
    unsigned int *pix_n;
    int *pt_pix;
    int32_t *part_a, *part_s, *part_u;
    int N, H, W;
    float dist_th, dot_th;
};

template <int I, int F>
static void launch_pass1(const Args &a, int grid_cap, hipStream_t st) {
    const int nchunks = (a.N + T * I - 1) / (T * I);
    const int grid = grid_cap > 0 && grid_cap < nchunks ? grid_cap : nchunks;
    const float umax = (float)((double)a.W - 0.999), vmax = (float)((double)a.H - 0.999);
    hipLaunchKernelGGL((v_pass1_k<I, F>), dim3(grid), dim3(T), 0, st, a.mp, a.mn, a.cc, a.counts, nchunks, a.poses, a.Ks, a.H, a.W, umax,
                       vmax, a.gv, a.gn, a.dist_th, a.dot_th, a.pix_key, a.pt_pix, a.part_a, a.part_s);
}
template <int I, int F>
static void launch_merge(const Args &a, int grid_cap, hipStream_t st) {
    const int nchunks = (a.N + T * I - 1) / (T * I);
    const int grid = grid_cap > 0 && grid_cap < nchunks ? grid_cap : nchunks;
    hipLaunchKernelGGL((v_merge_k<I, F>), dim3(grid), dim3(T), 0, st, (const int *)a.pt_pix, (const unsigned int *)a.pix_n, a.counts,
                       nchunks, a.gv, a.gn, a.rgb, a.alpha, a.p, a.nn, a.cl, a.ccw, a.part_u);
}

template <int I>
static int pass1_f(int F, const Args &a, int cap, hipStream_t st) {
    switch (F) {
        case 0: launch_pass1<I, 0>(a, cap, st); return 0;
        case 1: launch_pass1<I, 1>(a, cap, st); return 0;
        case 2: launch_pass1<I, 2>(a, cap, st); return 0;
        case 3: launch_pass1<I, 3>(a, cap, st); return 0;
        case 4: launch_pass1<I, 4>(a, cap, st); return 0;
        case 8: launch_pass1<I, 8>(a, cap, st); return 0;
        case 24: launch_pass1<I, 24>(a, cap, st); return 0;
    }
    return -1;
}
template <int I>
static int merge_f(int F, const Args &a, int cap, hipStream_t st) {
    switch (F) {
        case 0: launch_merge<I, 0>(a, cap, st); return 0;
        case 1: launch_merge<I, 1>(a, cap, st); return 0;
        case 2: launch_merge<I, 2>(a, cap, st); return 0;
    }
    return -1;
}

extern "C" int corr_variant(int which, int I, int F, int grid_cap, const Args *a, void *stream) {
    hipStream_t st = (hipStream_t)stream;
    int rc = -1;
    if (which == 0) {
        if (I == 1) rc = pass1_f<1>(F, *a, grid_cap, st);
        if (I == 2) rc = pass1_f<2>(F, *a, grid_cap, st);
        if (I == 4) rc = pass1_f<4>(F, *a, grid_cap, st);
        if (I == 8) rc = pass1_f<8>(F, *a, grid_cap, st);
    } else {
        if (I == 1) rc = merge_f<1>(F, *a, grid_cap, st);
        if (I == 2) rc = merge_f<2>(F, *a, grid_cap, st);
        if (I == 4) rc = merge_f<4>(F, *a, grid_cap, st);
        if (I == 8) rc = merge_f<8>(F, *a, grid_cap, st);
    }
    if (rc != 0) return rc;
    return (int)hipGetLastError();
}
