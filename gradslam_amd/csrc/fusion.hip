// fusion.hip -- PointFusion map update: similarity test (C), unique best correspondence per pixel
// (U), confidence-weighted merge (F) and the new-point mask (A).
//
// All HBM/L2-bound gathers keyed by the [b,n,h,w] table.  U replaces the reference's lexicographic
// row sort (torch.unique(dim=0) over P x 6 floats, ~90 % of its mapping time) by two per-pixel atomic
// min passes on order-preserving integer keys followed by a stable compaction over pixels, which
// yields the identical winners in the identical (b,h,w) output order.  The stand-alone entry points keep
// the reference's tables; gs_pointfusion_update runs the same tests as ONE table-free chain over the map
// points (corr_pass1_k / corr_pass2_k / merge_corr_k below).
#include "gs_common.hpp"
#include "gs_compact.hpp"
#include "gs_project.hpp"

namespace gs {

// ------------------------------------------------------------------ C
__global__ void similar_k(const int64_t *__restrict__ rows, const int32_t *__restrict__ d_n, const float *__restrict__ gv,
                          const float *__restrict__ gn, int H, int W, const float *__restrict__ mp,
                          const float *__restrict__ mn, int Nmax, float dist_th, float dot_th, uint8_t *__restrict__ keep,
                          float *__restrict__ max_dot) {
    const int n = *d_n;
    float md = 0.0f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const longlong4 r = *reinterpret_cast<const longlong4 *>(rows + 4 * i);
        const int64_t pix = ((int64_t)r.x * H + r.z) * W + r.w;
        const int64_t pt = (int64_t)r.x * Nmax + r.y;
        const f3 fv = ld3(gv, pix), fn = ld3(gn, pix), p = ld3(mp, pt), q = ld3(mn, pt);
        const float dx = fv.x - p.x, dy = fv.y - p.y, dz = fv.z - p.z;
        // (a-b).norm(dim=-1): sqrt(fma(z,z,fma(y,y,x*x)));  (a*b).sum(-1): unfused
        const float dist = sqrtf(__fmaf_rn(dz, dz, __fmaf_rn(dy, dy, dx * dx)));
        const float dot = (fn.x * q.x + fn.y * q.y) + fn.z * q.z;
        keep[i] = (dist < dist_th && dot > dot_th) ? 1 : 0;
        md = fmaxf(md, dot);
    }
    if (max_dot) {
        // the caller only warns when a dot product exceeds 1.001 (un-normalised normals): publish the
        // maximum only when it is above 1, so that the usual case costs no same-address atomics at all
        // (one per wave on one word serialises: ~80 us for a 1 M-row table)
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) md = fmaxf(md, __shfl_xor(md, off, kWave));
        if ((threadIdx.x & 63) == 0 && md > 1.0f) atomicMax(reinterpret_cast<int *>(max_dot), __float_as_int(md));
    }
}

// ------------------------------------------------------------------ U
// key = (1/(c + 1e-20), squared ray distance) as order-preserving bits (both are >= +0)
__device__ __forceinline__ unsigned long long unique_key(const float *__restrict__ gv, const float *__restrict__ mp,
                                                         const float *__restrict__ cc, int64_t pix, int64_t pt) {
    const float inv_c = 1.0f / (cc[pt] + 1e-20f);
    const f3 fv = ld3(gv, pix), p = ld3(mp, pt);
    const float dx = p.x - fv.x, dy = p.y - fv.y, dz = p.z - fv.z;
    const float ray = (dx * dx + dy * dy) + dz * dz;  // ((a-b)**2).sum(-1): unfused
    return ((unsigned long long)fbits(inv_c) << 32) | fbits(ray);
}

__global__ void unique_pass1_k(const int64_t *__restrict__ rows, const uint8_t *__restrict__ keep,
                               const int32_t *__restrict__ d_n, const float *__restrict__ gv, int H, int W,
                               const float *__restrict__ mp, const float *__restrict__ cc, int Nmax,
                               unsigned long long *__restrict__ pix_key) {
    const int n = *d_n;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        if (keep && !keep[i]) continue;
        const longlong4 r = *reinterpret_cast<const longlong4 *>(rows + 4 * i);
        const int64_t pix = ((int64_t)r.x * H + r.z) * W + r.w;
        atomicMin(pix_key + pix, unique_key(gv, mp, cc, pix, (int64_t)r.x * Nmax + r.y));
    }
}
__global__ void unique_pass2_k(const int64_t *__restrict__ rows, const uint8_t *__restrict__ keep,
                               const int32_t *__restrict__ d_n, const float *__restrict__ gv, int H, int W,
                               const float *__restrict__ mp, const float *__restrict__ cc, int Nmax,
                               const unsigned long long *__restrict__ pix_key, unsigned int *__restrict__ pix_n) {
    const int n = *d_n;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        if (keep && !keep[i]) continue;
        const longlong4 r = *reinterpret_cast<const longlong4 *>(rows + 4 * i);
        const int64_t pix = ((int64_t)r.x * H + r.z) * W + r.w;
        if (unique_key(gv, mp, cc, pix, (int64_t)r.x * Nmax + r.y) == pix_key[pix]) atomicMin(pix_n + pix, (unsigned int)r.y);
    }
}
struct PixPred {
    const unsigned int *pix_n;
    __device__ bool operator()(int64_t i) const { return pix_n[i] != 0xffffffffu; }
};
struct PixWriter {
    const unsigned int *pix_n;
    int64_t *rows;
    int H, W;
    __device__ void operator()(int64_t i, int64_t pos) const {
        const int64_t hw = (int64_t)H * W;
        const int b = (int)(i / hw);
        const int rem = (int)(i - (int64_t)b * hw);
        longlong4 r;
        r.x = b; r.y = pix_n[i]; r.z = rem / W; r.w = rem % W;
        *reinterpret_cast<longlong4 *>(rows + 4 * pos) = r;
    }
};

// fuse_with_map's append mask (slam/fusionutils.py:702-707) straight from the unique stage's per-pixel winner:
// valid depth and no correspondence at that pixel
struct AppendPredPix {
    const float *depth;
    const unsigned int *pix_n;
    __device__ bool operator()(int64_t i) const { return depth[i] > 0.0f && pix_n[i] == 0xffffffffu; }
};

struct ValidDepthPred {  // update_map_aggregate: every pixel with a valid depth (structures/utils.py:47-50)
    const float *depth;
    __device__ bool operator()(int64_t i) const { return depth[i] > 0.0f; }
};

// ------------------------------------------------------------------ F
__global__ void fill_i32_k(int *__restrict__ p, int64_t n, int v) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] = v;
}
__global__ void scatter_match_k(const int64_t *__restrict__ rows, const int32_t *__restrict__ d_n, int H, int W, int Nmax,
                                int *__restrict__ match_pix) {
    const int n = *d_n;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const longlong4 r = *reinterpret_cast<const longlong4 *>(rows + 4 * i);
        match_pix[(int64_t)r.x * Nmax + r.y] = (int)(r.z * W + r.w);
    }
}
// Every map point goes through the reference's formula, matched or not: an unmatched point becomes
// (c*x + 0*0) * (1/c), which is NOT bit-identical to x -- part of the reference's observable result
// (slam/fusionutils.py:678-699 operate on the whole padded tensors).
__global__ void merge_k(const int *__restrict__ match_pix, const int32_t *__restrict__ counts, int Nmax, int HW,
                        const float *__restrict__ gv, const float *__restrict__ gn, const float *__restrict__ rgb,
                        const float *__restrict__ alpha, const float *__restrict__ ip, const float *__restrict__ inn,
                        const float *__restrict__ ic, const float *__restrict__ icc, float *__restrict__ op,
                        float *__restrict__ on, float *__restrict__ oc, float *__restrict__ occ) {
    const int b = blockIdx.y;
    const int cnt = counts[b];
    for (int n = blockIdx.x * blockDim.x + threadIdx.x; n < Nmax; n += gridDim.x * blockDim.x) {
        const int64_t pt = (int64_t)b * Nmax + n;
        if (n >= cnt) {  // padding stays zero
            st3(op, pt, f3{0, 0, 0}); st3(on, pt, f3{0, 0, 0}); st3(oc, pt, f3{0, 0, 0}); occ[pt] = 0.0f;
            continue;
        }
        const int m = match_pix[pt];
        float a = 0.0f;
        f3 fp{0, 0, 0}, fn{0, 0, 0}, fc{0, 0, 0};
        if (m >= 0) {
            const int64_t pix = (int64_t)b * HW + m;
            a = alpha[pix]; fp = ld3(gv, pix); fn = ld3(gn, pix); fc = ld3(rgb, pix);
        }
        const float c = icc[pt];
        const float c2 = c + a;
        const float inv = 1.0f / (c2 == 0.0f ? 1.0f : c2);
        const f3 x = ld3(ip, pt), y = ld3(inn, pt), z = ld3(ic, pt);
        st3(op, pt, f3{((c * x.x) + (a * fp.x)) * inv, ((c * x.y) + (a * fp.y)) * inv, ((c * x.z) + (a * fp.z)) * inv});
        st3(on, pt, f3{((c * y.x) + (a * fn.x)) * inv, ((c * y.y) + (a * fn.y)) * inv, ((c * y.z) + (a * fn.z)) * inv});
        st3(oc, pt, f3{((c * z.x) + (a * fc.x)) * inv, ((c * z.y) + (a * fc.y)) * inv, ((c * z.z) + (a * fc.z)) * inv});
        occ[pt] = c2;
    }
}

// adjoint of merge_k.  x' = (c x + a xf) / c2 (c2 != 0):
//   x_bar = (c/c2) x'_bar ; xf_bar = (a/c2) x'_bar ; c_bar = c2_bar + sum (x - x').x'_bar / c2 ;
//   a_bar = c2_bar + sum (xf - x').x'_bar / c2      (sums over points, normals, colours)
// In-place form for an arena-backed map: same arithmetic on the rows that exist, padding untouched, and
// nothing at all when there is no correspondence (fuse_with_map skips the merge then, fusionutils.py:654).
__global__ void merge_inplace_k(const int *__restrict__ match_pix, const int32_t *__restrict__ counts,
                                const int32_t *__restrict__ d_n_rows, int Nmax, int HW, const float *__restrict__ gv,
                                const float *__restrict__ gn, const float *__restrict__ rgb, const float *__restrict__ alpha,
                                float *p, float *nn, float *cl, float *cc) {
    if (*d_n_rows == 0) return;
    const int b = blockIdx.y;
    const int cnt = min(counts[b], Nmax);
    for (int n = blockIdx.x * blockDim.x + threadIdx.x; n < cnt; n += gridDim.x * blockDim.x) {
        const int64_t pt = (int64_t)b * Nmax + n;
        const int m = match_pix[pt];
        float a = 0.0f;
        f3 fp{0, 0, 0}, fn{0, 0, 0}, fc{0, 0, 0};
        if (m >= 0) {
            const int64_t pix = (int64_t)b * HW + m;
            a = alpha[pix]; fp = ld3(gv, pix); fn = ld3(gn, pix); fc = ld3(rgb, pix);
        }
        const float c = cc[pt];
        const float c2 = c + a;
        const float inv = 1.0f / (c2 == 0.0f ? 1.0f : c2);
        const f3 x = ld3(p, pt), y = ld3(nn, pt), z = ld3(cl, pt);
        st3(p, pt, f3{((c * x.x) + (a * fp.x)) * inv, ((c * x.y) + (a * fp.y)) * inv, ((c * x.z) + (a * fp.z)) * inv});
        st3(nn, pt, f3{((c * y.x) + (a * fn.x)) * inv, ((c * y.y) + (a * fn.y)) * inv, ((c * y.z) + (a * fn.z)) * inv});
        st3(cl, pt, f3{((c * z.x) + (a * fc.x)) * inv, ((c * z.y) + (a * fc.y)) * inv, ((c * z.z) + (a * fc.z)) * inv});
        cc[pt] = c2;
    }
}

__global__ void merge_bwd_k(const int *__restrict__ match_pix, const int32_t *__restrict__ counts, int Nmax, int HW,
                            const float *__restrict__ gv, const float *__restrict__ gn, const float *__restrict__ rgb,
                            const float *__restrict__ alpha, const float *__restrict__ ip, const float *__restrict__ inn,
                            const float *__restrict__ ic, const float *__restrict__ icc, const float *__restrict__ gop,
                            const float *__restrict__ gon, const float *__restrict__ goc, const float *__restrict__ gocc,
                            float *__restrict__ gip, float *__restrict__ ginn, float *__restrict__ gic,
                            float *__restrict__ gicc, float *__restrict__ ggv, float *__restrict__ ggn,
                            float *__restrict__ grgb, float *__restrict__ galpha) {
    const int b = blockIdx.y;
    const int cnt = counts[b];
    for (int n = blockIdx.x * blockDim.x + threadIdx.x; n < Nmax; n += gridDim.x * blockDim.x) {
        const int64_t pt = (int64_t)b * Nmax + n;
        if (n >= cnt) {
            if (gip) st3(gip, pt, f3{0, 0, 0});
            if (ginn) st3(ginn, pt, f3{0, 0, 0});
            if (gic) st3(gic, pt, f3{0, 0, 0});
            if (gicc) gicc[pt] = 0.0f;
            continue;
        }
        const int m = match_pix[pt];
        const int64_t pix = (int64_t)b * HW + (m >= 0 ? m : 0);
        float a = 0.0f;
        f3 fp{0, 0, 0}, fn{0, 0, 0}, fc{0, 0, 0};
        if (m >= 0) { a = alpha[pix]; fp = ld3(gv, pix); fn = ld3(gn, pix); fc = ld3(rgb, pix); }
        const float c = icc[pt], c2 = c + a;
        const float inv = 1.0f / (c2 == 0.0f ? 1.0f : c2);
        const f3 x = ld3(ip, pt), y = ld3(inn, pt), z = ld3(ic, pt);
        const f3 gx = gop ? ld3(gop, pt) : f3{0, 0, 0}, gy = gon ? ld3(gon, pt) : f3{0, 0, 0},
                 gz = goc ? ld3(goc, pt) : f3{0, 0, 0};
        const float gc2 = gocc ? gocc[pt] : 0.0f;
        if (gip) st3(gip, pt, f3{c * inv * gx.x, c * inv * gx.y, c * inv * gx.z});
        if (ginn) st3(ginn, pt, f3{c * inv * gy.x, c * inv * gy.y, c * inv * gy.z});
        if (gic) st3(gic, pt, f3{c * inv * gz.x, c * inv * gz.y, c * inv * gz.z});
        // d x'/d c2 = -(c x + a xf) inv^2 = -x' inv (zero when c2 == 0, where the where() picks the constant 1)
        const f3 xo{((c * x.x) + (a * fp.x)) * inv, ((c * x.y) + (a * fp.y)) * inv, ((c * x.z) + (a * fp.z)) * inv};
        const f3 yo{((c * y.x) + (a * fn.x)) * inv, ((c * y.y) + (a * fn.y)) * inv, ((c * y.z) + (a * fn.z)) * inv};
        const f3 zo{((c * z.x) + (a * fc.x)) * inv, ((c * z.y) + (a * fc.y)) * inv, ((c * z.z) + (a * fc.z)) * inv};
        const float dinv = (c2 == 0.0f) ? 0.0f : 1.0f;
        const float s_out = (xo.x * gx.x + xo.y * gx.y + xo.z * gx.z) + (yo.x * gy.x + yo.y * gy.y + yo.z * gy.z) +
                            (zo.x * gz.x + zo.y * gz.y + zo.z * gz.z);
        const float s_in = (x.x * gx.x + x.y * gx.y + x.z * gx.z) + (y.x * gy.x + y.y * gy.y + y.z * gy.z) +
                           (z.x * gz.x + z.y * gz.y + z.z * gz.z);
        const float s_f = (fp.x * gx.x + fp.y * gx.y + fp.z * gx.z) + (fn.x * gy.x + fn.y * gy.y + fn.z * gy.z) +
                          (fc.x * gz.x + fc.y * gz.y + fc.z * gz.z);
        if (gicc) gicc[pt] = gc2 + (s_in - dinv * s_out) * inv;
        if (m >= 0) {  // unique rows: one map point per pixel -> plain stores
            if (ggv) st3(ggv, pix, f3{a * inv * gx.x, a * inv * gx.y, a * inv * gx.z});
            if (ggn) st3(ggn, pix, f3{a * inv * gy.x, a * inv * gy.y, a * inv * gy.z});
            if (grgb) st3(grgb, pix, f3{a * inv * gz.x, a * inv * gz.y, a * inv * gz.z});
            if (galpha) galpha[pix] = gc2 + (s_f - dinv * s_out) * inv;
        }
    }
}

// ------------------------------------------------------------------ differentiable arena update (one node per sequence)
// Tape of one in-place PointFusion update: what the reverse pass cannot recompute once later frames have changed the
// arena -- the per-pixel winner (which map point every pixel merged into; none: the pixel was appended if its depth
// is valid) and the ten attribute floats that map point held BEFORE the merge.
__global__ void tape_old_k(const unsigned int *__restrict__ pix_n, int64_t npix, int HW, int Nmax, const float *__restrict__ p,
                           const float *__restrict__ nn, const float *__restrict__ cl, const float *__restrict__ cc,
                           unsigned int *__restrict__ t_pix_n, float *__restrict__ t_old /* (npix, 10) */) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += (int64_t)gridDim.x * blockDim.x) {
        const unsigned int n = pix_n[i];
        t_pix_n[i] = n;
        if (n == 0xffffffffu) continue;
        const int64_t pt = (i / HW) * (int64_t)Nmax + n;
        const f3 x = ld3(p, pt), y = ld3(nn, pt), z = ld3(cl, pt);
        float *o = t_old + 10 * i;
        o[0] = x.x; o[1] = x.y; o[2] = x.z; o[3] = y.x; o[4] = y.y; o[5] = y.z; o[6] = z.x; o[7] = z.y; o[8] = z.z; o[9] = cc[pt];
    }
}
// Reverse of the merge for the matched pixels (the formulas of merge_bwd_k; an UNMATCHED map point passes its
// adjoint through unchanged -- x' = (c x + 0) / c -- so nothing is done for it: the pass costs O(pixels), not O(map)):
// the map point's adjoint G is pulled back in place, the frame's adjoints are written (one map point per pixel: plain
// stores), and the arena row gets its pre-merge values back, so that the arena is the map of the previous frame again
// when the reverse pass moves on.
__global__ void fuse_bwd_matched_k(const unsigned int *__restrict__ t_pix_n, const float *__restrict__ t_old, int64_t npix, int HW,
                                   int Nmax, const float *__restrict__ gv, const float *__restrict__ gn, const float *__restrict__ rgb,
                                   const float *__restrict__ alpha, float *__restrict__ p, float *__restrict__ nn,
                                   float *__restrict__ cl, float *__restrict__ cc, float *__restrict__ Gp, float *__restrict__ Gn,
                                   float *__restrict__ Gc, float *__restrict__ Gcc, float *__restrict__ ggv, float *__restrict__ ggn,
                                   float *__restrict__ grgb, float *__restrict__ galpha, const float *__restrict__ depth) {
    for (int64_t pix = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; pix < npix; pix += (int64_t)gridDim.x * blockDim.x) {
        const unsigned int n = t_pix_n[pix];
        if (n == 0xffffffffu) {
            // appended pixels (valid depth, no match) were written by the append's reverse; the rest receive nothing
            if (!(depth[pix] > 0.0f)) { st3(ggv, pix, f3{0, 0, 0}); st3(ggn, pix, f3{0, 0, 0}); st3(grgb, pix, f3{0, 0, 0}); galpha[pix] = 0.0f; }
            continue;
        }
        const int64_t pt = (pix / HW) * (int64_t)Nmax + n;
        const float *o = t_old + 10 * pix;
        const f3 x{o[0], o[1], o[2]}, y{o[3], o[4], o[5]}, z{o[6], o[7], o[8]};
        const float c = o[9], a = alpha[pix];
        const f3 fp = ld3(gv, pix), fn = ld3(gn, pix), fc = ld3(rgb, pix);
        const float c2 = c + a;
        const float inv = 1.0f / (c2 == 0.0f ? 1.0f : c2), dinv = (c2 == 0.0f) ? 0.0f : 1.0f;
        const f3 gx = ld3(Gp, pt), gy = ld3(Gn, pt), gz = ld3(Gc, pt);
        const float gc2 = Gcc[pt];
        const f3 xo{((c * x.x) + (a * fp.x)) * inv, ((c * x.y) + (a * fp.y)) * inv, ((c * x.z) + (a * fp.z)) * inv};
        const f3 yo{((c * y.x) + (a * fn.x)) * inv, ((c * y.y) + (a * fn.y)) * inv, ((c * y.z) + (a * fn.z)) * inv};
        const f3 zo{((c * z.x) + (a * fc.x)) * inv, ((c * z.y) + (a * fc.y)) * inv, ((c * z.z) + (a * fc.z)) * inv};
        const float s_out = (xo.x * gx.x + xo.y * gx.y + xo.z * gx.z) + (yo.x * gy.x + yo.y * gy.y + yo.z * gy.z) +
                            (zo.x * gz.x + zo.y * gz.y + zo.z * gz.z);
        const float s_in = (x.x * gx.x + x.y * gx.y + x.z * gx.z) + (y.x * gy.x + y.y * gy.y + y.z * gy.z) +
                           (z.x * gz.x + z.y * gz.y + z.z * gz.z);
        const float s_f = (fp.x * gx.x + fp.y * gx.y + fp.z * gx.z) + (fn.x * gy.x + fn.y * gy.y + fn.z * gy.z) +
                          (fc.x * gz.x + fc.y * gz.y + fc.z * gz.z);
        st3(Gp, pt, f3{c * inv * gx.x, c * inv * gx.y, c * inv * gx.z});
        st3(Gn, pt, f3{c * inv * gy.x, c * inv * gy.y, c * inv * gy.z});
        st3(Gc, pt, f3{c * inv * gz.x, c * inv * gz.y, c * inv * gz.z});
        Gcc[pt] = gc2 + (s_in - dinv * s_out) * inv;
        st3(ggv, pix, f3{a * inv * gx.x, a * inv * gx.y, a * inv * gx.z});
        st3(ggn, pix, f3{a * inv * gy.x, a * inv * gy.y, a * inv * gy.z});
        st3(grgb, pix, f3{a * inv * gz.x, a * inv * gz.y, a * inv * gz.z});
        galpha[pix] = gc2 + (s_f - dinv * s_out) * inv;
        st3(p, pt, x); st3(nn, pt, y); st3(cl, pt, z); cc[pt] = c;  // the arena row as it was before this frame
    }
}
// appended rows: row base + k of the arena is the k-th unmatched valid pixel in row-major order -- the adjoint of the
// append is the same compaction run backwards (the writer reads where AppendWriter wrote)
struct AppendBwdWriter {
    const float *G[4];
    float *out[4];
    int words[4];
    const int32_t *base;
    __device__ void operator()(int64_t i, int64_t pos) const {
        const int64_t at = (int64_t)(*base) + pos;
#pragma unroll
        for (int a = 0; a < 4; ++a)
            for (int w = 0; w < words[a]; ++w) out[a][(int64_t)words[a] * i + w] = G[a][(int64_t)words[a] * at + w];
    }
};

// ------------------------------------------------------------------ A
__global__ void valid_mask_k(const float *__restrict__ depth, int64_t n, uint8_t *__restrict__ mask) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        mask[i] = depth[i] > 0.0f ? 1 : 0;
}
__global__ void clear_matched_k(const int64_t *__restrict__ rows, const int32_t *__restrict__ d_n, int H, int W,
                                uint8_t *__restrict__ mask) {
    const int n = *d_n;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const longlong4 r = *reinterpret_cast<const longlong4 *>(rows + 4 * i);
        mask[((int64_t)r.x * H + r.z) * W + r.w] = 0;
    }
}

static inline int grid1d(int64_t n) { int g = cdiv(n > 0 ? n : 1, 256); return g > 2048 ? 2048 : g; }

}  // namespace gs

using namespace gs;

extern "C" {

int gs_fusion_similar(const int64_t *rows, const int32_t *d_n_rows, int64_t max_rows, const float *gvertex,
                      const float *gnormal, int H, int W, const float *map_points, const float *map_normals, int Nmax,
                      float dist_th, float dot_th, uint8_t *keep, float *max_dot, gs_stream_t stream) {
    GS_REQUIRE(rows && d_n_rows && gvertex && gnormal && map_points && map_normals && keep, "gs_fusion_similar: NULL argument");
    GS_REQUIRE(H > 0 && W > 0 && Nmax > 0 && max_rows >= 0, "gs_fusion_similar: bad shape");
    hipStream_t st = (hipStream_t)stream;
    if (max_dot) GS_HIP(hipMemsetAsync(max_dot, 0, sizeof(float), st), "gs_fusion_similar/memset");
    if (max_rows == 0) return GS_OK;
    hipLaunchKernelGGL(similar_k, dim3(grid1d(max_rows)), dim3(256), 0, st, rows, d_n_rows, gvertex, gnormal, H, W,
                       map_points, map_normals, Nmax, dist_th, dot_th, keep, max_dot);
    GS_LAUNCH_CHECK("gs_fusion_similar");
    return GS_OK;
}

size_t gs_fusion_unique_ws_bytes(int B, int H, int W) {
    const size_t npix = (size_t)B * H * W;
    return align_up(npix * 8, 256) + align_up(npix * 4, 256) + compact_ws_bytes((int64_t)npix);
}

int gs_fusion_unique(const int64_t *rows, const uint8_t *keep, const int32_t *d_n_rows, int64_t max_rows,
                     const float *gvertex, int B, int H, int W, const float *map_points, const float *map_ccounts,
                     int Nmax, int64_t *out_rows, int32_t *out_count, void *ws, size_t ws_bytes, gs_stream_t stream) {
    GS_REQUIRE(rows && d_n_rows && gvertex && map_points && map_ccounts && out_rows && out_count, "gs_fusion_unique: NULL argument");
    GS_REQUIRE(B > 0 && H > 0 && W > 0 && Nmax > 0 && max_rows >= 0, "gs_fusion_unique: bad shape");
    if (!ws || ws_bytes < gs_fusion_unique_ws_bytes(B, H, W)) {
        set_error("gs_fusion_unique: workspace too small (%zu < %zu)", ws_bytes, gs_fusion_unique_ws_bytes(B, H, W));
        return GS_ERR_WORKSPACE_TOO_SMALL;
    }
    hipStream_t st = (hipStream_t)stream;
    const size_t npix = (size_t)B * H * W;
    unsigned long long *pix_key = (unsigned long long *)ws;
    unsigned int *pix_n = (unsigned int *)((char *)ws + align_up(npix * 8, 256));
    void *cws = (char *)pix_n + align_up(npix * 4, 256);
    GS_HIP(hipMemsetAsync(pix_key, 0xff, npix * 8, st), "gs_fusion_unique/memset");
    GS_HIP(hipMemsetAsync(pix_n, 0xff, npix * 4, st), "gs_fusion_unique/memset");
    if (max_rows > 0) {
        hipLaunchKernelGGL(unique_pass1_k, dim3(grid1d(max_rows)), dim3(256), 0, st, rows, keep, d_n_rows, gvertex, H, W,
                           map_points, map_ccounts, Nmax, pix_key);
        GS_LAUNCH_CHECK("gs_fusion_unique/1");
        hipLaunchKernelGGL(unique_pass2_k, dim3(grid1d(max_rows)), dim3(256), 0, st, rows, keep, d_n_rows, gvertex, H, W,
                           map_points, map_ccounts, Nmax, pix_key, pix_n);
        GS_LAUNCH_CHECK("gs_fusion_unique/2");
    }
    PixPred pred{pix_n};
    PixWriter wr{pix_n, out_rows, H, W};
    return compact_launch((int64_t)npix, pred, wr, out_count, cws, st, "gs_fusion_unique/compact");
}

size_t gs_fusion_merge_ws_bytes(int B, int Nmax) { return align_up((size_t)B * Nmax * 4, 256); }

static int build_match(const int64_t *rows, const int32_t *d_n_rows, int64_t max_rows, int B, int H, int W, int Nmax,
                       void *ws, size_t ws_bytes, hipStream_t st, const char *name) {
    if (!ws || ws_bytes < gs_fusion_merge_ws_bytes(B, Nmax)) {
        set_error("%s: workspace too small", name);
        return GS_ERR_WORKSPACE_TOO_SMALL;
    }
    hipLaunchKernelGGL(fill_i32_k, dim3(grid1d((int64_t)B * Nmax)), dim3(256), 0, st, (int *)ws, (int64_t)B * Nmax, -1);
    GS_LAUNCH_CHECK(name);
    if (max_rows > 0) {
        hipLaunchKernelGGL(scatter_match_k, dim3(grid1d(max_rows)), dim3(256), 0, st, rows, d_n_rows, H, W, Nmax, (int *)ws);
        GS_LAUNCH_CHECK(name);
    }
    return GS_OK;
}

int gs_fusion_merge(const int64_t *rows, const int32_t *d_n_rows, int64_t max_rows, const float *gvertex,
                    const float *gnormal, const float *rgb, const float *alpha, int B, int H, int W, int Nmax,
                    const int32_t *counts, const float *in_points, const float *in_normals, const float *in_colors,
                    const float *in_ccounts, float *out_points, float *out_normals, float *out_colors,
                    float *out_ccounts, void *ws, size_t ws_bytes, gs_stream_t stream) {
    GS_REQUIRE(rows && d_n_rows && gvertex && gnormal && rgb && alpha && counts, "gs_fusion_merge: NULL argument");
    GS_REQUIRE(in_points && in_normals && in_colors && in_ccounts && out_points && out_normals && out_colors && out_ccounts,
               "gs_fusion_merge: NULL map array");
    GS_REQUIRE(B > 0 && B <= 65535 && H > 0 && W > 0 && Nmax > 0 && max_rows >= 0, "gs_fusion_merge: bad shape");
    hipStream_t st = (hipStream_t)stream;
    int rc = build_match(rows, d_n_rows, max_rows, B, H, W, Nmax, ws, ws_bytes, st, "gs_fusion_merge");
    if (rc != GS_OK) return rc;
    hipLaunchKernelGGL(merge_k, dim3(grid1d(Nmax), B), dim3(256), 0, st, (const int *)ws, counts, Nmax, H * W, gvertex,
                       gnormal, rgb, alpha, in_points, in_normals, in_colors, in_ccounts, out_points, out_normals,
                       out_colors, out_ccounts);
    GS_LAUNCH_CHECK("gs_fusion_merge");
    return GS_OK;
}

}  // extern "C"

namespace gs {
// ------------------------------------------------------------------ fused correspondence chain (gs_pointfusion_update)
// find_correspondences (slam/fusionutils.py:549-577 = :247-282 + :381-401 + :489-546) without its tables.  The reference
// materialises the (P,4) table of active rows, filters it into the similar rows, and sorts those to keep one per pixel.
// Inside the update none of the three tables is an output, so ONE pass over the map does, per map point: transform +
// project + in-frame test (project_point: the very function the table-building entry point uses), gather of the frame's
// global vertex / normal at the pixel, distance and angle tests (similar_k's arithmetic), and -- for a similar point -- the
// 64-bit atomicMin of its (1 / (c + 1e-20), squared ray distance) key on the pixel (unique_pass1_k's).  What it leaves
// behind is 4 bytes per map point: the pixel it competes for, or -1.  Pass 2 (ties between equal keys go to the
// smallest point index, like the reference's row sort) and the merge read that word instead of 32-byte rows.
constexpr int CORR_T = 256, CORR_I = 4, CORR_B = CORR_T * CORR_I;
constexpr int FLAG_ANY = 4;  // word of the counter block: 1 if any similar point was found (the merge runs only then); pass 2's first block sets it

// The three map-wide kernels below are written in PHASES over the block's CORR_I items per thread: every phase issues
// its loads for all items before anything consumes them (clamped indices instead of branches around loads).  A thread
// that walks its items one after the other runs 3-4 dependent memory round trips PER ITEM back to back -- measured: 36 us
// for 1.7 M points in that form, i.e. neither the 2.9 TB/s it moved nor its instruction count, but 4 x 4 exposed
// latencies; in phases the same trips overlap fourfold.
__global__ __launch_bounds__(CORR_T) void corr_pass1_k(const float *__restrict__ mp, const float *__restrict__ mn,
                                                       const float *__restrict__ cc, const int32_t *__restrict__ counts, int Nmax,
                                                       const float *__restrict__ poses, const float *__restrict__ Ks, int H, int W,
                                                       float umax, float vmax, const float *__restrict__ gv,
                                                       const float *__restrict__ gn, float dist_th, float dot_th,
                                                       unsigned long long *__restrict__ pix_key, int *__restrict__ pt_pix,
                                                       int32_t *__restrict__ part_active, int32_t *__restrict__ part_similar,
                                                       int32_t *__restrict__ ctr) {
    __shared__ Cam cam;
    __shared__ int red[2][CORR_T / 64];
    const int b = blockIdx.y;
    if (threadIdx.x < 32) {  // the camera: 32 words in ONE round of loads (a single lane doing make_cam alone would chain them)
        const float w = threadIdx.x < 16 ? poses[16 * b + threadIdx.x] : Ks[16 * b + threadIdx.x - 16];
        __shared__ float raw[32];
        raw[threadIdx.x] = w;
        __builtin_amdgcn_wave_barrier();
        if (threadIdx.x == 0) cam = make_cam(raw, raw + 16);
    }
    const int cnt = min(counts[b], Nmax);
    const int64_t HW = (int64_t)H * W;
    const int64_t base = (int64_t)b * Nmax;
    const int n0 = blockIdx.x * CORR_B + threadIdx.x;
    // phase 1: the points
    f3 p[CORR_I];
    bool live[CORR_I];
#pragma unroll
    for (int k = 0; k < CORR_I; ++k) {
        const int n = n0 + k * CORR_T;
        live[k] = n < cnt;
        p[k] = ld3(mp, base + (live[k] ? n : 0));
    }
    __syncthreads();
    // phase 2: projection; then the frame's vertex / normal at the pixel and the point's own normal / confidence
    int px[CORR_I];
    bool act[CORR_I];
#pragma unroll
    for (int k = 0; k < CORR_I; ++k) {
        int h, w;
        act[k] = project_point(cam, p[k], H, W, umax, vmax, h, w) && live[k];
        px[k] = act[k] ? h * W + w : 0;
    }
    f3 fv[CORR_I], fn[CORR_I], q[CORR_I];
    float c[CORR_I];
#pragma unroll
    for (int k = 0; k < CORR_I; ++k) {
        const int64_t pt = base + ((live[k] && act[k]) ? n0 + k * CORR_T : 0), pix = b * HW + px[k];
        fv[k] = ld3(gv, pix); fn[k] = ld3(gn, pix); q[k] = ld3(mn, pt); c[k] = cc[pt];
    }
    // phase 3: tests (similar_k's arithmetic), keys (unique_key's), atomics
    int n_act = 0, n_sim = 0;
    float md = 0.0f;
#pragma unroll
    for (int k = 0; k < CORR_I; ++k) {
        if (!live[k]) continue;
        int out = -1;
        if (act[k]) {
            ++n_act;
            const float dx = fv[k].x - p[k].x, dy = fv[k].y - p[k].y, dz = fv[k].z - p[k].z;
            // (a-b).norm(dim=-1): sqrt(fma(z,z,fma(y,y,x*x)));  (a*b).sum(-1): unfused      (similar_k)
            const float dist = sqrtf(__fmaf_rn(dz, dz, __fmaf_rn(dy, dy, dx * dx)));
            const float dot = (fn[k].x * q[k].x + fn[k].y * q[k].y) + fn[k].z * q[k].z;
            md = fmaxf(md, dot);
            if (dist < dist_th && dot > dot_th) {
                out = px[k];
                ++n_sim;
                // unique_key: 1 / (c + 1e-20) and ((p - fv)**2).sum(-1), unfused
                const float inv_c = 1.0f / (c[k] + 1e-20f);
                const float ex = p[k].x - fv[k].x, ey = p[k].y - fv[k].y, ez = p[k].z - fv[k].z;
                const float ray = (ex * ex + ey * ey) + ez * ez;
                atomicMin(pix_key + b * HW + px[k], ((unsigned long long)fbits(inv_c) << 32) | fbits(ray));
            }
        }
        pt_pix[base + n0 + k * CORR_T] = out;
    }
    n_act = wave_sum_i(n_act);
    n_sim = wave_sum_i(n_sim);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) md = fmaxf(md, __shfl_xor(md, off, kWave));
    if ((threadIdx.x & 63) == 0) {
        red[0][threadIdx.x >> 6] = n_act;
        red[1][threadIdx.x >> 6] = n_sim;
        // the caller only warns when a dot product exceeds 1.001 (un-normalised normals): publish the maximum only when
        // it is above 1, so that the usual case costs no same-address atomics (similar_k)
        if (md > 1.0f) atomicMax(ctr + 3, __float_as_int(md));
    }
    __syncthreads();
    if (threadIdx.x < 2) {  // per-block counts, summed by later launches: no same-address traffic from 25 k waves
        int sum = 0;
        for (int wv = 0; wv < CORR_T / 64; ++wv) sum += red[threadIdx.x][wv];
        (threadIdx.x == 0 ? part_active : part_similar)[blockIdx.y * gridDim.x + blockIdx.x] = sum;
    }
}

// among the similar points whose key IS the pixel's minimum, the smallest point index wins (unique_pass2_k)
__global__ __launch_bounds__(CORR_T) void corr_pass2_k(const int *__restrict__ pt_pix, const int32_t *__restrict__ counts, int Nmax,
                                                       int64_t HW, const float *__restrict__ gv, const float *__restrict__ mp,
                                                       const float *__restrict__ cc, const unsigned long long *__restrict__ pix_key,
                                                       unsigned int *__restrict__ pix_n, const int32_t *__restrict__ part_similar,
                                                       int nparts, int32_t *__restrict__ ctr) {
    if (blockIdx.x == 0 && blockIdx.y == 0) {  // "any similar point at all?" for the merge (the launch after this one)
        int any = 0;
        for (int i = threadIdx.x; i < nparts; i += CORR_T) any |= part_similar[i];
        if (__any(any != 0) && (threadIdx.x & 63) == 0) ctr[FLAG_ANY] = 1;
    }
    const int b = blockIdx.y;
    const int cnt = min(counts[b], Nmax);
    const int64_t base = (int64_t)b * Nmax;
    const int n0 = blockIdx.x * CORR_B + threadIdx.x;
    int px[CORR_I];
#pragma unroll
    for (int k = 0; k < CORR_I; ++k) {
        const int n = n0 + k * CORR_T;
        px[k] = n < cnt ? pt_pix[base + n] : -1;
    }
    f3 fv[CORR_I], p[CORR_I];
    float c[CORR_I];
    unsigned long long kmin[CORR_I];
#pragma unroll
    for (int k = 0; k < CORR_I; ++k) {
        const bool sim = px[k] >= 0;
        const int64_t pt = base + (sim ? n0 + k * CORR_T : 0), pix = b * HW + (sim ? px[k] : 0);
        fv[k] = ld3(gv, pix); p[k] = ld3(mp, pt); c[k] = cc[pt]; kmin[k] = pix_key[pix];
    }
#pragma unroll
    for (int k = 0; k < CORR_I; ++k) {
        if (px[k] < 0) continue;
        const float inv_c = 1.0f / (c[k] + 1e-20f);
        const float ex = p[k].x - fv[k].x, ey = p[k].y - fv[k].y, ez = p[k].z - fv[k].z;
        const float ray = (ex * ex + ey * ey) + ez * ez;
        if ((((unsigned long long)fbits(inv_c) << 32) | fbits(ray)) == kmin[k]) atomicMin(pix_n + b * HW + px[k], (unsigned int)(n0 + k * CORR_T));
    }
}

// merge_inplace_k with the match read from (pt_pix, pix_n): point n merges with its pixel iff it is that pixel's winner.
// Also counts the winners (= unique correspondences) per block, for the update's statistics.
__global__ __launch_bounds__(CORR_T) void merge_corr_k(const int *__restrict__ pt_pix, const unsigned int *__restrict__ pix_n,
                                                       const int32_t *__restrict__ counts, const int32_t *__restrict__ ctr, int Nmax,
                                                       int HW, const float *__restrict__ gv, const float *__restrict__ gn,
                                                       const float *__restrict__ rgb, const float *__restrict__ alpha, float *p,
                                                       float *nn, float *cl, float *cc, int32_t *__restrict__ part_u) {
    __shared__ int red[CORR_T / 64];
    int n_u = 0;
    if (ctr[FLAG_ANY] != 0) {  // no correspondence at all: fuse_with_map skips the merge (fusionutils.py:654)
        const int b = blockIdx.y;
        const int cnt = min(counts[b], Nmax);
        const int64_t base = (int64_t)b * Nmax, pbase = (int64_t)b * HW;
        const int n0 = blockIdx.x * CORR_B + threadIdx.x;
        // phase 1: the point's row and the pixel it competes for
        f3 x[CORR_I], y[CORR_I], z[CORR_I];
        float c[CORR_I];
        int px[CORR_I];
        bool live[CORR_I];
#pragma unroll
        for (int k = 0; k < CORR_I; ++k) {
            const int n = n0 + k * CORR_T;
            live[k] = n < cnt;
            const int64_t pt = base + (live[k] ? n : 0);
            px[k] = pt_pix[pt];
            x[k] = ld3(p, pt); y[k] = ld3(nn, pt); z[k] = ld3(cl, pt); c[k] = cc[pt];
        }
        // phase 2: is it the pixel's winner?  (then the frame's values at the pixel)
        unsigned int win[CORR_I];
#pragma unroll
        for (int k = 0; k < CORR_I; ++k) {
            if (!live[k]) px[k] = -1;
            win[k] = pix_n[pbase + (px[k] >= 0 ? px[k] : 0)];
        }
        f3 fp[CORR_I], fn[CORR_I], fc[CORR_I];
        float a[CORR_I];
#pragma unroll
        for (int k = 0; k < CORR_I; ++k) {
            const bool m = px[k] >= 0 && win[k] == (unsigned int)(n0 + k * CORR_T);
            if (!m) px[k] = -1;
            const int64_t pix = pbase + (m ? px[k] : 0);
            a[k] = alpha[pix]; fp[k] = ld3(gv, pix); fn[k] = ld3(gn, pix); fc[k] = ld3(rgb, pix);
        }
#pragma unroll
        for (int k = 0; k < CORR_I; ++k) {
            if (!live[k]) continue;
            const bool m = px[k] >= 0;
            n_u += m ? 1 : 0;
            const float ak = m ? a[k] : 0.0f;
            const f3 vp = m ? fp[k] : f3{0, 0, 0}, vn = m ? fn[k] : f3{0, 0, 0}, vc = m ? fc[k] : f3{0, 0, 0};
            const float c2 = c[k] + ak;
            const float inv = 1.0f / (c2 == 0.0f ? 1.0f : c2);
            const int64_t pt = base + n0 + k * CORR_T;
            st3(p, pt, f3{((c[k] * x[k].x) + (ak * vp.x)) * inv, ((c[k] * x[k].y) + (ak * vp.y)) * inv, ((c[k] * x[k].z) + (ak * vp.z)) * inv});
            st3(nn, pt, f3{((c[k] * y[k].x) + (ak * vn.x)) * inv, ((c[k] * y[k].y) + (ak * vn.y)) * inv, ((c[k] * y[k].z) + (ak * vn.z)) * inv});
            st3(cl, pt, f3{((c[k] * z[k].x) + (ak * vc.x)) * inv, ((c[k] * z[k].y) + (ak * vc.y)) * inv, ((c[k] * z[k].z) + (ak * vc.z)) * inv});
            cc[pt] = c2;
        }
    }
    n_u = wave_sum_i(n_u);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = n_u;
    __syncthreads();
    if (threadIdx.x == 0) {
        int sum = 0;
        for (int wv = 0; wv < CORR_T / 64; ++wv) sum += red[wv];
        part_u[blockIdx.y * gridDim.x + blockIdx.x] = sum;
    }
}

// the update's last launch: the per-block counts of the two passes summed (fixed order), every sequence's row count
// advanced by what its append pass selected (append_count_k for all b at once), the statistics row written
__global__ __launch_bounds__(1024) void fuse_finish_k(const int32_t *__restrict__ part_active, const int32_t *__restrict__ part_u,
                                                      int nparts, int32_t *__restrict__ ctr, int32_t *__restrict__ counts,
                                                      const int *__restrict__ totals, int B, int cap, int32_t *__restrict__ appended,
                                                      int32_t *__restrict__ stats) {
    __shared__ int red[2][16];
    int a = 0, u = 0;
    for (int i = threadIdx.x; i < nparts; i += 1024) { a += part_active[i]; u += part_u[i]; }
    a = wave_sum_i(a); u = wave_sum_i(u);
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = a; red[1][threadIdx.x >> 6] = u; }
    if ((int)threadIdx.x < B) {
        const int64_t want = (int64_t)counts[threadIdx.x] + totals[threadIdx.x];
        const int now = (int)(want > cap ? cap : want);
        appended[threadIdx.x] = now - counts[threadIdx.x];
        if (want > cap) ctr[2] = 1;
        counts[threadIdx.x] = now;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int sa = 0, su = 0;
        for (int wv = 0; wv < 16; ++wv) { sa += red[0][wv]; su += red[1][wv]; }
        ctr[0] = sa; ctr[1] = su;
        if (stats) { stats[0] = sa; stats[1] = su; stats[2] = ctr[2]; stats[3] = ctr[3]; }
    }
    if (stats && (int)threadIdx.x < B) stats[4 + threadIdx.x] = appended[threadIdx.x];
}

// ---- the fused chain's host side (slam.hip drives it).  state = [pix_key (npix x 8) | pix_n (npix x 4) | pt_pix
// (B*Nmax x 4) | part_active, part_u (blocks x 4 each)]; pix_key / pix_n are initialised by the maps kernel (VnExtra).
static inline int corr_blocks(int Nmax) { return cdiv(Nmax > 0 ? Nmax : 1, CORR_B); }
size_t fusion_corr_state_bytes(int B, int H, int W, int Nmax) {
    const size_t npix = (size_t)B * H * W;
    return align_up(npix * 8, 256) + align_up(npix * 4, 256) + align_up((size_t)B * Nmax * 4, 256) +
           3 * align_up((size_t)B * corr_blocks(Nmax) * 4, 256);
}
struct CorrState {
    unsigned long long *pix_key;
    unsigned int *pix_n;
    int *pt_pix;
    int32_t *part_active, *part_similar, *part_u;
};
static inline CorrState corr_state_ptrs(void *state, int B, int H, int W, int Nmax) {
    const size_t npix = (size_t)B * H * W;
    char *p = (char *)state;
    CorrState c;
    c.pix_key = (unsigned long long *)p; p += align_up(npix * 8, 256);
    c.pix_n = (unsigned int *)p; p += align_up(npix * 4, 256);
    c.pt_pix = (int *)p; p += align_up((size_t)B * Nmax * 4, 256);
    c.part_active = (int32_t *)p; p += align_up((size_t)B * corr_blocks(Nmax) * 4, 256);
    c.part_similar = (int32_t *)p; p += align_up((size_t)B * corr_blocks(Nmax) * 4, 256);
    c.part_u = (int32_t *)p;
    return c;
}
void fusion_corr_init_ptrs(void *state, int B, int H, int W, int Nmax, unsigned long long **pix_key, unsigned int **pix_n) {
    const CorrState c = corr_state_ptrs(state, B, H, W, Nmax);
    *pix_key = c.pix_key; *pix_n = c.pix_n;
}
// passes 1 and 2: afterwards pix_n holds every pixel's winner (or ~0) and pt_pix every map point's pixel (or -1)
int fusion_correspond(void *state, const float *map_points, const float *map_normals, const float *map_ccounts, const int32_t *counts,
                      int B, int Nmax, const float *poses, const float *intrinsics, int H, int W, const float *gvertex,
                      const float *gnormal, float dist_th, float dot_th, int32_t *ctr, hipStream_t st) {
    const CorrState c = corr_state_ptrs(state, B, H, W, Nmax);
    const float umax = (float)((double)W - 0.999), vmax = (float)((double)H - 0.999);
    const dim3 grid(corr_blocks(Nmax), B);
    hipLaunchKernelGGL(corr_pass1_k, grid, dim3(CORR_T), 0, st, map_points, map_normals, map_ccounts, counts, Nmax, poses, intrinsics, H, W,
                       umax, vmax, gvertex, gnormal, dist_th, dot_th, c.pix_key, c.pt_pix, c.part_active, c.part_similar, ctr);
    hipLaunchKernelGGL(corr_pass2_k, grid, dim3(CORR_T), 0, st, (const int *)c.pt_pix, counts, Nmax, (int64_t)H * W, gvertex, map_points,
                       map_ccounts, (const unsigned long long *)c.pix_key, c.pix_n, (const int32_t *)c.part_similar, B * corr_blocks(Nmax),
                       ctr);
    GS_LAUNCH_CHECK("gs_pointfusion_update/correspond");
    return GS_OK;
}
int fusion_merge_corr(void *state, const int32_t *ctr, const float *gvertex, const float *gnormal, const float *rgb, const float *alpha,
                      int B, int H, int W, int Nmax, const int32_t *counts, float *points, float *normals, float *colors,
                      float *ccounts, hipStream_t st) {
    const CorrState c = corr_state_ptrs(state, B, H, W, Nmax);
    hipLaunchKernelGGL(merge_corr_k, dim3(corr_blocks(Nmax), B), dim3(CORR_T), 0, st, (const int *)c.pt_pix, (const unsigned int *)c.pix_n,
                       counts, ctr, Nmax, H * W, gvertex, gnormal, rgb, alpha, points, normals, colors, ccounts, c.part_u);
    GS_LAUNCH_CHECK("gs_pointfusion_update/merge");
    return GS_OK;
}
// unmatched valid pixels of batch element b appended behind the rows its arena holds; the row count itself advances
// in fusion_finish (AppendWriter reads the count as the append offset, so it must not move before every b is written)
int fusion_append_corr(void *state, int B, int H, int W, int Nmax, int b, const float *depth, const float *const *h_src,
                       const int *h_row_floats, float *const *h_dst, const int32_t *d_count, int cap, int *d_total, void *cws,
                       hipStream_t st) {
    const CorrState c = corr_state_ptrs(state, B, H, W, Nmax);
    const int64_t HW = (int64_t)H * W;
    AppendWriter wr;
    wr.n_arrays = 4; wr.base = d_count; wr.cap = cap;
    for (int a = 0; a < 4; ++a) {
        wr.src[a] = (const uint32_t *)h_src[a]; wr.out[a] = (uint32_t *)h_dst[a]; wr.words[a] = h_row_floats[a];
    }
    AppendPredPix pred{depth + b * HW, c.pix_n + b * HW};
    return compact_launch(HW, pred, wr, d_total, cws, st, "gs_pointfusion_update/append");
}
int fusion_finish(void *state, int B, int H, int W, int Nmax, int32_t *ctr, int32_t *counts, const int *totals, int cap,
                  int32_t *appended, int32_t *stats, hipStream_t st) {
    const CorrState c = corr_state_ptrs(state, B, H, W, Nmax);
    hipLaunchKernelGGL(fuse_finish_k, dim3(1), dim3(1024), 0, st, (const int32_t *)c.part_active, (const int32_t *)c.part_u,
                       B * corr_blocks(Nmax), ctr, counts, totals, B, cap, appended, stats);
    GS_LAUNCH_CHECK("gs_pointfusion_update/finish");
    return GS_OK;
}
const unsigned int *fusion_corr_pix_n(const void *state, int B, int H, int W, int Nmax) {
    return corr_state_ptrs((void *)state, B, H, W, Nmax).pix_n;
}
size_t fusion_tape_bytes(int B, int H, int W) {
    const size_t npix = (size_t)B * H * W;
    return 2 * align_up((size_t)B * 4, 256) + align_up(npix * 4, 256) + align_up(npix * 40, 256);
}
struct FusionTape {
    int32_t *n_before, *appended;
    unsigned int *pix_n;
    float *old;
};
static inline FusionTape fusion_tape_ptrs(void *tape, int B, int H, int W) {
    const size_t npix = (size_t)B * H * W;
    char *p = (char *)tape;
    FusionTape t;
    t.n_before = (int32_t *)p; p += align_up((size_t)B * 4, 256);
    t.appended = (int32_t *)p; p += align_up((size_t)B * 4, 256);
    t.pix_n = (unsigned int *)p; p += align_up(npix * 4, 256);
    t.old = (float *)p;
    return t;
}
// after fusion_correspond, before the merge: winners, pre-merge values and the row counts before the append
int fusion_tape_record(const void *state, void *tape, int B, int H, int W, int Nmax, const int32_t *counts, const float *points,
                       const float *normals, const float *colors, const float *ccounts, hipStream_t st) {
    const unsigned int *pix_n = fusion_corr_pix_n(state, B, H, W, Nmax);
    const FusionTape t = fusion_tape_ptrs(tape, B, H, W);
    const int64_t npix = (int64_t)B * H * W;
    hipLaunchKernelGGL(tape_old_k, dim3(grid1d(npix)), dim3(256), 0, st, pix_n, npix, H * W, Nmax, points, normals,
                       colors, ccounts, t.pix_n, t.old);
    GS_LAUNCH_CHECK("gs_pointfusion_update_taped/record");
    GS_HIP(hipMemcpyAsync(t.n_before, counts, (size_t)B * 4, hipMemcpyDeviceToDevice, st), "gs_pointfusion_update_taped/counts");
    return GS_OK;
}
int fusion_tape_appended(void *tape, int B, int H, int W, const int32_t *appended, hipStream_t st) {
    const FusionTape t = fusion_tape_ptrs(tape, B, H, W);
    GS_HIP(hipMemcpyAsync(t.appended, appended, (size_t)B * 4, hipMemcpyDeviceToDevice, st), "gs_pointfusion_update_taped/appended");
    return GS_OK;
}
// reverse of merge + append for one frame (see fuse_bwd_matched_k): frame adjoints fully written, G pulled back in
// place, arena rows and counts restored to the previous frame's
int fusion_update_reverse(const void *tape, int B, int H, int W, int Nmax, const float *depth, const float *gvertex,
                          const float *gnormal, const float *rgb, const float *alpha, float *points, float *normals, float *colors,
                          float *ccounts, int32_t *counts, float *Gp, float *Gn, float *Gc, float *Gcc, float *g_gvertex,
                          float *g_gnormal, float *g_rgb, float *g_alpha, void *cws, hipStream_t st) {
    const char *name = "gs_pointfusion_update_backward";
    const FusionTape t = fusion_tape_ptrs((void *)tape, B, H, W);
    const int64_t npix = (int64_t)B * H * W, HW = (int64_t)H * W;
    // every pixel is written exactly once: appended ones here, matched and invalid ones by fuse_bwd_matched_k
    for (int b = 0; b < B; ++b) {  // appended pixels: their rows sit behind n_before[b]
        AppendBwdWriter wr;
        const float *G[4] = {Gp + (size_t)b * Nmax * 3, Gn + (size_t)b * Nmax * 3, Gc + (size_t)b * Nmax * 3, Gcc + (size_t)b * Nmax};
        float *out[4] = {g_gvertex + b * HW * 3, g_gnormal + b * HW * 3, g_rgb + b * HW * 3, g_alpha + b * HW};
        const int words[4] = {3, 3, 3, 1};
        for (int a = 0; a < 4; ++a) { wr.G[a] = G[a]; wr.out[a] = out[a]; wr.words[a] = words[a]; }
        wr.base = t.n_before + b;
        int *total = (int *)((char *)cws + compact_ws_bytes(HW));
        AppendPredPix pred{depth + b * HW, t.pix_n + b * HW};
        const int rc = compact_launch(HW, pred, wr, total, cws, st, name);
        if (rc) return rc;
    }
    hipLaunchKernelGGL(fuse_bwd_matched_k, dim3(grid1d(npix)), dim3(256), 0, st, (const unsigned int *)t.pix_n, (const float *)t.old, npix,
                       H * W, Nmax, gvertex, gnormal, rgb, alpha, points, normals, colors, ccounts, Gp, Gn, Gc, Gcc, g_gvertex, g_gnormal,
                       g_rgb, g_alpha, depth);
    GS_LAUNCH_CHECK(name);
    GS_HIP(hipMemcpyAsync(counts, t.n_before, (size_t)B * 4, hipMemcpyDeviceToDevice, st), name);
    return GS_OK;
}
// valid pixels of batch element b (n_arrays row arrays) appended behind the rows its arena already holds
int append_valid_pixels(int n_arrays, const float *depth_b, int64_t HW, const float *const *h_src, const int *h_row_floats,
                        float *const *h_dst, int32_t *d_count, int cap, int32_t *d_appended, int32_t *d_overflow, void *cws,
                        hipStream_t st) {
    AppendWriter wr;
    wr.n_arrays = n_arrays; wr.base = d_count; wr.cap = cap;
    for (int a = 0; a < 4; ++a) {
        wr.src[a] = a < n_arrays ? (const uint32_t *)h_src[a] : nullptr;
        wr.out[a] = a < n_arrays ? (uint32_t *)h_dst[a] : nullptr;
        wr.words[a] = a < n_arrays ? h_row_floats[a] : 0;
    }
    int *total = (int *)((char *)cws + compact_ws_bytes(HW));
    ValidDepthPred pred{depth_b};
    const int rc = compact_launch(HW, pred, wr, total, cws, st, "gs_aggregate_update/append");
    if (rc) return rc;
    hipLaunchKernelGGL(append_count_k, dim3(1), dim3(64), 0, st, d_count, total, cap, d_appended, d_overflow);
    GS_LAUNCH_CHECK("gs_aggregate_update/append count");
    return GS_OK;
}
}  // namespace gs

extern "C" {

size_t gs_fusion_merge_inplace_ws_bytes(int B, int Nmax) { return gs_fusion_merge_ws_bytes(B, Nmax); }

int gs_fusion_merge_inplace(const int64_t *rows, const int32_t *d_n_rows, int64_t max_rows, const float *gvertex,
                            const float *gnormal, const float *rgb, const float *alpha, int B, int H, int W, int Nmax,
                            const int32_t *counts, float *points, float *normals, float *colors, float *ccounts, void *ws,
                            size_t ws_bytes, gs_stream_t stream) {
    GS_REQUIRE(rows && d_n_rows && gvertex && gnormal && rgb && alpha && counts && points && normals && colors && ccounts,
               "gs_fusion_merge_inplace: NULL argument");
    GS_REQUIRE(B > 0 && B <= 65535 && H > 0 && W > 0 && Nmax > 0 && max_rows >= 0, "gs_fusion_merge_inplace: bad shape");
    hipStream_t st = (hipStream_t)stream;
    int rc = build_match(rows, d_n_rows, max_rows, B, H, W, Nmax, ws, ws_bytes, st, "gs_fusion_merge_inplace");
    if (rc != GS_OK) return rc;
    hipLaunchKernelGGL(merge_inplace_k, dim3(grid1d(Nmax), B), dim3(256), 0, st, (const int *)ws, counts, d_n_rows, Nmax, H * W,
                       gvertex, gnormal, rgb, alpha, points, normals, colors, ccounts);
    GS_LAUNCH_CHECK("gs_fusion_merge_inplace");
    return GS_OK;
}

int gs_fusion_merge_backward(const int64_t *rows, const int32_t *d_n_rows, int64_t max_rows, const float *gvertex,
                             const float *gnormal, const float *rgb, const float *alpha, int B, int H, int W, int Nmax,
                             const int32_t *counts, const float *in_points, const float *in_normals,
                             const float *in_colors, const float *in_ccounts, const float *g_out_points,
                             const float *g_out_normals, const float *g_out_colors, const float *g_out_ccounts,
                             float *g_in_points, float *g_in_normals, float *g_in_colors, float *g_in_ccounts,
                             float *g_gvertex, float *g_gnormal, float *g_rgb, float *g_alpha, void *ws, size_t ws_bytes,
                             gs_stream_t stream) {
    GS_REQUIRE(rows && d_n_rows && gvertex && gnormal && rgb && alpha && counts, "gs_fusion_merge_backward: NULL argument");
    GS_REQUIRE(in_points && in_normals && in_colors && in_ccounts, "gs_fusion_merge_backward: NULL map array");
    GS_REQUIRE(B > 0 && B <= 65535 && H > 0 && W > 0 && Nmax > 0 && max_rows >= 0, "gs_fusion_merge_backward: bad shape");
    hipStream_t st = (hipStream_t)stream;
    int rc = build_match(rows, d_n_rows, max_rows, B, H, W, Nmax, ws, ws_bytes, st, "gs_fusion_merge_backward");
    if (rc != GS_OK) return rc;
    hipLaunchKernelGGL(merge_bwd_k, dim3(grid1d(Nmax), B), dim3(256), 0, st, (const int *)ws, counts, Nmax, H * W, gvertex,
                       gnormal, rgb, alpha, in_points, in_normals, in_colors, in_ccounts, g_out_points, g_out_normals,
                       g_out_colors, g_out_ccounts, g_in_points, g_in_normals, g_in_colors, g_in_ccounts, g_gvertex,
                       g_gnormal, g_rgb, g_alpha);
    GS_LAUNCH_CHECK("gs_fusion_merge_backward");
    return GS_OK;
}

int gs_fusion_new_mask(const float *depth, const int64_t *rows, const int32_t *d_n_rows, int64_t max_rows, int B, int H,
                       int W, uint8_t *mask, gs_stream_t stream) {
    GS_REQUIRE(depth && mask && B > 0 && H > 0 && W > 0 && max_rows >= 0, "gs_fusion_new_mask: bad arguments");
    GS_REQUIRE(max_rows == 0 || (rows && d_n_rows), "gs_fusion_new_mask: rows/d_n_rows NULL with max_rows > 0");
    hipStream_t st = (hipStream_t)stream;
    const int64_t n = (int64_t)B * H * W;
    hipLaunchKernelGGL(valid_mask_k, dim3(grid1d(n)), dim3(256), 0, st, depth, n, mask);
    GS_LAUNCH_CHECK("gs_fusion_new_mask/valid");
    if (max_rows > 0) {
        hipLaunchKernelGGL(clear_matched_k, dim3(grid1d(max_rows)), dim3(256), 0, st, rows, d_n_rows, H, W, mask);
        GS_LAUNCH_CHECK("gs_fusion_new_mask/clear");
    }
    return GS_OK;
}

}  // extern "C"
