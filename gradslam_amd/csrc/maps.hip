// maps.hip -- depth -> vertex / normal maps (local + global) in one pass, its adjoint,
// get_alpha, the frame-side downsample and the generic stable row compaction.
//
// HBM-bound stream: 4 B of depth in, up to 48 B of maps out per pixel.  One 64x4 pixel tile per
// 256-thread workgroup; the (64+1)x(4+1) depth tile the forward-difference stencil needs is staged
// through LDS once, so each depth value is fetched from HBM exactly once per tile.
#include <stdarg.h>

#include "gs_common.hpp"
#include "gs_compact.hpp"
#include "gs_project.hpp"

namespace gs {

// ------------------------------------------------------------------ error slot (one per thread)
static thread_local char g_err[512] = "";
void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

constexpr int TW = 64, TH = 4;

struct Kinv4 {
    float a, c, e, f;  // x = a*w + c ; y = e*h + f   (reference geometry/projutils.py:444-449)
};

__device__ __forceinline__ Kinv4 load_kinv(const float *__restrict__ K) {
    const float eps = 1e-6f;
    const float fx = K[0] + eps, fy = K[5] + eps;
    return Kinv4{1.0f / fx, (-1.0f * K[2]) / fx, 1.0f / fy, (-1.0f * K[6]) / fy};
}

// local vertex of pixel (h, w) with depth d:  (Kinv . [w,h,1]) * d * (d > 0)
// the reference's einsum contracts [a,0,c].[w,h,1] as fma(c,1,fma(0,h,a*w)) == a*w + c
__device__ __forceinline__ f3 vertex_of(const Kinv4 k, int h, int w, float d) {
    const float m = d > 0.0f ? 1.0f : 0.0f;
    const float x = k.a * (float)w + k.c;
    const float y = k.e * (float)h + k.f;
    return f3{(x * d) * m, (y * d) * m, (1.0f * d) * m};
}

// What the PointFusion update lets ride on this pass over the pixels (slam.hip, fusion.hip; all optional): the sample
// confidence alpha = get_alpha(local vertex) (slam/fusionutils.py:69-73, the arithmetic of alpha_k below), the
// "no candidate / no winner" initialisation of the correspondence stage's per-pixel state, and the zeroing of its counter
// block -- three launches (alpha_k, two memsets) that become stores of a kernel that visits every pixel anyway.
struct VnExtra {
    float *alpha;                  // (B*L*H*W) or NULL
    float alpha_den, alpha_eps;    // 2 sigma^2, lower clamp
    unsigned long long *pix_key;   // (B*L*H*W) or NULL: set to ~0
    unsigned int *pix_n;           // (B*L*H*W) or NULL: set to ~0
    int32_t *zero;                 // n_zero words zeroed by the first block, or NULL
    int n_zero;
    float *cam_out;                // (B*L, 32) or NULL: a copy of every frame's pose | intrinsics at an address of the CALLEE's
                                   // choosing (gs_slam_localize: its workspace -- what a captured graph of the ICP loops may bake in)
};

__global__ __launch_bounds__(TW *TH) void vertex_normal_k(const float *__restrict__ depth, const float *__restrict__ Ks,
                                                          const float *__restrict__ poses, int L, int H, int W,
                                                          float *__restrict__ vertex, float *__restrict__ normal,
                                                          float *__restrict__ gvertex, float *__restrict__ gnormal, VnExtra ex) {
    __shared__ float sd[TH + 1][TW + 1];
    if (ex.zero && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0)
        for (int i = threadIdx.x; i < ex.n_zero; i += TW * TH) ex.zero[i] = 0;
    if (ex.cam_out && poses && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x < 32) {
        const int blz = blockIdx.z;
        ex.cam_out[32 * blz + threadIdx.x] = threadIdx.x < 16 ? poses[16 * (int64_t)blz + threadIdx.x] : Ks[16 * (blz / L) + threadIdx.x - 16];
    }
    const int bl = blockIdx.z;  // b*L + l
    const int b = bl / L;
    const int w0 = blockIdx.x * TW, h0 = blockIdx.y * TH;
    const int tx = threadIdx.x & (TW - 1), ty = threadIdx.x / TW;
    const float *dimg = depth + (int64_t)bl * H * W;

    // stage the depth tile (+1 halo column/row, clamped at the image edge)
    for (int i = threadIdx.x; i < (TH + 1) * (TW + 1); i += TW * TH) {
        const int r = i / (TW + 1), c = i - r * (TW + 1);
        const int hh = min(h0 + r, H - 1), ww = min(w0 + c, W - 1);
        sd[r][c] = dimg[(int64_t)hh * W + ww];
    }
    __syncthreads();

    const int h = h0 + ty, w = w0 + tx;
    if (h >= H || w >= W) return;
    const Kinv4 k = load_kinv(Ks + 16 * b);
    const float d = sd[ty][tx];
    const f3 v = vertex_of(k, h, w, d);

    // forward differences; the last column / row re-use the previous difference
    // (reference structures/rgbdimages.py:724-731).  Neighbour validity is NOT checked.
    f3 dh, dv;
    if (w < W - 1) {
        const f3 vr = vertex_of(k, h, w + 1, sd[ty][tx + 1]);
        dh = f3{vr.x - v.x, vr.y - v.y, vr.z - v.z};
    } else {  // w == W-1: V(h,W-1) - V(h,W-2); W-2 may sit in the previous tile -> global read
        const float dl = (tx > 0) ? sd[ty][tx - 1] : dimg[(int64_t)h * W + (w - 1)];
        const f3 vl = vertex_of(k, h, w - 1, dl);
        dh = f3{v.x - vl.x, v.y - vl.y, v.z - vl.z};
    }
    if (h < H - 1) {
        const f3 vd = vertex_of(k, h + 1, w, sd[ty + 1][tx]);
        dv = f3{vd.x - v.x, vd.y - v.y, vd.z - v.z};
    } else {
        const float du = (ty > 0) ? sd[ty - 1][tx] : dimg[(int64_t)(h - 1) * W + w];
        const f3 vu = vertex_of(k, h - 1, w, du);
        dv = f3{v.x - vu.x, v.y - vu.y, v.z - vu.z};
    }
    // torch.cross contracts each component as fma(a1, b2, -(a2*b1))
    f3 n;
    n.x = __fmaf_rn(dh.y, dv.z, -(dh.z * dv.y));
    n.y = __fmaf_rn(dh.z, dv.x, -(dh.x * dv.z));
    n.z = __fmaf_rn(dh.x, dv.y, -(dh.y * dv.x));
    // .norm(dim) contracts as sqrt(fma(z,z,fma(y,y,x*x)))
    float nn = sqrtf(__fmaf_rn(n.z, n.z, __fmaf_rn(n.y, n.y, n.x * n.x)));
    nn = (nn == 0.0f) ? 1.0f : nn;
    const float m = d > 0.0f ? 1.0f : 0.0f;
    n = f3{(n.x / nn) * m, (n.y / nn) * m, (n.z / nn) * m};

    const int64_t pix = (int64_t)bl * H * W + (int64_t)h * W + w;
    if (ex.alpha) {  // alpha_k's arithmetic on the local vertex
        const float ss = (v.x * v.x + v.y * v.y) + v.z * v.z;
        ex.alpha[pix] = fminf(fmaxf(expf((-ss) / ex.alpha_den), ex.alpha_eps), 1.01f);
    }
    if (ex.pix_key) ex.pix_key[pix] = ~0ull;
    if (ex.pix_n) ex.pix_n[pix] = ~0u;
    if (vertex) st3(vertex, pix, v);
    if (normal) st3(normal, pix, n);
    if (gvertex || gnormal) {
        if (poses) {
            const float *T = poses + 16 * (int64_t)bl;
            if (gvertex) {
                f3 g = xform(T, v);
                st3(gvertex, pix, f3{g.x * m, g.y * m, g.z * m});
            }
            if (gnormal) {
                st3(gnormal, pix,
                    f3{dot3_fma(T[0], T[1], T[2], n.x, n.y, n.z), dot3_fma(T[4], T[5], T[6], n.x, n.y, n.z),
                       dot3_fma(T[8], T[9], T[10], n.x, n.y, n.z)});
            }
        } else {
            if (gvertex) st3(gvertex, pix, v);
            if (gnormal) st3(gnormal, pix, n);
        }
    }
}

// ------------------------------------------------------------------ adjoint
// pass 1: per pixel, adjoint of the normal w.r.t. its two difference vectors -> ws (dh_bar, dv_bar);
//         also accumulates d/dR from the global normal map.
// pass 2: per pixel, gather the stencil adjoints, add the vertex adjoints, reduce to depth / K / pose.
constexpr int VN_PART = 25;  // pass 2: R_bar(9) t_bar(3) a,c,e,f_bar(4) ; pass 1: R_bar(9) from the normals

__global__ __launch_bounds__(256) void vn_bwd_pass1_k(const float *__restrict__ depth, const float *__restrict__ Ks,
                                                      const float *__restrict__ poses, int L, int H, int W,
                                                      const float *__restrict__ g_normal,
                                                      const float *__restrict__ g_gnormal,
                                                      float *__restrict__ dhb, float *__restrict__ dvb,
                                                      float *__restrict__ part /* [bl][block][VN_PART] */) {
    const int bl = blockIdx.y;
    const int b = bl / L;
    const int64_t HW = (int64_t)H * W;
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    float rbar[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (p < HW) {
        const int h = (int)(p / W), w = (int)(p - (int64_t)h * W);
        const float *dimg = depth + (int64_t)bl * HW;
        const Kinv4 k = load_kinv(Ks + 16 * b);
        const float d = dimg[p];
        const float m = d > 0.0f ? 1.0f : 0.0f;
        const int wl = (w < W - 1) ? w : w - 1, hl = (h < H - 1) ? h : h - 1;
        const f3 a0 = vertex_of(k, h, wl, dimg[(int64_t)h * W + wl]);
        const f3 a1 = vertex_of(k, h, wl + 1, dimg[(int64_t)h * W + wl + 1]);
        const f3 b0 = vertex_of(k, hl, w, dimg[(int64_t)hl * W + w]);
        const f3 b1 = vertex_of(k, hl + 1, w, dimg[(int64_t)(hl + 1) * W + w]);
        const f3 dh{a1.x - a0.x, a1.y - a0.y, a1.z - a0.z}, dv{b1.x - b0.x, b1.y - b0.y, b1.z - b0.z};
        // same contractions as the forward pass, so degenerate pixels (both neighbours invalid ->
        // dh == dv, cross = fma residue) take the same branch and see the same tiny norm
        f3 c{__fmaf_rn(dh.y, dv.z, -(dh.z * dv.y)), __fmaf_rn(dh.z, dv.x, -(dh.x * dv.z)),
             __fmaf_rn(dh.x, dv.y, -(dh.y * dv.x))};
        const float s = sqrtf(__fmaf_rn(c.z, c.z, __fmaf_rn(c.y, c.y, c.x * c.x)));
        const float sdiv = (s == 0.0f) ? 1.0f : s;
        const f3 nh{c.x / sdiv, c.y / sdiv, c.z / sdiv};
        // total adjoint of the masked local normal N = m * nh
        f3 nb{0, 0, 0};
        const int64_t pix = (int64_t)bl * HW + p;
        if (g_normal) {
            const f3 t = ld3(g_normal, pix);
            nb = t;
        }
        if (g_gnormal) {
            const f3 t = ld3(g_gnormal, pix);
            if (poses) {
                const float *T = poses + 16 * (int64_t)bl;
                nb.x += T[0] * t.x + T[4] * t.y + T[8] * t.z;  // R^T t
                nb.y += T[1] * t.x + T[5] * t.y + T[9] * t.z;
                nb.z += T[2] * t.x + T[6] * t.y + T[10] * t.z;
                const f3 N{nh.x * m, nh.y * m, nh.z * m};
                rbar[0] = t.x * N.x; rbar[1] = t.x * N.y; rbar[2] = t.x * N.z;
                rbar[3] = t.y * N.x; rbar[4] = t.y * N.y; rbar[5] = t.y * N.z;
                rbar[6] = t.z * N.x; rbar[7] = t.z * N.y; rbar[8] = t.z * N.z;
            } else {
                nb.x += t.x; nb.y += t.y; nb.z += t.z;
            }
        }
        nb = f3{nb.x * m, nb.y * m, nb.z * m};
        f3 cb;
        if (s == 0.0f) {
            cb = nb;
        } else {
            const float dt = nh.x * nb.x + nh.y * nb.y + nh.z * nb.z;
            cb = f3{(nb.x - nh.x * dt) / s, (nb.y - nh.y * dt) / s, (nb.z - nh.z * dt) / s};
        }
        // c = dh x dv  ->  dh_bar = dv x c_bar ; dv_bar = c_bar x dh
        st3(dhb, pix, f3{dv.y * cb.z - dv.z * cb.y, dv.z * cb.x - dv.x * cb.z, dv.x * cb.y - dv.y * cb.x});
        st3(dvb, pix, f3{cb.y * dh.z - cb.z * dh.y, cb.z * dh.x - cb.x * dh.z, cb.x * dh.y - cb.y * dh.x});
    }
    {   // per-block partial sums; vn_bwd_final_k adds them up (no same-address atomics, fixed order)
        __shared__ float sm[4][9];
        const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            const float v = wave_sum(rbar[i]);
            if (lane == 0) sm[wid][i] = v;
        }
        __syncthreads();
        if (threadIdx.x < 9)
            part[((int64_t)bl * gridDim.x + blockIdx.x) * VN_PART + 16 + threadIdx.x] =
                sm[0][threadIdx.x] + sm[1][threadIdx.x] + sm[2][threadIdx.x] + sm[3][threadIdx.x];
    }
}

__global__ __launch_bounds__(256) void vn_bwd_pass2_k(const float *__restrict__ depth, const float *__restrict__ Ks,
                                                      const float *__restrict__ poses, int L, int H, int W,
                                                      const float *__restrict__ g_vertex,
                                                      const float *__restrict__ g_gvertex,
                                                      const float *__restrict__ dhb, const float *__restrict__ dvb,
                                                      float *__restrict__ g_depth, float *__restrict__ part) {
    const int bl = blockIdx.y;
    const int b = bl / L;
    const int64_t HW = (int64_t)H * W;
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    // accumulators: 0..8 R_bar, 9..11 t_bar, 12 a_bar, 13 c_bar, 14 e_bar, 15 f_bar
    float acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
    if (p < HW) {
        const int h = (int)(p / W), w = (int)(p - (int64_t)h * W);
        const int64_t base = (int64_t)bl * HW;
        const int64_t pix = base + p;
        const Kinv4 k = load_kinv(Ks + 16 * b);
        const float d = depth[pix];
        const float m = d > 0.0f ? 1.0f : 0.0f;
        f3 vb{0, 0, 0};
        if (g_vertex) vb = ld3(g_vertex, pix);
        if (g_gvertex) {
            f3 t = ld3(g_gvertex, pix);
            t = f3{t.x * m, t.y * m, t.z * m};
            if (poses) {
                const float *T = poses + 16 * (int64_t)bl;
                vb.x += T[0] * t.x + T[4] * t.y + T[8] * t.z;
                vb.y += T[1] * t.x + T[5] * t.y + T[9] * t.z;
                vb.z += T[2] * t.x + T[6] * t.y + T[10] * t.z;
                const f3 v = vertex_of(k, h, w, d);
                acc[0] = t.x * v.x; acc[1] = t.x * v.y; acc[2] = t.x * v.z;
                acc[3] = t.y * v.x; acc[4] = t.y * v.y; acc[5] = t.y * v.z;
                acc[6] = t.z * v.x; acc[7] = t.z * v.y; acc[8] = t.z * v.z;
                acc[9] = t.x; acc[10] = t.y; acc[11] = t.z;
            } else {
                vb.x += t.x; vb.y += t.y; vb.z += t.z;
            }
        }
        if (dhb) {
            // horizontal: pixel q=(h,w) is the right end of dh(h,w-1) [w>=1], the left end of dh(h,w)
            // [w<W-1]; the replicated last column adds +dh(h,W-1) at w==W-1 and -dh(h,W-1) at w==W-2.
            auto add = [&](int64_t q, float sgn) {
                const f3 t = ld3(dhb, base + q);
                vb.x += sgn * t.x; vb.y += sgn * t.y; vb.z += sgn * t.z;
            };
            auto addv = [&](int64_t q, float sgn) {
                const f3 t = ld3(dvb, base + q);
                vb.x += sgn * t.x; vb.y += sgn * t.y; vb.z += sgn * t.z;
            };
            if (w >= 1) add(p - 1, 1.0f);
            if (w < W - 1) add(p, -1.0f);
            if (w == W - 1) add(p, 1.0f);
            if (w == W - 2) add(p + 1, -1.0f);
            if (h >= 1) addv(p - W, 1.0f);
            if (h < H - 1) addv(p, -1.0f);
            if (h == H - 1) addv(p, 1.0f);
            if (h == H - 2) addv(p + W, -1.0f);
        }
        // V = m d r(p),  r = (a w + c, e h + f, 1)
        const float rx = k.a * (float)w + k.c, ry = k.e * (float)h + k.f;
        if (g_depth) g_depth[pix] += m * (vb.x * rx + vb.y * ry + vb.z);
        const float rbx = m * d * vb.x, rby = m * d * vb.y;
        acc[12] = rbx * (float)w; acc[13] = rbx; acc[14] = rby * (float)h; acc[15] = rby;
    }
    __shared__ float sm[4][16];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const float v = wave_sum(acc[i]);
        if (lane == 0) sm[wid][i] = v;
    }
    __syncthreads();
    if (threadIdx.x < 16)
        part[((int64_t)bl * gridDim.x + blockIdx.x) * VN_PART + threadIdx.x] =
            sm[0][threadIdx.x] + sm[1][threadIdx.x] + sm[2][threadIdx.x] + sm[3][threadIdx.x];
}

// one block per (b,l): adds the per-block partials up and applies them to the pose / intrinsics adjoints
__global__ __launch_bounds__(1024) void vn_bwd_final_k(const float *__restrict__ part, int nblocks, int have_pass1,
                                                      const float *__restrict__ Ks, const float *__restrict__ poses, int L,
                                                      float *__restrict__ g_K, float *__restrict__ g_poses) {
    __shared__ float stage[VN_PART][33];
    __shared__ float tot[VN_PART];
    const int bl = blockIdx.x, b = bl / L;
    const int k = threadIdx.x & 31, g = threadIdx.x >> 5;  // 32 groups
    float v = 0.0f;
    if (k < VN_PART && (k < 16 || have_pass1)) {
#pragma unroll 4
        for (int i = g; i < nblocks; i += 32) v += part[((int64_t)bl * nblocks + i) * VN_PART + k];
    }
    if (k < VN_PART) stage[k][g] = v;
    __syncthreads();
    if (threadIdx.x < VN_PART) {
        float t = 0.0f;
#pragma unroll
        for (int q = 0; q < 32; ++q) t += stage[threadIdx.x][q];
        tot[threadIdx.x] = t;
    }
    __syncthreads();
    if (threadIdx.x < 12 && g_poses && poses) {
        const int i = threadIdx.x;
        const int slot = (i < 9) ? 4 * (i / 3) + (i % 3) : 4 * (i - 9) + 3;
        g_poses[16 * (int64_t)bl + slot] += tot[i] + (i < 9 ? tot[16 + i] : 0.0f);
    }
    if (threadIdx.x == 32 && g_K) {
        // a = 1/(fx+eps), c = -cx/(fx+eps), e = 1/(fy+eps), f = -cy/(fy+eps)
        const float *K = Ks + 16 * b;
        const float fx = K[0] + 1e-6f, fy = K[5] + 1e-6f;
        float *gk = g_K + 16 * b;  // L frames share one K: atomics (L per address)
        atomicAdd(gk + 0, -tot[12] / (fx * fx) + tot[13] * K[2] / (fx * fx));
        atomicAdd(gk + 2, -tot[13] / fx);
        atomicAdd(gk + 5, -tot[14] / (fy * fy) + tot[15] * K[6] / (fy * fy));
        atomicAdd(gk + 6, -tot[15] / fy);
    }
}

// ------------------------------------------------------------------ get_alpha
__global__ void alpha_k(const float *__restrict__ pts, int64_t n, float inv2s2_den, float eps, float *__restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const f3 p = ld3(pts, i);
        const float ss = (p.x * p.x + p.y * p.y) + p.z * p.z;  // torch.sum(points**2): no fusion
        float a = expf((-ss) / inv2s2_den);
        a = fminf(fmaxf(a, eps), 1.01f);
        out[i] = a;
    }
}
__global__ void alpha_bwd_k(const float *__restrict__ pts, int64_t n, float den, float eps,
                            const float *__restrict__ g_alpha, float *__restrict__ g_pts) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const f3 p = ld3(pts, i);
        const float ss = (p.x * p.x + p.y * p.y) + p.z * p.z;
        const float a = expf((-ss) / den);
        // clamp passes gradient only strictly inside [eps, 1.01] (torch.clamp: inclusive bounds pass)
        const float g = (a >= eps && a <= 1.01f) ? g_alpha[i] * a * (-2.0f / den) : 0.0f;
        float *q = g_pts + 3 * i;
        q[0] += g * p.x; q[1] += g * p.y; q[2] += g * p.z;
    }
}

// ------------------------------------------------------------------ frame downsample (D)
// (DsPred / DsWriter: gs_project.hpp -- project.hip runs them next to the map's projection in gs_slam_localize)

// ------------------------------------------------------------------ generic row compaction
struct MaskPred {
    const uint8_t *mask;
    __device__ bool operator()(int64_t i) const { return mask[i] != 0; }
};
struct RowWriter {
    const uint32_t *src;
    uint32_t *out;
    int words;
    __device__ void operator()(int64_t i, int64_t pos) const {
        for (int k = 0; k < words; ++k) out[pos * words + k] = src[i * words + k];
    }
};

struct MultiWriter {
    const uint32_t *src[4];
    uint32_t *out[4];
    int words[4];
    int n_arrays;
    __device__ void operator()(int64_t i, int64_t pos) const {
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            if (a >= n_arrays) break;
            copy_row(src[a], out[a], i, pos, words[a]);
        }
    }
};

// Raw sensor frames -> the float tensors of RGBDImages, on the device (reference datasets/tum.py:346,
// :455-499): depth = uint16 / scaling_factor with nearest-neighbour resize, colour = uint8 (optionally / 255)
// with bilinear resize (pixel centres at +0.5, edges clamped).  fp64 inside, rounded once, like the
// reference's float64 numpy arithmetic followed by .float().  Same-size frames are exact copies.
__global__ void frames_from_raw_k(const uint16_t *__restrict__ depth_raw, const uint8_t *__restrict__ rgb_raw, int Hs, int Ws,
                                  int Hd, int Wd, double depth_scale, int normalise, float *__restrict__ depth,
                                  float *__restrict__ rgb) {
    const int b = blockIdx.y;
    const int64_t npix = (int64_t)Hd * Wd;
    const double ry = (double)Hs / Hd, rx = (double)Ws / Wd;
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < npix; p += (int64_t)gridDim.x * blockDim.x) {
        const int y = (int)(p / Wd), x = (int)(p - (int64_t)y * Wd);
        if (depth_raw) {
            const int sy = min((int)floor(y * ry), Hs - 1), sx = min((int)floor(x * rx), Ws - 1);
            depth[(int64_t)b * npix + p] = (float)((double)depth_raw[((int64_t)b * Hs + sy) * Ws + sx] / depth_scale);
        }
        if (rgb_raw) {
            double fy = (y + 0.5) * ry - 0.5, fx = (x + 0.5) * rx - 0.5;
            int y0 = (int)floor(fy), x0 = (int)floor(fx);
            fy -= y0; fx -= x0;
            if (y0 < 0) { y0 = 0; fy = 0.0; }
            if (x0 < 0) { x0 = 0; fx = 0.0; }
            if (y0 >= Hs - 1) { y0 = Hs - 1; fy = 0.0; }
            if (x0 >= Ws - 1) { x0 = Ws - 1; fx = 0.0; }
            const int y1 = min(y0 + 1, Hs - 1), x1 = min(x0 + 1, Ws - 1);
            const uint8_t *img = rgb_raw + (int64_t)b * Hs * Ws * 3;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const double v00 = img[((int64_t)y0 * Ws + x0) * 3 + c], v01 = img[((int64_t)y0 * Ws + x1) * 3 + c];
                const double v10 = img[((int64_t)y1 * Ws + x0) * 3 + c], v11 = img[((int64_t)y1 * Ws + x1) * 3 + c];
                double v = (v00 * (1.0 - fx) + v01 * fx) * (1.0 - fy) + (v10 * (1.0 - fx) + v11 * fx) * fy;
                if (normalise) v /= 255.0;
                rgb[((int64_t)b * npix + p) * 3 + c] = (float)v;
            }
        }
    }
}

// adjoint of MultiWriter: row i of every output = the compacted row it went to, or zero
struct ExpandWriter {
    const uint32_t *g[4];
    uint32_t *out[4];
    int words[4];
    int n_arrays;
    __device__ void operator()(int64_t i, int64_t pos) const {
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            if (a >= n_arrays) break;
            const int w = words[a];
            for (int k = 0; k < w; ++k) out[a][i * w + k] = g[a][pos * w + k];
        }
    }
    __device__ void skip(int64_t i) const {
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            if (a >= n_arrays) break;
            const int w = words[a];
            for (int k = 0; k < w; ++k) out[a][i * w + k] = 0u;
        }
    }
};

}  // namespace gs

using namespace gs;

extern "C" {

int gs_abi_version(void) { return GS_ABI_VERSION; }
const char *gs_last_error(void) { return gs::g_err; }

int gs_vertex_normal_maps(const float *depth, const float *intrinsics, const float *poses, int B, int L, int H, int W,
                          float *vertex, float *normal, float *gvertex, float *gnormal, gs_stream_t stream) {
    GS_REQUIRE(depth && intrinsics, "gs_vertex_normal_maps: depth/intrinsics must not be NULL");
    GS_REQUIRE(B > 0 && L > 0 && H >= 2 && W >= 2, "gs_vertex_normal_maps: need B,L>0 and H,W>=2 (got %d,%d,%d,%d)", B, L, H, W);
    GS_REQUIRE((int64_t)B * L <= 65535, "gs_vertex_normal_maps: B*L must be <= 65535");
    dim3 grid(cdiv(W, TW), cdiv(H, TH), B * L);
    hipLaunchKernelGGL(vertex_normal_k, grid, dim3(TW * TH), 0, (hipStream_t)stream, depth, intrinsics, poses, L, H, W,
                       vertex, normal, gvertex, gnormal, VnExtra{nullptr, 1.0f, 0.0f, nullptr, nullptr, nullptr, 0, nullptr});
    GS_LAUNCH_CHECK("gs_vertex_normal_maps");
    return GS_OK;
}

}  // extern "C"

namespace gs {
// the maps of ONE frame per batch element for the PointFusion update, with its riders (VnExtra): global maps and alpha
// out, per-pixel correspondence state initialised, counter block zeroed
int vertex_normal_maps_fusion(const float *depth, const float *intrinsics, const float *poses, int B, int H, int W, float *gvertex,
                              float *gnormal, float *alpha, float sigma, float eps, unsigned long long *pix_key, unsigned int *pix_n,
                              int32_t *zero, int n_zero, hipStream_t st) {
    dim3 grid(cdiv(W, TW), cdiv(H, TH), B);
    hipLaunchKernelGGL(vertex_normal_k, grid, dim3(TW * TH), 0, st, depth, intrinsics, poses, 1, H, W, (float *)nullptr,
                       (float *)nullptr, gvertex, gnormal, VnExtra{alpha, 2.0f * (sigma * sigma), eps, pix_key, pix_n, zero, n_zero, nullptr});
    GS_LAUNCH_CHECK("gs_pointfusion_update/maps");
    return GS_OK;
}
// the maps of gs_slam_localize (one frame per batch element) + a copy of (pose | intrinsics) per batch element at cam_out
int vertex_normal_maps_cam(const float *depth, const float *intrinsics, const float *poses, int B, int H, int W, float *vertex,
                           float *normal, float *gvertex, float *gnormal, float *cam_out, hipStream_t st) {
    dim3 grid(cdiv(W, TW), cdiv(H, TH), B);
    hipLaunchKernelGGL(vertex_normal_k, grid, dim3(TW * TH), 0, st, depth, intrinsics, poses, 1, H, W, vertex, normal, gvertex, gnormal,
                       VnExtra{nullptr, 1.0f, 0.0f, nullptr, nullptr, nullptr, 0, cam_out});
    GS_LAUNCH_CHECK("gs_slam_localize/maps");
    return GS_OK;
}
}  // namespace gs

extern "C" {

size_t gs_vertex_normal_maps_backward_ws_bytes(int B, int L, int H, int W) {
    return align_up((size_t)B * L * H * W * 3 * sizeof(float), 256) * 2 +
           align_up((size_t)B * L * cdiv((int64_t)H * W, 256) * VN_PART * sizeof(float), 256);
}

int gs_vertex_normal_maps_backward(const float *depth, const float *intrinsics, const float *poses, int B, int L, int H,
                                   int W, const float *g_vertex, const float *g_normal, const float *g_gvertex,
                                   const float *g_gnormal, float *g_depth, float *g_intrinsics, float *g_poses,
                                   void *ws, size_t ws_bytes, gs_stream_t stream) {
    GS_REQUIRE(depth && intrinsics, "gs_vertex_normal_maps_backward: depth/intrinsics must not be NULL");
    GS_REQUIRE(B > 0 && L > 0 && H >= 2 && W >= 2 && (int64_t)B * L <= 65535, "gs_vertex_normal_maps_backward: bad shape");
    const bool need_n = g_normal || g_gnormal;
    float *dhb = nullptr, *dvb = nullptr;
    hipStream_t st = (hipStream_t)stream;
    const int64_t HW = (int64_t)H * W;
    dim3 grid(cdiv(HW, 256), B * L);
    if (ws_bytes < gs_vertex_normal_maps_backward_ws_bytes(B, L, H, W) || !ws) {
        set_error("gs_vertex_normal_maps_backward: workspace too small");
        return GS_ERR_WORKSPACE_TOO_SMALL;
    }
    const size_t map_b = align_up((size_t)B * L * HW * 3 * sizeof(float), 256);
    float *part = (float *)((char *)ws + 2 * map_b);
    if (need_n) {
        dhb = (float *)ws;
        dvb = (float *)((char *)ws + map_b);
        hipLaunchKernelGGL(vn_bwd_pass1_k, grid, dim3(256), 0, st, depth, intrinsics, poses, L, H, W, g_normal,
                           g_gnormal, dhb, dvb, part);
        GS_LAUNCH_CHECK("gs_vertex_normal_maps_backward/1");
    }
    hipLaunchKernelGGL(vn_bwd_pass2_k, grid, dim3(256), 0, st, depth, intrinsics, poses, L, H, W, g_vertex, g_gvertex,
                       dhb, dvb, g_depth, part);
    if (g_intrinsics || (g_poses && poses))
        hipLaunchKernelGGL(vn_bwd_final_k, dim3(B * L), dim3(1024), 0, st, part, (int)grid.x, need_n ? 1 : 0, intrinsics, poses, L,
                           g_intrinsics, g_poses);
    GS_LAUNCH_CHECK("gs_vertex_normal_maps_backward/2");
    return GS_OK;
}

int gs_get_alpha(const float *points, int64_t n, float sigma, float eps, float *alpha, gs_stream_t stream) {
    GS_REQUIRE(points && alpha && n >= 0, "gs_get_alpha: bad arguments");
    if (n == 0) return GS_OK;
    const float den = 2.0f * (sigma * sigma);
    hipLaunchKernelGGL(alpha_k, dim3(min(cdiv(n, 256), 4096)), dim3(256), 0, (hipStream_t)stream, points, n, den, eps, alpha);
    GS_LAUNCH_CHECK("gs_get_alpha");
    return GS_OK;
}

int gs_get_alpha_backward(const float *points, int64_t n, float sigma, float eps, const float *g_alpha, float *g_points,
                          gs_stream_t stream) {
    GS_REQUIRE(points && g_alpha && g_points && n >= 0, "gs_get_alpha_backward: bad arguments");
    if (n == 0) return GS_OK;
    const float den = 2.0f * (sigma * sigma);
    hipLaunchKernelGGL(alpha_bwd_k, dim3(min(cdiv(n, 256), 4096)), dim3(256), 0, (hipStream_t)stream, points, n, den, eps,
                       g_alpha, g_points);
    GS_LAUNCH_CHECK("gs_get_alpha_backward");
    return GS_OK;
}

size_t gs_compact_ws_bytes(int64_t n_rows) { return compact_ws_bytes(n_rows); }

int gs_compact_rows(const float *src, const uint8_t *mask, int64_t n_rows, int row_floats, float *out,
                    int32_t *out_count, void *ws, size_t ws_bytes, gs_stream_t stream) {
    GS_REQUIRE(mask && out_count && n_rows >= 0 && row_floats >= 0, "gs_compact_rows: bad arguments");
    GS_REQUIRE(row_floats == 0 || (src && out), "gs_compact_rows: src/out must not be NULL when row_floats > 0");
    if (ws_bytes < compact_ws_bytes(n_rows) || !ws) {
        set_error("gs_compact_rows: workspace too small (%zu < %zu)", ws_bytes, compact_ws_bytes(n_rows));
        return GS_ERR_WORKSPACE_TOO_SMALL;
    }
    MaskPred pred{mask};
    RowWriter wr{(const uint32_t *)src, (uint32_t *)out, row_floats};
    return compact_launch(n_rows, pred, wr, out_count, ws, (hipStream_t)stream, "gs_compact_rows");
}

int gs_compact_multi(int n_arrays, const float *const *h_src, const int *h_row_floats, float *const *h_out,
                     const uint8_t *mask, int64_t n_rows, int32_t *out_count, void *ws, size_t ws_bytes,
                     gs_stream_t stream) {
    GS_REQUIRE(n_arrays >= 1 && n_arrays <= 4 && h_src && h_row_floats && h_out && mask && out_count && n_rows >= 0,
               "gs_compact_multi: bad arguments (1..4 arrays)");
    if (ws_bytes < compact_ws_bytes(n_rows) || !ws) {
        set_error("gs_compact_multi: workspace too small (%zu < %zu)", ws_bytes, compact_ws_bytes(n_rows));
        return GS_ERR_WORKSPACE_TOO_SMALL;
    }
    MultiWriter wr;
    wr.n_arrays = n_arrays;
    for (int a = 0; a < 4; ++a) {
        wr.src[a] = a < n_arrays ? (const uint32_t *)h_src[a] : nullptr;
        wr.out[a] = a < n_arrays ? (uint32_t *)h_out[a] : nullptr;
        wr.words[a] = a < n_arrays ? h_row_floats[a] : 0;
        GS_REQUIRE(a >= n_arrays || (h_src[a] && h_out[a] && h_row_floats[a] > 0), "gs_compact_multi: NULL array %d", a);
    }
    MaskPred pred{mask};
    return compact_launch(n_rows, pred, wr, out_count, ws, (hipStream_t)stream, "gs_compact_multi");
}

int gs_expand_multi(int n_arrays, const float *const *h_grad, const int *h_row_floats, float *const *h_out,
                    const uint8_t *mask, int64_t n_rows, void *ws, size_t ws_bytes, gs_stream_t stream) {
    GS_REQUIRE(n_arrays >= 1 && n_arrays <= 4 && h_grad && h_row_floats && h_out && mask && n_rows >= 0,
               "gs_expand_multi: bad arguments (1..4 arrays)");
    if (ws_bytes < compact_ws_bytes(n_rows) || !ws) {
        set_error("gs_expand_multi: workspace too small (%zu < %zu)", ws_bytes, compact_ws_bytes(n_rows));
        return GS_ERR_WORKSPACE_TOO_SMALL;
    }
    ExpandWriter wr;
    wr.n_arrays = n_arrays;
    for (int a = 0; a < 4; ++a) {
        wr.g[a] = a < n_arrays ? (const uint32_t *)h_grad[a] : nullptr;
        wr.out[a] = a < n_arrays ? (uint32_t *)h_out[a] : nullptr;
        wr.words[a] = a < n_arrays ? h_row_floats[a] : 0;
        GS_REQUIRE(a >= n_arrays || (h_grad[a] && h_out[a] && h_row_floats[a] > 0), "gs_expand_multi: NULL array %d", a);
    }
    MaskPred pred{mask};
    return compact_launch(n_rows, pred, wr, (int *)nullptr, ws, (hipStream_t)stream, "gs_expand_multi");
}

int gs_frames_from_raw(const uint16_t *depth_raw, const uint8_t *rgb_raw, int B, int Hs, int Ws, int Hd, int Wd,
                       float depth_scale, int normalise_color, float *depth, float *rgb, gs_stream_t stream) {
    GS_REQUIRE(B > 0 && B <= 65535 && Hs > 0 && Ws > 0 && Hd > 0 && Wd > 0, "gs_frames_from_raw: bad shape");
    GS_REQUIRE((!depth_raw || depth) && (!rgb_raw || rgb) && (depth_raw || rgb_raw), "gs_frames_from_raw: NULL output for a given input");
    GS_REQUIRE(!depth_raw || depth_scale > 0.0f, "gs_frames_from_raw: depth_scale must be positive");
    hipLaunchKernelGGL(frames_from_raw_k, dim3(min(cdiv((int64_t)Hd * Wd, 256), 2048), B), dim3(256), 0, (hipStream_t)stream,
                       depth_raw, rgb_raw, Hs, Ws, Hd, Wd, (double)depth_scale, normalise_color, depth, rgb);
    GS_LAUNCH_CHECK("gs_frames_from_raw");
    return GS_OK;
}

size_t gs_append_rows_ws_bytes(int64_t n_rows) { return compact_ws_bytes(n_rows) + 256; }

int gs_append_rows(int n_arrays, const float *const *h_src, const int *h_row_floats, float *const *h_dst, const uint8_t *mask,
                   int64_t n_rows, int32_t *d_count, int cap, int32_t *d_appended, int32_t *d_overflow, void *ws,
                   size_t ws_bytes, gs_stream_t stream) {
    GS_REQUIRE(n_arrays >= 1 && n_arrays <= 4 && h_src && h_row_floats && h_dst && mask && d_count && n_rows >= 0 && cap >= 0,
               "gs_append_rows: bad arguments (1..4 arrays)");
    if (ws_bytes < gs_append_rows_ws_bytes(n_rows) || !ws) {
        set_error("gs_append_rows: workspace too small (%zu < %zu)", ws_bytes, gs_append_rows_ws_bytes(n_rows));
        return GS_ERR_WORKSPACE_TOO_SMALL;
    }
    AppendWriter wr;
    wr.n_arrays = n_arrays;
    wr.base = d_count;
    wr.cap = cap;
    for (int a = 0; a < 4; ++a) {
        wr.src[a] = a < n_arrays ? (const uint32_t *)h_src[a] : nullptr;
        wr.out[a] = a < n_arrays ? (uint32_t *)h_dst[a] : nullptr;
        wr.words[a] = a < n_arrays ? h_row_floats[a] : 0;
        GS_REQUIRE(a >= n_arrays || (h_src[a] && h_dst[a] && h_row_floats[a] > 0), "gs_append_rows: NULL array %d", a);
    }
    int *total = (int *)((char *)ws + compact_ws_bytes(n_rows));
    MaskPred pred{mask};
    const int rc = compact_launch(n_rows, pred, wr, total, ws, (hipStream_t)stream, "gs_append_rows");
    if (rc) return rc;
    hipLaunchKernelGGL(append_count_k, dim3(1), dim3(64), 0, (hipStream_t)stream, d_count, total, cap, d_appended, d_overflow);
    GS_LAUNCH_CHECK("gs_append_rows/count");
    return GS_OK;
}

size_t gs_downsample_frame_ws_bytes(int H, int W, int ds) {
    return compact_ws_bytes((int64_t)cdiv(H, ds) * cdiv(W, ds));
}

int gs_downsample_frame(const float *depth, const float *gvertex, const float *gnormal, const float *rgb, int B, int H,
                        int W, int ds, int cap, float *out_points, float *out_normals, float *out_colors,
                        int32_t *out_pix, int32_t *counts, void *ws, size_t ws_bytes, gs_stream_t stream) {
    GS_REQUIRE(depth && counts && B > 0 && H > 0 && W > 0 && ds > 0, "gs_downsample_frame: bad arguments");
    const int Hd = cdiv(H, ds), Wd = cdiv(W, ds);
    GS_REQUIRE(cap >= Hd * Wd, "gs_downsample_frame: cap (%d) < ceil(H/ds)*ceil(W/ds) (%d)", cap, Hd * Wd);
    GS_REQUIRE((!out_points || gvertex) && (!out_normals || gnormal) && (!out_colors || rgb),
               "gs_downsample_frame: an output was requested without its source map");
    if (ws_bytes < gs_downsample_frame_ws_bytes(H, W, ds) || !ws) {
        set_error("gs_downsample_frame: workspace too small");
        return GS_ERR_WORKSPACE_TOO_SMALL;
    }
    const int64_t HW = (int64_t)H * W;
    for (int b = 0; b < B; ++b) {
        DsPred pred{depth + b * HW, W, Wd, ds};
        DsWriter wr{gvertex ? gvertex + 3 * b * HW : nullptr, gnormal ? gnormal + 3 * b * HW : nullptr,
                    rgb ? rgb + 3 * b * HW : nullptr,
                    out_points ? out_points + 3 * (int64_t)b * cap : nullptr,
                    out_normals ? out_normals + 3 * (int64_t)b * cap : nullptr,
                    out_colors ? out_colors + 3 * (int64_t)b * cap : nullptr,
                    out_pix ? out_pix + (int64_t)b * cap : nullptr, W, Wd, ds};
        int rc = compact_launch((int64_t)Hd * Wd, pred, wr, counts + b, ws, (hipStream_t)stream, "gs_downsample_frame");
        if (rc != GS_OK) return rc;
    }
    return GS_OK;
}

}  // extern "C"
