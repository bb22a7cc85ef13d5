// gs_project.hpp -- the active-map-point projection shared by project.hip (the table-building entry points) and
// fusion.hip (the fused correspondence pass of gs_pointfusion_update): ONE definition, so that both take exactly the
// same fp32 decisions.
#pragma once

#include "gs_common.hpp"

namespace gs {

struct Cam {
    float R[9];     // camera pose rotation (world <- cam), row-major
    float tinv[3];  // -R^T t
    float K[16];
};

// One Cam per batch element, computed on device (no host math, no sync).
// inverse_transformation semantics (R^T, -R^T t); the small batched matmul(-R^T, t) does not fuse.
__device__ __forceinline__ Cam make_cam(const float *__restrict__ T, const float *__restrict__ K) {
    Cam c;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) c.R[3 * i + j] = T[4 * i + j];
    const float t0 = T[3], t1 = T[7], t2 = T[11];
#pragma unroll
    for (int j = 0; j < 3; ++j) c.tinv[j] = ((-T[j]) * t0 + (-T[4 + j]) * t1) + (-T[8 + j]) * t2;
#pragma unroll
    for (int i = 0; i < 16; ++i) c.K[i] = K[i];
    return c;
}
// Projects map point p of batch element with camera c.  Returns true iff the point is active and
// fills (h, w).  reference slam/fusionutils.py:250-274, structures/pointclouds.py:501-517,423-425,
// geometry/projutils.py:221-236.
__device__ __forceinline__ bool project_point(const Cam &c, f3 p, int H, int W, float umax, float vmax, int &h, int &w) {
    // p' = p . Rinv^T + tinv, Rinv^T == R: p'_k = sum_j p_j R[j][k]   (GEMM contraction)
    const float x = dot3_fma(p.x, p.y, p.z, c.R[0], c.R[3], c.R[6]) + c.tinv[0];
    const float y = dot3_fma(p.x, p.y, p.z, c.R[1], c.R[4], c.R[7]) + c.tinv[1];
    const float z = dot3_fma(p.x, p.y, p.z, c.R[2], c.R[5], c.R[8]) + c.tinv[2];
    const bool front = z > 0.0f;
    // K4x4 . [p',1]: broadcast batched 4x4 @ 4x1 -> plain (unfused) accumulation
    const float *K = c.K;
    const float qx = ((K[0] * x + K[1] * y) + K[2] * z) + K[3] * 1.0f;
    const float qy = ((K[4] * x + K[5] * y) + K[6] * z) + K[7] * 1.0f;
    const float qz = ((K[8] * x + K[9] * y) + K[10] * z) + K[11] * 1.0f;
    const float zs = (qz != 0.0f) ? qz : 1.0f;
    const float u = qx / zs, v = qy / zs;
    // umax = fp32(W - 0.999), vmax = fp32(H - 0.999): formed in double on the host like the reference's
    // Python scalars, then rounded once
    const bool in = (u > -1e-3f) && (u < umax) && (v > -1e-3f) && (v < vmax) && front;
    // round-half-to-even like torch.round; clamp like .clamp(0, H-1)
    const float ru = rintf(u), rv = rintf(v);
    w = (int)fminf(fmaxf(ru, 0.0f), (float)(W - 1));
    h = (int)fminf(fmaxf(rv, 0.0f), (float)(H - 1));
    return in;
}


// ------------------------------------------------------------------ the ds-grid source cloud of a frame (gs_downsample_frame)
// reference structures/structutils.py downsample_rgbdimages: every ds-th pixel of every ds-th row with valid depth
struct DsPred {
    const float *depth;  // (H,W) of one batch element
    int W, Wd, ds;
    __device__ bool operator()(int64_t i) const {
        const int r = (int)(i / Wd), c = (int)(i - (int64_t)r * Wd);
        return depth[(int64_t)(r * ds) * W + c * ds] > 0.0f;
    }
};
struct DsWriter {
    const float *gv, *gn, *rgb;
    float *op, *on, *oc;
    int32_t *opix;
    int W, Wd, ds;
    __device__ void operator()(int64_t i, int64_t pos) const {
        const int r = (int)(i / Wd), c = (int)(i - (int64_t)r * Wd);
        const int64_t pix = (int64_t)(r * ds) * W + c * ds;
        if (opix) opix[pos] = (int32_t)i;  // ds-grid pixel id r * Wd + c
        if (op) st3(op, pos, ld3(gv, pix));
        if (on) st3(on, pos, ld3(gn, pix));
        if (oc) st3(oc, pos, ld3(rgb, pix));
    }
};

// one sequence's job for project_target1 (project.hip), which runs it on the launches of the map's projection
struct DsJob {
    const float *depth, *gvertex;  // (H, W), (H, W, 3)
    float *out_points;             // (cap, 3)
    int32_t *out_pix;              // (cap)
    int32_t *count;                // device count of selected pixels
};

}  // namespace gs
