// gs_common.hpp -- shared host/device helpers for libgradslam_hip (gfx950 / CDNA4 only).
//
// Build flags matter for parity: the library is compiled with -ffp-contract=off so that every
// fused multiply-add in the kernels is an explicit __fmaf_rn placed where the reference's CPU
// kernels were measured to fuse (DESIGN.md "rounding conventions"), and nowhere else.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/gradslam_hip.h"

namespace gs {

constexpr int kWave = 64;  // CDNA wavefront

// ------------------------------------------------------------------ host-side error plumbing
void set_error(const char *fmt, ...);

#define GS_REQUIRE(cond, ...)              \
    do {                                   \
        if (!(cond)) {                     \
            gs::set_error(__VA_ARGS__);    \
            return GS_ERR_INVALID_ARG;     \
        }                                  \
    } while (0)

#define GS_LAUNCH_CHECK(name)                                                    \
    do {                                                                         \
        hipError_t e__ = hipGetLastError();                                      \
        if (e__ != hipSuccess) {                                                 \
            gs::set_error("%s: launch failed: %s", name, hipGetErrorString(e__)); \
            return (int)e__;                                                     \
        }                                                                        \
    } while (0)

#define GS_HIP(call, name)                                                \
    do {                                                                  \
        hipError_t e__ = (call);                                          \
        if (e__ != hipSuccess) {                                          \
            gs::set_error("%s: %s", name, hipGetErrorString(e__));        \
            return (int)e__;                                              \
        }                                                                 \
    } while (0)

static inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }
static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// ------------------------------------------------------------------ device helpers
struct f3 {
    float x, y, z;
};

__device__ __forceinline__ f3 ld3(const float *__restrict__ p, int64_t i) {
    const float *q = p + 3 * i;
    return f3{q[0], q[1], q[2]};
}
__device__ __forceinline__ void st3(float *__restrict__ p, int64_t i, f3 v) {
    float *q = p + 3 * i;
    q[0] = v.x;
    q[1] = v.y;
    q[2] = v.z;
}

// K=3 contraction in the order the reference's CPU GEMM was measured to use:
//   fma(a2, b2, fma(a1, b1, a0 * b0))
__device__ __forceinline__ float dot3_fma(float a0, float a1, float a2, float b0, float b1, float b2);
// kornia.geometry.linalg.compose_transformations semantics (call site reference slam/icpslam.py:245-247):
// R = R01 R12 ; t = R01 t12 + t01 ; bottom row [0,0,0,1].   o = a . p   (4x4 row-major each)
__device__ __forceinline__ void compose44(const float *a, const float *p, float *o) {
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) o[4 * i + j] = dot3_fma(a[4 * i], a[4 * i + 1], a[4 * i + 2], p[j], p[4 + j], p[8 + j]);
        o[4 * i + 3] = dot3_fma(a[4 * i], a[4 * i + 1], a[4 * i + 2], p[3], p[7], p[11]) + a[4 * i + 3];
    }
    o[12] = 0.0f; o[13] = 0.0f; o[14] = 0.0f; o[15] = 1.0f;
}
__device__ __forceinline__ float dot3_fma(float a0, float a1, float a2, float b0, float b1, float b2) {
    return __fmaf_rn(a2, b2, __fmaf_rn(a1, b1, a0 * b0));
}
// elementwise-then-sum contraction (no fusion): (a0 b0 + a1 b1) + a2 b2
__device__ __forceinline__ float dot3_plain(float a0, float a1, float a2, float b0, float b1, float b2) {
    return (a0 * b0 + a1 * b1) + a2 * b2;
}

// rigid transform p' = R p + t with the GEMM contraction, T row-major 4x4
__device__ __forceinline__ f3 xform(const float *__restrict__ T, f3 p) {
    return f3{dot3_fma(T[0], T[1], T[2], p.x, p.y, p.z) + T[3],
              dot3_fma(T[4], T[5], T[6], p.x, p.y, p.z) + T[7],
              dot3_fma(T[8], T[9], T[10], p.x, p.y, p.z) + T[11]};
}

// 64-lane butterfly sum; every lane ends with the total (deterministic order)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, kWave);
    return v;
}
__device__ __forceinline__ int wave_sum_i(int v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, kWave);
    return v;
}

// exclusive prefix sum of one int per thread over a block of NT threads (NT multiple of 64,
// NT <= 1024).  Returns the exclusive prefix; *total = block sum.  `sm` needs NT/64 + 1 ints.
template <int NT>
__device__ __forceinline__ int block_excl_scan(int v, int *sm, int *total) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    int inc = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        int t = __shfl_up(inc, off, kWave);
        if (lane >= off) inc += t;
    }
    if (lane == 63) sm[wid] = inc;
    __syncthreads();
    if (threadIdx.x == 0) {
        int run = 0;
        for (int w = 0; w < NT / 64; ++w) {
            int t = sm[w];
            sm[w] = run;
            run += t;
        }
        sm[NT / 64] = run;
    }
    __syncthreads();
    const int base = sm[wid];
    *total = sm[NT / 64];
    __syncthreads();  // sm may be reused by the caller
    return base + inc - v;
}

// order-preserving mapping float >= 0  ->  uint32
__device__ __forceinline__ uint32_t fbits(float f) { return __float_as_uint(f); }
__device__ __forceinline__ float bitsf(uint32_t u) { return __uint_as_float(u); }

}  // namespace gs
