// icp.hip -- exact 1-NN association (K), linearise + 6x6 reduce (J), the O(1) solve / SE(3)
// exponential / LM control (X), and whole ICP / gradICP loops that never leave the device.
//
// K  The reference's association is an exact K=1 nearest-neighbour search (squared L2 accumulated
//    x->y->z in fp32 without FMA, strict '<' so the lowest index wins ties).  Three kernels compute it:
//    * knn1_brute_k (the verifier): every (source, target) pair.  FP32-VALU bound (8 flop/pair, no dense contraction
//      -> no MFMA; a |p|^2+|q|^2-2p.q matrix form would change rounding and tie-breaks).  Target points are read
//      with wave-uniform addresses, so they stream through the scalar cache into SGPRs; the launch is split over
//      (source tiles) x (target ranges) and merged with one 64-bit atomic min on the packed key dist_bits<<32 | index.
//    * knn1_box_k / knn1_loop_k<false> (chunk-box search): the same pairs, minus those that provably cannot win.
//      Target points are grouped in chunks of CHUNK = 16 consecutive points with an AABB each.  One 1024-thread
//      block serves one tile of up to 64 source points (in the loops 64, or fewer on a dense target: loop_tile_points):
//      every wave holds the same points (lane = point) and the 16 waves
//      share the target chunks -- a coarse pass (lanes = chunk boxes, against the tile's box and loosest bound), then
//      per-lane bounds ((ex^2+ey^2)+ez^2, e = per-axis gap to the box) and scans with candidates broadcast by
//      v_readlane.  Rounding is monotone and the bound uses the distance's own operation order, so bound <= distance
//      holds exactly in fp32: no epsilon, a chunk is skipped only on a STRICT '>', and the result is bit-identical to
//      the brute-force scan (lexicographic (distance, index) minimum).
//    * knn1_loop_k<true> (grid search with a geometric proof, dense targets with search hints): every point
//      examines the targets of the 3x3 ds-grid pixels around the pixel it projects to (staged in LDS); every other
//      target lies outside a pyramid through the camera centre, and the point's distance to the pyramid's faces proves
//      that the window's best is the nearest neighbour; points whose proof fails take the chunk-box search (see
//      cam_bound2 and the comment above knn1_loop_k).
// J  gather + 29-term reduction, HBM/L2-bound at 40 algorithmic bytes per source point; wave
//    butterflies + a fixed-order two-level tree (deterministic, no float atomics).  Fused into the association
//    kernel's epilogue inside the loops; linearize_k / finalize44_k serve the stand-alone entry points.
// X  the O(1) step of an iteration (reduce the partials, LM / gradLM decision, fp64 6x6 solve, SE(3)
//    exponential) runs in the prologue of the NEXT association launch: every block recomputes it on its wave 0
//    (no block barriers inside; the other waves stage the grid search meanwhile), block 0 publishes it; only a loop's
//    last step is a launch of its own.  Buffers are addressed through device-side role indices, so accept/reject
//    needs no host round trip and no copies.  What is constant over a loop (pointers, hints, parameters) is read from
//    a LoopConst in the workspace, not from kernel arguments: the SGPR count decides whether two blocks share a CU.
// Gradients (X-bar): the taped loops + a device-side reverse pass.  The reverse pass scatters the adjoints of the
//    associated target points / normals with float atomics (-munsafe-fp-atomics): forward results are bit-stable run
//    to run, those two gradient arrays are not (sums of a few terms per target in arrival order, ~1e-7 relative).
#include <stddef.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

#include "gs_common.hpp"
#include <type_traits>
#include "gs_project.hpp"

namespace gs {

constexpr int KNN_T = 256;      // brute force: threads per block
constexpr int KNN_NW = 16;      // pruned search: waves per block, ALL serving the same 64 source points
constexpr int KNN_BT = KNN_NW * 64;
constexpr int KNN_COARSE = 512; // target points sampled by the seed pass when no seed is given
constexpr int CHUNK = 16;       // target points per AABB chunk
constexpr int WROWS = 3;        // grid search: rows of the window (radius 1; radius 2 = WROWS 5 with its six bands costs ~1 us per launch
                                // on a dense target and gains nothing on a sparse one: 22.9 against 21.1 us per launch at c2, r03f)
constexpr int WBANDS = WROWS + 1;  // row bands a tile stages at most (its lanes sit in two adjacent rows)
constexpr int SUPER = 64;       // chunks per super-box (= 1024 target points = one block of icp_prepare_k)
constexpr int KNN_LIST = 4096;  // chunk boxes handled per round (capacity of the LDS survivor list)
constexpr int NACC = 29;        // 21 (upper H) + 6 (g) + e + count
constexpr int LIN_T = 256;
constexpr int LIN_MAXB = 1024;  // max partial blocks of the stand-alone J kernel (best of 512/1024/2048 measured at 2^24 points)
constexpr unsigned long long KEY_NONE = ~0ull;

__device__ __forceinline__ unsigned long long pack_key(float d, int j) {
    return ((unsigned long long)fbits(d) << 32) | (unsigned int)j;
}
__device__ __forceinline__ float dist2(f3 s, float tx, float ty, float tz) {
    const float dx = s.x - tx, dy = s.y - ty, dz = s.z - tz;
    return (dx * dx + dy * dy) + dz * dz;  // contraction off: x->y->z, no fma
}

// ------------------------------------------------------------------ J: row algebra (shared by K's epilogue)
struct Row {
    float a[6], b;
    bool valid;
};

// reference odometry/icputils.py:203-230; every product / difference is rounded on its own
// (elementwise torch ops), so nothing here may fuse.
__device__ __forceinline__ Row make_row(const float *__restrict__ src, const float *__restrict__ tgt,
                                        const float *__restrict__ nrm, const unsigned long long *__restrict__ best,
                                        int i, int ns, float thresh) {
    Row r;
    r.valid = false;
    if (i >= ns) return r;
    const unsigned long long key = best[i];
    if (key == KEY_NONE) return r;  // no target at all
    const uint32_t j = (uint32_t)(key & 0xffffffffu);
    const float d2 = bitsf((uint32_t)(key >> 32));
    if (thresh >= 0.0f && !(d2 < thresh)) return r;  // NB squared distance vs threshold
    const f3 s = ld3(src, i), d = ld3(tgt, j), n = ld3(nrm, j);
    r.a[0] = n.x; r.a[1] = n.y; r.a[2] = n.z;
    r.a[3] = n.z * s.y - n.y * s.z;
    r.a[4] = n.x * s.z - n.z * s.x;
    r.a[5] = n.y * s.x - n.x * s.y;
    r.b = (n.x * (d.x - s.x) + n.y * (d.y - s.y)) + n.z * (d.z - s.z);
    r.valid = true;
    return r;
}

__device__ __forceinline__ void accumulate_row(const Row &r, float *acc) {
    int q = 0;
#pragma unroll
    for (int u = 0; u < 6; ++u)
#pragma unroll
        for (int v = u; v < 6; ++v) { acc[q] = __fmaf_rn(r.a[u], r.a[v], acc[q]); ++q; }
#pragma unroll
    for (int u = 0; u < 6; ++u) acc[21 + u] = __fmaf_rn(r.a[u], r.b, acc[21 + u]);
    acc[27] = __fmaf_rn(r.b, r.b, acc[27]);
    acc[28] += 1.0f;
}

// same row from values already in registers (the association kernel's epilogue)
__device__ __forceinline__ Row make_row_from(const f3 s, const bool ok, const unsigned long long key,
                                             const float *__restrict__ tgt, const float *__restrict__ nrm, float thresh) {
    Row r;
    r.valid = false;
    if (!ok || key == KEY_NONE) return r;
    const uint32_t j = (uint32_t)(key & 0xffffffffu);
    const float d2 = bitsf((uint32_t)(key >> 32));
    if (thresh >= 0.0f && !(d2 < thresh)) return r;
    const f3 d = ld3(tgt, j), n = ld3(nrm, j);
    r.a[0] = n.x; r.a[1] = n.y; r.a[2] = n.z;
    r.a[3] = n.z * s.y - n.y * s.z;
    r.a[4] = n.x * s.z - n.z * s.x;
    r.a[5] = n.y * s.x - n.x * s.y;
    r.b = (n.x * (d.x - s.x) + n.y * (d.y - s.y)) + n.z * (d.z - s.z);
    r.valid = true;
    return r;
}

// ------------------------------------------------------------------ K: brute force (verifier / tiny inputs)
__global__ __launch_bounds__(KNN_T) void knn1_brute_k(const float *__restrict__ src, const int32_t *__restrict__ d_ns,
                                                      const float *__restrict__ tgt, const int32_t *__restrict__ d_nt,
                                                      int nsplit, unsigned long long *__restrict__ best) {
    const int ns = *d_ns, nt = *d_nt;
    const int i = blockIdx.x * KNN_T + threadIdx.x;
    if (blockIdx.x * KNN_T >= ns) return;
    const int chunk = (nt + nsplit - 1) / nsplit;
    const int j0 = blockIdx.y * chunk, j1 = min(nt, j0 + chunk);
    const bool ok = i < ns;
    const f3 s = ok ? ld3(src, i) : f3{0.0f, 0.0f, 0.0f};
    float bd = INFINITY;
    int bi = 0;
    for (int j = j0; j < j1; ++j) {  // wave-uniform j: tgt[j] lives in SGPRs
        const float d = dist2(s, tgt[3 * j], tgt[3 * j + 1], tgt[3 * j + 2]);
        if (d < bd) { bd = d; bi = j; }  // strict: lowest index wins
    }
    if (ok && bd < INFINITY) atomicMin(best + i, pack_key(bd, bi));
}

// ------------------------------------------------------------------ K: AABB-pruned exact search
// boxes: per chunk 6 floats (lo.xyz, hi.xyz).  One wave covers 64 target points = 64/CHUNK chunks
// (segmented butterfly inside each CHUNK-lane group).
__global__ __launch_bounds__(64) void tgt_boxes_k(const float *__restrict__ tgt, const int32_t *__restrict__ d_nt,
                                                  float *__restrict__ boxes) {
    const int nt = *d_nt;
    const int j = blockIdx.x * 64 + threadIdx.x;
    if (blockIdx.x * 64 >= nt) return;
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    if (j < nt) {
        const f3 p = ld3(tgt, j);
        lo[0] = hi[0] = p.x; lo[1] = hi[1] = p.y; lo[2] = hi[2] = p.z;
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
#pragma unroll
        for (int off = CHUNK / 2; off > 0; off >>= 1) {
            lo[a] = fminf(lo[a], __shfl_xor(lo[a], off, kWave));
            hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], off, kWave));
        }
    }
    if ((threadIdx.x % CHUNK) == 0 && j < nt) {
        float *b = boxes + 6 * (int64_t)(j / CHUNK);
        b[0] = lo[0]; b[1] = lo[1]; b[2] = lo[2]; b[3] = hi[0]; b[4] = hi[1]; b[5] = hi[2];
    }
}

// Cooperative exact search of one 64-point source tile by the KNN_NW waves of a block.  Every wave
// holds the same 64 source points (lane = point); the waves share the work over TARGET chunks:
//   seed   : one real candidate per lane (given index, or the best of a strided sample of the target)
//   coarse : lanes = chunk boxes.  A chunk survives iff the gap between ITS box and the TILE's box is not
//            above the largest seed distance of the tile: 64 boxes per wave-instruction, ~1 instruction
//            sequence per wave for a whole 19 k-point target.
//   fine   : surviving chunks are dealt round-robin to the waves; each is tested against every lane's
//            own bound (lanes = source points) and, if some lane still needs it, scanned: one coalesced
//            load of its CHUNK points, candidates broadcast with v_readlane.
// The lanes' running best lives in LDS as packed keys: waves publish improvements with ds_min_u64 and
// re-read before every fine test, so a hit found by one wave prunes the others' remaining chunks.  A
// stale read only prunes less, never wrongly.  All bounds use the distance's own operation order, so
// bound <= distance holds exactly in fp32 (monotone rounding): the result is the brute-force scan's.
constexpr int POOL = 4096;  // grid search: target points staged in LDS per tile (all window rows together): 64 KiB; two such
                             // blocks share a CU (tools/micro/coresidency.hip: up to 80 KiB each)

// the bucketing camera: rotation / translation world -> camera in project_point's layout, pinhole constants, grid
struct CamK {
    float R[9], T[3], fx, fy, cx, cy;
    int ds, Wd, Hd;
};
struct KnnShared {
    unsigned long long key[64];
    int cnt;
    float tbox[6];             // the source tile's AABB (lo.xyz, hi.xyz)
    // grid search (knn1_loop_k<true>)
    int win[WROWS][64];        // per lane: the packed window rows (LaneWin), from the staging waves
    int wflag[64];             // per lane: rel | full << 2 | window radius << 3
    int centre[64];            // per lane: the window's centre pixel
    CamK cam;                  // the bucketing camera (copied once per block: the proof reads it at LDS, not scalar-cache, latency)
    int band[2 * WBANDS + 1];  // staged bands: first slot x WBANDS, pool offset x WBANDS, pool fill
    int plan[2 * WBANDS];      // per band: its pool offset if it is in the pool and not empty, else INT_MAX | its length (for the staging waves)
    unsigned int plan_ready;   // set (release) by the planning wave once band / plan are written
    float seed[2][64][4];      // the seed for either outcome of the step: target point, reference index bits
    union alignas(16) {
        struct {
            int list[KNN_LIST];
            float rows[NACC][65];  // linearise epilogue: per-point products, padded against bank conflicts
            float part[NACC][16];
        } a;
        float stage[POOL * 4];  // window rows: (x, y, z, reference index bits) per target point
    } u;
};

// Counters (gs_loop_counts): what the DEVICE decided -- {loops prepared, loops whose association ran as a grid search
// (variant launched AND the target's actual count dense enough), loops cut into small tiles, tiles whose point-serial
// search overflowed its pair list and fell back to the tile-level search}.  One atomic per loop from icp_prepare_k (and
// one per overflowing tile: a rare path); tests read them to make sure a run really exercised those paths.
__device__ unsigned int g_loop_counts[4];

#ifdef GS_DIAG_STAMPS
// Diagnostic build only (libgradslam_hip_diag.so, never loaded by the product): per-wave phase stamps.
__device__ unsigned long long *g_diag = nullptr;
#define GS_STAMP(slot)                                                                                   \
    do {                                                                                                 \
        if (g_diag && (threadIdx.x & 63) == 0)                                                           \
            g_diag[((size_t)blockIdx.x * KNN_NW + (threadIdx.x >> 6)) * 16 + (slot)] = wall_clock64();   \
    } while (0)
#define GS_COUNT(slot, v)                                                                                \
    do {                                                                                                 \
        if (g_diag && (threadIdx.x & 63) == 0)                                                           \
            g_diag[((size_t)blockIdx.x * KNN_NW + (threadIdx.x >> 6)) * 16 + (slot)] = (v);             \
    } while (0)
#define GS_TICK(var) const unsigned int var = (unsigned int)wall_clock64()
#define GS_ACCUM(acc, t0) acc += (unsigned int)wall_clock64() - (t0)
#else
#define GS_STAMP(slot)
#define GS_COUNT(slot, v)
#define GS_TICK(var)
#define GS_ACCUM(acc, t0)
#endif

// wave-uniform broadcast of lane l's value (v_readlane_b32: no memory round trip)
__device__ __forceinline__ float rlane(float v, int l) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l));
}
// Minimum / maximum over the wave's 64 lanes, uniform result.  Inside each row of 16 lanes by DPP (quad_perm [1,0,3,2],
// [2,3,0,1], row_half_mirror, row_mirror: eight VALU instructions, no LDS), across the four rows through scalars.  The
// __shfl_xor butterfly these replace is six DEPENDENT ds_bpermute round trips (~0.3 us per reduction on the planner's path).
template <int CTRL>
__device__ __forceinline__ int dpp_i(int v) {
    return __builtin_amdgcn_update_dpp(v, v, CTRL, 0xf, 0xf, false);
}
template <class Op>
__device__ __forceinline__ int wave_reduce_i(int v, Op op) {
    v = op(v, dpp_i<0xB1>(v));
    v = op(v, dpp_i<0x4E>(v));
    v = op(v, dpp_i<0x141>(v));
    v = op(v, dpp_i<0x140>(v));
    return op(op(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)),
              op(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
}
template <class Op>
__device__ __forceinline__ float wave_reduce_f(float v, Op op) {
    auto d = [](float x, auto tag) { return __int_as_float(dpp_i<decltype(tag)::value>(__float_as_int(x))); };
    v = op(v, d(v, std::integral_constant<int, 0xB1>{}));
    v = op(v, d(v, std::integral_constant<int, 0x4E>{}));
    v = op(v, d(v, std::integral_constant<int, 0x141>{}));
    v = op(v, d(v, std::integral_constant<int, 0x140>{}));
    return op(op(rlane(v, 0), rlane(v, 16)), op(rlane(v, 32), rlane(v, 48)));
}
__device__ __forceinline__ float wave_min_f(float v) {
    return wave_reduce_f(v, [](float a, float b) { return fminf(a, b); });
}
__device__ __forceinline__ float wave_max_f(float v) {
    return wave_reduce_f(v, [](float a, float b) { return fmaxf(a, b); });
}

// Test the n (<= 64) target points held one per lane in (px,py,pz) with target index pj against the
// lane's source point s; branch-free (distance, index) lexicographic update.
__device__ __forceinline__ void scan_held(const f3 s, const float px, const float py, const float pz, const int pj,
                                          const int n, float &bd, int &bi) {
    for (int k = 0; k < n; ++k) {
        const float d = dist2(s, rlane(px, k), rlane(py, k), rlane(pz, k));
        const int j = __builtin_amdgcn_readlane(pj, k);
        const bool better = (d < bd) | ((d == bd) & (j < bi));
        bd = better ? d : bd;
        bi = better ? j : bi;
    }
}

__device__ __forceinline__ void key_unpack(unsigned long long k, float &bd, int &bi) {
    bd = (k == KEY_NONE) ? INFINITY : bitsf((uint32_t)(k >> 32));
    bi = (k == KEY_NONE) ? 0x7fffffff : (int)(uint32_t)(k & 0xffffffffu);
}

// First association with pixel hints: every lane looks at the targets bucketed on the (2R+1)^2 ds-grid
// pixels around its own pixel (scan order = pixel order, pix_start = first slot per pixel), the window
// pixels shared over the waves.  A projective guess used as a SEED only: it hands the exact search a bound
// that is already the true nearest distance for almost every lane.
__device__ __forceinline__ void knn_window_seed(KnnShared &sh, const f3 s, const bool ok, const int i,
                                                const gs_icp_hints &h, const int nt) {
    constexpr int R = 2, WIN = (2 * R + 1) * (2 * R + 1), CAP = 4;  // at most CAP targets per window pixel
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (wave == 0) sh.key[lane] = KEY_NONE;
    __syncthreads();
    float bd = INFINITY;
    int bi = 0x7fffffff;
    const int npix = h.grid_w * h.grid_h;
    const int p = ok ? min(max(h.src_pix[i], 0), npix - 1) : 0;
    const int pr = p / h.grid_w, pc = p - pr * h.grid_w;
    for (int wdx = wave; wdx < WIN; wdx += KNN_NW) {
        const int rr = pr + wdx / (2 * R + 1) - R, cc = pc + wdx % (2 * R + 1) - R;
        if (!ok || rr < 0 || rr >= h.grid_h || cc < 0 || cc >= h.grid_w) continue;
        const int q0 = rr * h.grid_w + cc;
        const int s0 = h.pix_start[q0], s1 = min(h.pix_start[q0 + 1], s0 + CAP);
        for (int slot = s0; slot < s1; ++slot) {
            const f3 q = ld3(h.scan_points, slot);
            const int oj = h.scan_orig[slot];
            const float d = dist2(s, q.x, q.y, q.z);
            const bool better = (d < bd) | ((d == bd) & (oj < bi));
            bd = better ? d : bd;
            bi = better ? oj : bi;
        }
    }
    if (wave == 0 && ok && bd == INFINITY) {  // empty window: the next target in pixel order is a valid seed
        const int slot = min(max(h.pix_start[p], 0), nt - 1);
        const f3 q = ld3(h.scan_points, slot);
        bd = dist2(s, q.x, q.y, q.z);
        bi = h.scan_orig[slot];
    }
    if (ok && bd < INFINITY) atomicMin(&sh.key[lane], pack_key(bd, bi));
    __syncthreads();
}

// The source tile's box over the lanes selected by `act`: one wave per component, published through LDS
// (the caller synchronises before reading sh.tbox).
__device__ __forceinline__ void tile_box(KnnShared &sh, const f3 s, const bool act) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (wave >= 1 && wave <= 6) {
        const int a = wave - 1;
        const float v = (a % 3 == 0) ? s.x : ((a % 3 == 1) ? s.y : s.z);
        const float r = (a < 3) ? wave_min_f(act ? v : INFINITY) : wave_max_f(act ? v : -INFINITY);
        if (lane == 0) sh.tbox[a] = r;
    }
}

// Per-lane examined slot ranges of the grid search (three window rows, chunk aligned: a chunk whose first slot
// lies in a range lies in it entirely).
struct LaneWin {
    // per window row: first chunk << 9 | number of chunks (<= POOL / CHUNK < 512); 0 = empty
    __device__ __forceinline__ static int pack(int lo, int n) { return n > 0 ? ((lo / CHUNK) << 9) | ((n + CHUNK - 1) / CHUNK) : 0; }
    __device__ __forceinline__ static int lo(int r) { return (r >> 9) * CHUNK; }
    __device__ __forceinline__ static int len(int r, int nt) { return min((r & 511) * CHUNK, nt - lo(r)); }  // slots
    // does the window of point `who` (rows in sh.win) contain the chunk that starts at `slot`?
    __device__ __forceinline__ static bool covers(const KnnShared &sh, int who, int slot) {
        const int c = slot / CHUNK;
        // (the empty asm ties the LDS reads to this call: hoisted out of the search loops the rows would cost the
        // registers that decide whether two blocks share a CU; a few LDS reads per box test are cheap on this rare path)
        asm volatile("" : "+v"(who));
        bool in = false;
#pragma unroll
        for (int r = 0; r < WROWS; ++r) {
            const int w = sh.win[r][who];
            in |= (unsigned)(c - (w >> 9)) < (unsigned)(w & 511);
        }
        return in;
    }
};
__device__ __forceinline__ int wave_min_i(int v) {
    return wave_reduce_i(v, [](int a, int b) { return min(a, b); });
}
__device__ __forceinline__ int wave_max_i(int v) {
    return wave_reduce_i(v, [](int a, int b) { return max(a, b); });
}
__device__ __forceinline__ int sel4(int k, int a0, int a1, int a2, int a3) { return k == 0 ? a0 : (k == 1 ? a1 : (k == 2 ? a2 : a3)); }

// Exact search over the chunk boxes for the lanes selected by `act`, seeded by sh.key (tile box in sh.tbox, both
// visible): coarse pass with the tile's box and loosest bound, fine pass with per-lane bounds (see knn_tile).
// GRID: chunks inside a lane's own window `win` were examined already and are skipped for that lane.
// Ends with a barrier.
template <bool GRID>
__device__ __forceinline__ void knn_prune_search(KnnShared &sh, const f3 s, const bool ok, const bool act,
                                                 const float *__restrict__ scan, const int32_t *__restrict__ scan_orig,
                                                 const float *__restrict__ boxes, const float *__restrict__ sboxes /* or NULL */,
                                                 const int nt) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int n_scanned = 0;
    // the tile's box (from LDS) and its loosest bound (same 64 points in every wave -> same value)
    float bd0;
    int bi0;
    key_unpack(sh.key[lane], bd0, bi0);
    const float tlx = sh.tbox[0], tly = sh.tbox[1], tlz = sh.tbox[2];
    const float thx = sh.tbox[3], thy = sh.tbox[4], thz = sh.tbox[5];
    const float bdmax = wave_max_f(act ? bd0 : 0.0f);

    const int nchunks = (nt + CHUNK - 1) / CHUNK;
#ifdef GS_DIAG_STAMPS
    unsigned int t_coarse = 0, t_fine = 0, t_bar = 0, n_tested = 0;
#endif
    for (int r0 = 0; r0 < nchunks; r0 += KNN_LIST) {
        if (threadIdx.x == 0) sh.cnt = 0;
        __syncthreads();
        GS_TICK(tc0);
        const int r1 = min(nchunks, r0 + KNN_LIST);
        // coarse: lanes = chunk boxes; box-to-box gap with the distance's accumulation order
        for (int c0 = r0 + wave * 64; c0 < r1; c0 += KNN_NW * 64) {
            if (sboxes) {
                // the 64 chunks of this round are one super-box (c0 is a multiple of SUPER): its box contains theirs, so
                // its gap to the tile's box is, axis by axis, at most theirs and -- same operation order, monotone
                // rounding -- its bound at most each of theirs: above the tile's loosest bound, all 64 are pruned at once
                static_assert(SUPER == 64 && KNN_LIST % SUPER == 0, "one coarse round = one super-box");
                const float *b = sboxes + 6 * (int64_t)__builtin_amdgcn_readfirstlane(c0 / SUPER);
                const float ex = fmaxf(fmaxf(b[0] - thx, tlx - b[3]), 0.0f);
                const float ey = fmaxf(fmaxf(b[1] - thy, tly - b[4]), 0.0f);
                const float ez = fmaxf(fmaxf(b[2] - thz, tlz - b[5]), 0.0f);
                const float lbs = (ex * ex + ey * ey) + ez * ez;
                if (!(lbs <= bdmax)) continue;
            }
            const int c = c0 + lane;
            bool pass = false;
            if (c < r1) {
                const float *b = boxes + 6 * (int64_t)c;
                const float ex = fmaxf(fmaxf(b[0] - thx, tlx - b[3]), 0.0f);
                const float ey = fmaxf(fmaxf(b[1] - thy, tly - b[4]), 0.0f);
                const float ez = fmaxf(fmaxf(b[2] - thz, tlz - b[5]), 0.0f);
                const float lbt = (ex * ex + ey * ey) + ez * ez;
                pass = lbt <= bdmax;
            }
            const unsigned long long m = __ballot(pass);
            if (m) {
                int base = 0;
                if (lane == 0) base = atomicAdd(&sh.cnt, __popcll(m));
                base = __builtin_amdgcn_readfirstlane(base);
                if (pass) sh.u.a.list[base + __popcll(m & ((1ull << lane) - 1ull))] = c;
            }
        }
        GS_ACCUM(t_coarse, tc0);
        GS_TICK(tb0);
        __syncthreads();
        GS_ACCUM(t_bar, tb0);
        GS_TICK(tf0);
        // fine: survivors dealt round-robin to the waves, handled four at a time: ONE round of loads
        // brings the boxes and the 4 x CHUNK candidate points of a group into registers (lane l holds
        // point l%CHUNK of the group's survivor l/CHUNK), then tests and scans run without memory ops
        const int nlist = sh.cnt;
        const int ni = (nlist > wave) ? (nlist - wave + KNN_NW - 1) / KNN_NW : 0;
        constexpr int SEG = 64 / CHUNK;
        for (int g = 0; g < ni; g += SEG) {
            const int seg = lane / CHUNK, idx = g + seg;
            const bool have = idx < ni;
            const int c = have ? sh.u.a.list[wave + KNN_NW * idx] : 0;
            const int j = c * CHUNK + (lane % CHUNK);
            const bool pv = have && j < nt;
            const f3 q = pv ? ld3(scan, j) : f3{0.0f, 0.0f, 0.0f};
            const int pj = pv ? (scan_orig ? scan_orig[j] : j) : 0x7fffffff;
            float b0 = 0, b1 = 0, b2 = 0, b3 = 0, b4 = 0, b5 = 0;
            if (have) {
                const float *b = boxes + 6 * (int64_t)c;
                b0 = b[0]; b1 = b[1]; b2 = b[2]; b3 = b[3]; b4 = b[4]; b5 = b[5];
            }
            const int ng = min(SEG, ni - g);
            for (int sg = 0; sg < ng; ++sg) {
                const int l0 = sg * CHUNK;
                float bd;
                int bi;
                key_unpack(sh.key[lane], bd, bi);
                const float ex = fmaxf(fmaxf(rlane(b0, l0) - s.x, s.x - rlane(b3, l0)), 0.0f);
                const float ey = fmaxf(fmaxf(rlane(b1, l0) - s.y, s.y - rlane(b4, l0)), 0.0f);
                const float ez = fmaxf(fmaxf(rlane(b2, l0) - s.z, s.z - rlane(b5, l0)), 0.0f);
                const float lb = (ex * ex + ey * ey) + ez * ez;
                const int cc = __builtin_amdgcn_readlane(c, l0);
                bool hit;
                if (GRID) {
                    const bool live = act & !LaneWin::covers(sh, lane, cc * CHUNK);  // not examined by this lane yet
                    hit = live & (lb <= bd);
                } else {
                    hit = act & (lb <= bd);
                }
                // skip the chunk iff EVERY lane's bound is strictly above its best
                if (!__any(hit)) continue;
                const int m = min(CHUNK, nt - cc * CHUNK);
                const float bdp = bd;
                const int bip = bi;
                for (int k = 0; k < m; ++k) {
                    const float d = dist2(s, rlane(q.x, l0 + k), rlane(q.y, l0 + k), rlane(q.z, l0 + k));
                    const int jj = __builtin_amdgcn_readlane(pj, l0 + k);
                    const bool better = (d < bd) | ((d == bd) & (jj < bi));
                    bd = better ? d : bd;
                    bi = better ? jj : bi;
                }
                if (ok && (bd < bdp || bi < bip)) atomicMin(&sh.key[lane], pack_key(bd, bi));
                ++n_scanned;
            }
#ifdef GS_DIAG_STAMPS
            n_tested += ng;
#endif
        }
        GS_ACCUM(t_fine, tf0);
        GS_TICK(tb1);
        __syncthreads();
        GS_ACCUM(t_bar, tb1);
    }
    GS_COUNT(8, (unsigned long long)t_coarse);
    GS_COUNT(9, (unsigned long long)t_fine);
    GS_COUNT(10, (unsigned long long)t_bar);
    GS_COUNT(11, (unsigned long long)n_tested);
    GS_COUNT(4, (unsigned long long)n_scanned);
    // diagnostic build: survivors of the last round | HW_ID << 16 | XCC_ID << 48 (which CU the block ran on)
    GS_COUNT(5, (unsigned long long)sh.cnt | ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) << 16) |
                    ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 48));
    (void)n_scanned;
}

// Exact search for a FEW points of the tile (bits of `need_mask`) by the whole block, two phases:
//   A  lanes = super-boxes (SUPER chunks each): every point is tested against all of them with its own bound; a
//      survivor becomes a (point, super-box) pair in the LDS list, the others bound the point's certificate radius;
//   B  the pairs are dealt round-robin to the waves: lanes = the super-box's chunks, tested against the point's
//      bound (chunks inside its window are skipped), survivors scanned at once, four per round, lanes = candidates.
// For one or two stragglers this costs a fraction of the tile-level search -- what a converging loop needs once nearly
// every proof holds.  sh.cnt must be zero on entry (all waves past their last use of the
// list's storage); ends with a barrier.  Returns false (block-uniform) when the pair list overflowed: nothing found is
// final then and the caller must search again with knn_prune_search<true>.
__device__ __forceinline__ bool knn_point_search(KnnShared &sh, const f3 s, const unsigned long long need_mask,
                                                 const float *__restrict__ scan, const int32_t *__restrict__ scan_orig,
                                                 const float *__restrict__ boxes, const float *__restrict__ sboxes, const int nt) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nchunks = (nt + CHUNK - 1) / CHUNK, nsb = (nchunks + SUPER - 1) / SUPER;
    for (unsigned long long rest = need_mask; rest; rest &= rest - 1) {  // phase A
        const int L = __builtin_ctzll(rest);
        const float px = rlane(s.x, L), py = rlane(s.y, L), pz = rlane(s.z, L);
        float bd;
        int bi;
        key_unpack(sh.key[L], bd, bi);  // wave-uniform
        for (int b0 = wave * 64; b0 < nsb; b0 += KNN_NW * 64) {
            const int sb = b0 + lane;
            bool hit = false;
            if (sb < nsb) {
                const float *b = sboxes + 6 * (int64_t)sb;
                const float ex = fmaxf(fmaxf(b[0] - px, px - b[3]), 0.0f);
                const float ey = fmaxf(fmaxf(b[1] - py, py - b[4]), 0.0f);
                const float ez = fmaxf(fmaxf(b[2] - pz, pz - b[5]), 0.0f);
                const float lb = (ex * ex + ey * ey) + ez * ez;
                hit = lb <= bd;
            }
            const unsigned long long m = __ballot(hit);
            if (m) {
                int base = 0;
                if (lane == 0) base = atomicAdd(&sh.cnt, __popcll(m));
                base = __builtin_amdgcn_readfirstlane(base);
                const int at = base + __popcll(m & ((1ull << lane) - 1ull));
                if (hit && at < KNN_LIST) sh.u.a.list[at] = (L << 24) | sb;
            }
        }
    }
    __syncthreads();
    // The list holds KNN_LIST (point, super-box) pairs IN TOTAL -- ~680 super-boxes (700 k targets) per point for six
    // points.  A far-away straggler (huge bound: every super-box passes) on a large target overflows it; which pairs
    // were dropped would depend on the atomics' arrival order, so nothing of this attempt is used: the caller runs the
    // tile-level search for these points instead (block-uniform decision).
    if (sh.cnt > KNN_LIST) {
        if (threadIdx.x == 0) atomicAdd(&g_loop_counts[3], 1u);
        __syncthreads();  // every wave has read sh.cnt before the caller's next search resets it
        return false;
    }
    const int npairs = sh.cnt;
    for (int pi = wave; pi < npairs; pi += KNN_NW) {  // phase B
        const int pr = sh.u.a.list[pi];
        const int L = pr >> 24, sb = pr & 0xffffff;
        const f3 p{rlane(s.x, L), rlane(s.y, L), rlane(s.z, L)};
        float bd;
        int bi;
        key_unpack(sh.key[L], bd, bi);
        const int c = sb * SUPER + lane;
        bool hit = false;
        if (c < nchunks && !LaneWin::covers(sh, L, c * CHUNK)) {
            const float *b = boxes + 6 * (int64_t)c;
            const float ex = fmaxf(fmaxf(b[0] - p.x, p.x - b[3]), 0.0f);
            const float ey = fmaxf(fmaxf(b[1] - p.y, p.y - b[4]), 0.0f);
            const float ez = fmaxf(fmaxf(b[2] - p.z, p.z - b[5]), 0.0f);
            const float lb = (ex * ex + ey * ey) + ez * ez;
            hit = lb <= bd;
        }
        unsigned long long hits = __ballot(hit);
        while (hits) {  // four surviving chunks per round: lane l takes candidate l % CHUNK of the (l / CHUNK)-th of them
            constexpr int SEG = 64 / CHUNK;
            int cc = -1;
#pragma unroll
            for (int q = 0; q < SEG; ++q) {
                const int b0 = hits ? __builtin_ctzll(hits) : -1;
                if (hits) hits &= hits - 1;
                if (lane / CHUNK == q) cc = b0;
            }
            unsigned long long k = KEY_NONE;
            if (cc >= 0) {
                const int j = (sb * SUPER + cc) * CHUNK + (lane % CHUNK);
                if (j < nt) {
                    const f3 q = ld3(scan, j);
                    const float d = dist2(p, q.x, q.y, q.z);
                    k = pack_key(d, scan_orig ? scan_orig[j] : j);
                }
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                const unsigned long long o = __shfl_xor(k, off, kWave);
                k = o < k ? o : k;
            }
            if (k < pack_key(bd, bi)) {  // wave-uniform
                if (lane == 0) atomicMin(&sh.key[L], k);
                key_unpack(k, bd, bi);
            }
        }
    }
    __syncthreads();
    return true;
}

// returns the packed key of lane's point (KEY_NONE when there is no target)
// tgt      : target points in REFERENCE order (seeds are reference indices; so are the returned ones)
// scan     : the same points in the order they are scanned (== tgt when scan_orig is NULL); boxes are
//            built over this order
// scan_orig: reference index of every scan slot, or NULL
__device__ __forceinline__ unsigned long long knn_tile(KnnShared &sh, const f3 s, const bool ok, const int seed_j,
                                                       const float *__restrict__ tgt, const float *__restrict__ scan,
                                                       const int32_t *__restrict__ scan_orig,
                                                       const float *__restrict__ boxes, const float *__restrict__ sboxes,
                                                       const int nt) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    GS_STAMP(0);
    if (wave == 0 && seed_j != -2) {  // -2: keys already seeded in LDS by knn_window_seed
        unsigned long long k0 = KEY_NONE;
        if (ok && seed_j >= 0) {
            const f3 q = ld3(tgt, seed_j);
            k0 = pack_key(dist2(s, q.x, q.y, q.z), seed_j);
        }
        sh.key[lane] = k0;
    }
    tile_box(sh, s, ok);
    __syncthreads();
    if (seed_j == -1) {
        // seed pass 1: a strided sample of KNN_COARSE target points, 16 per wave and step (uniform broadcast)
        const int M = min(nt, KNN_COARSE);
        const float stride = (float)nt / (float)M;
        float bd = INFINITY;
        int bi = 0x7fffffff;
        for (int k0 = wave * 16; k0 < M; k0 += KNN_NW * 16) {
            const int k = k0 + lane;
            const int n = min(16, M - k0);
            const int j = min((int)((float)k * stride), nt - 1);
            const f3 q = (lane < n) ? ld3(scan, j) : f3{0.0f, 0.0f, 0.0f};
            const int oj = (scan_orig && lane < n) ? scan_orig[j] : j;
            scan_held(s, q.x, q.y, q.z, oj, n, bd, bi);
        }
        if (ok && bd < INFINITY) atomicMin(&sh.key[lane], pack_key(bd, bi));
        __syncthreads();
        if (scan_orig == nullptr) {
        // seed pass 2: clouds are image ordered, so index neighbours of the best sample are spatial
        // neighbours: each lane refines over [j*-R, j*+R) of ITS sample, the waves split the offsets
        constexpr int R = 64;
        key_unpack(sh.key[lane], bd, bi);
        const int jstar = bi;
        for (int t = 0; t < 2 * R / KNN_NW; ++t) {
            const int j = min(max(jstar - R + wave * (2 * R / KNN_NW) + t, 0), nt - 1);
            const f3 q = ok ? ld3(tgt, j) : f3{0.0f, 0.0f, 0.0f};
            const float d = dist2(s, q.x, q.y, q.z);
            const bool better = (d < bd) | ((d == bd) & (j < bi));
            bd = better ? d : bd;
            bi = better ? j : bi;
        }
        if (ok && bi != jstar) atomicMin(&sh.key[lane], pack_key(bd, bi));
        __syncthreads();
        }
    }
    GS_STAMP(1);
    knn_prune_search<false>(sh, s, ok, ok, scan, scan_orig, boxes, sboxes, nt);
    GS_STAMP(2);
    GS_STAMP(3);
    return ok ? sh.key[lane] : KEY_NONE;
}

// Device-resident loop state.  Point clouds ping-pong between pts[0..1] and nearest-neighbour arrays
// between best[0..1]; `p_cur` / `b_cur` say which one holds the current cloud.
struct IcpState {
    float T[16];    // accumulated transform
    float dT[16];   // step the next association launch applies
    float cur[44];  // H|g|e|cnt of the current cloud
    float xi[6];
    float damp;
    int p_cur;      // pts[p_cur] = current cloud; the association writes pts[1 - p_cur]
    int b_cur;      // best[b_cur] = NN of the current cloud; the association writes best[1 - b_cur]
    int b_first;    // NN buffer of the cloud the last iteration's first solve used
    int it;
};

// Clouds and nearest-neighbour arrays live in numbered slots.  The plain loops use two of each and
// ping-pong; the taped loops (autograd) give every association launch a slot of its own, so the tape
// IS the loop's working storage and nothing is copied.
struct LoopBufs {
    float *pts;                  // slot s at pts + s * pts_stride (floats)
    unsigned long long *best;    // slot s at best + s * best_stride
    int64_t pts_stride, best_stride;
    __host__ __device__ float *P(int s) const { return pts + s * pts_stride; }
    __host__ __device__ unsigned long long *N(int s) const { return best + s * best_stride; }
};

// Stand-alone pruned search (gs_knn1): no transform, sampled seed pass.
__global__ __launch_bounds__(KNN_BT) void knn1_box_k(const float *__restrict__ src, const int32_t *__restrict__ d_ns,
                                                     const float *__restrict__ tgt, const float *__restrict__ boxes,
                                                     const int32_t *__restrict__ d_nt, unsigned long long *__restrict__ best) {
    __shared__ KnnShared sh;
    const int ns = *d_ns, nt = *d_nt;
    const int tile0 = blockIdx.x * 64;
    if (tile0 >= ns) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = tile0 + lane;
    const bool ok = i < ns;
    const f3 s = ok ? ld3(src, i) : f3{0.0f, 0.0f, 0.0f};
    if (nt <= 0) {
        if (ok && wave == 0) best[i] = KEY_NONE;
        return;
    }
    const unsigned long long key = knn_tile(sh, s, ok, -1, tgt, tgt, nullptr, boxes, nullptr, nt);
    if (ok && wave == 0) best[i] = key;
}

__global__ void knn_unpack_k(const unsigned long long *__restrict__ best, const int32_t *__restrict__ d_ns,
                             float *__restrict__ dist2_out, int64_t *__restrict__ idx) {
    const int ns = *d_ns;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < ns; i += gridDim.x * blockDim.x) {
        const unsigned long long k = best[i];
        if (dist2_out) dist2_out[i] = bitsf((uint32_t)(k >> 32));
        if (idx) idx[i] = (int64_t)(uint32_t)(k & 0xffffffffu);
    }
}

__global__ void fill_u64_k(unsigned long long *__restrict__ p, int n, unsigned long long v) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) p[i] = v;
}

// ------------------------------------------------------------------ J: stand-alone kernels
// block-level fixed-order reduction of the 29 accumulators -> partials[blockIdx.x]
__device__ __forceinline__ void block_reduce_store(float *acc, float *__restrict__ partials) {
    __shared__ float sm[LIN_T / 64][NACC];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < NACC; ++k) {
        const float v = wave_sum(acc[k]);
        if (lane == 0) sm[wid][k] = v;
    }
    __syncthreads();
    if (threadIdx.x < NACC) {
        float v = 0.0f;
#pragma unroll
        for (int w = 0; w < LIN_T / 64; ++w) v += sm[w][threadIdx.x];
        partials[blockIdx.x * NACC + threadIdx.x] = v;
    }
}

__global__ __launch_bounds__(LIN_T) void linearize_k(const float *__restrict__ src, const int32_t *__restrict__ d_ns,
                                                     const float *__restrict__ tgt, const float *__restrict__ nrm,
                                                     const unsigned long long *__restrict__ best, float thresh,
                                                     float *__restrict__ partials /* gridDim.x x NACC */) {
    const int ns = *d_ns;
    float acc[NACC];
#pragma unroll
    for (int k = 0; k < NACC; ++k) acc[k] = 0.0f;
    for (int i = blockIdx.x * LIN_T + threadIdx.x; i < ns; i += gridDim.x * LIN_T) {
        const Row r = make_row(src, tgt, nrm, best, i, ns, thresh);
        if (r.valid) accumulate_row(r, acc);
    }
    block_reduce_store(acc, partials);
}

// Fixed-order reduction of the per-block partials by a 1024-thread block into acc_sm[NACC]:
// thread (g, k) = (t / 32, t % 32) sums rows g, g+32, g+64, ... of accumulator k (coalesced over k),
// then 29 threads add the 32 group sums in order.  Two short LDS stages, no shuffle chains.
constexpr int RP_LOADS = 16;  // reduce_partials: loads in flight per thread
constexpr int RP_FEW = 10;    // knn1_loop_k for launches of at most 32 * RP_FEW blocks (a 160 x 120 frame: 300): every instruction of its
                              // prologue is executed by sixteen waves on four SIMDs
// The two halves of one round, for a caller that has other loads to put in flight between them (knn1_loop_k: a launch of at
// most 32 * RP_LOADS = 512 rows is ONE round): rp_issue requests thread (g, k)'s rows, rp_finish sums them in the order below.
template <int NL = RP_LOADS>
__device__ __forceinline__ void rp_issue(const float *__restrict__ partials, int nblocks, int b0, float (&a)[RP_LOADS]) {
    // Clamped addresses, unconditional loads; rp_sum masks what lies outside.  (A select -- or a branch -- at the load makes
    // the compiler wait for each value where it is requested: sixteen trips in a row instead of one.)
    // NL < RP_LOADS: the caller knows that nblocks <= 32 NL (the rows beyond are zeros in either form: same sums).
    const int kc = min((int)(threadIdx.x & 31), NACC - 1), last = max(nblocks - 1, 0);
#pragma unroll
    for (int u = 0; u < RP_LOADS; ++u) a[u] = u < NL ? partials[min(b0 + 32 * u, last) * NACC + kc] : 0.0f;
}
template <int NL = RP_LOADS>
__device__ __forceinline__ float rp_sum(const float (&a)[RP_LOADS], int nblocks, int b0, float v) {
    const int k = threadIdx.x & 31;
#pragma unroll
    for (int u = 0; u < NL; ++u) v += (k < NACC && b0 + 32 * u < nblocks) ? a[u] : 0.0f;
    return v;
}
// knn1_loop_k's form: rows b0, b0 + 32, ... b0 + 32 (NL - 1) exist and hold zeros where no block wrote (partial_rows_alloc,
// icp_prepare_k), so nothing is clamped or masked -- three instructions per row instead of seven, and each of them is
// executed by up to sixteen waves on four SIMDs.  Lanes k >= NACC sum words of the neighbouring row: never read.
template <int NL>
__device__ __forceinline__ void rp_issue_padded(const float *__restrict__ partials, int b0, float (&a)[RP_LOADS]) {
    const float *p = partials + b0 * NACC + (threadIdx.x & 31);
#pragma unroll
    for (int u = 0; u < RP_LOADS; ++u) a[u] = u < NL ? p[32 * NACC * u] : 0.0f;
}
template <int NL>
__device__ __forceinline__ float rp_sum_padded(const float (&a)[RP_LOADS]) {
    float v = 0.0f;
#pragma unroll
    for (int u = 0; u < NL; ++u) v += a[u];
    return v;
}
__device__ __forceinline__ void rp_finish(float v, float *acc_sm) {
    __shared__ float stage[32][33];
    const int k = threadIdx.x & 31, g = threadIdx.x >> 5;  // blockDim.x == 1024 -> g in [0, 32)
    stage[k][g] = v;
    __syncthreads();
    if (threadIdx.x < NACC) {
        float t = 0.0f;
#pragma unroll
        for (int q = 0; q < 32; ++q) t += stage[threadIdx.x][q];
        acc_sm[threadIdx.x] = t;
    }
    __syncthreads();
}
// rp_finish for a block in which only wave 0 needs the sums (knn1_loop_k<true>: the other fifteen waves stage the search's
// windows meanwhile, and the two block barriers above kept them from starting for 1.2 us -- phase stamps, r04a).  Every
// wave leaves its 128 sums in LDS, waits for its OWN LDS writes (the caller's state words among them) and counts itself in;
// wave 0 waits for the count, then adds the 32 group sums in the same order as rp_finish.  `cnt` must be zero and visible
// to all waves before the first of them gets here (the caller's raw barrier at kernel start).
__device__ __forceinline__ void rp_finish_wave0(float v, int g, float v2, int g2, float *acc_sm, unsigned int *cnt) {
    __shared__ float stage[32][33];
    const int k = threadIdx.x & 31;
    if (g >= 0) stage[k][g] = v;     // (g, g2: the row groups this thread summed, -1 = none -- knn1_loop_k hands the planning
    if (g2 >= 0) stage[k][g2] = v2;  // waves' groups to two of the waves that only wait)
    if (threadIdx.x >= 64) {
        if ((threadIdx.x & 63) == 0) __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    } else {
        while (__hip_atomic_load(cnt, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < (unsigned int)(KNN_NW - 1)) __builtin_amdgcn_s_sleep(1);
        if (threadIdx.x < NACC) {
            float t = 0.0f;
#pragma unroll
            for (int q = 0; q < 32; ++q) t += stage[threadIdx.x][q];
            acc_sm[threadIdx.x] = t;
        }
    }
}
__device__ __forceinline__ void reduce_partials(const float *__restrict__ partials, int nblocks, float *acc_sm) {
    const int g = threadIdx.x >> 5;
    float v = 0.0f;
    // The rows were written by the previous launch on other CUs: every read is a trip to memory-side
    // cache (~1.5 us), so what matters is how many of them are in flight -- sixteen per thread and round: the 512
    // rows of a full chip (two tiles per CU) in ONE round.
    for (int b0 = g; b0 < nblocks; b0 += 32 * RP_LOADS) {
        float a[RP_LOADS];
        rp_issue(partials, nblocks, b0, a);
        v = rp_sum(a, nblocks, b0, v);
    }
    rp_finish(v, acc_sm);
}

// H (6x6 symmetric) | g | e | cnt from the 29 accumulators
__device__ __forceinline__ void expand44(const float *acc, float *out44) {
    int q = 0;
    for (int u = 0; u < 6; ++u)
        for (int v = u; v < 6; ++v) { out44[6 * u + v] = acc[q]; out44[6 * v + u] = acc[q]; ++q; }
    for (int u = 0; u < 6; ++u) out44[36 + u] = acc[21 + u];
    out44[42] = acc[27];
    out44[43] = acc[28];
}

__global__ __launch_bounds__(1024) void finalize44_k(const float *__restrict__ partials, int nblocks, float *__restrict__ out44) {
    __shared__ float acc[NACC];
    reduce_partials(partials, nblocks, acc);
    if (threadIdx.x == 0) expand44(acc, out44);
}

__global__ void icp_rows_k(const float *__restrict__ src, const int32_t *__restrict__ d_ns, const float *__restrict__ tgt,
                           const float *__restrict__ nrm, const unsigned long long *__restrict__ best, float thresh,
                           float *__restrict__ A, float *__restrict__ bvec, uint8_t *__restrict__ keep) {
    const int ns = *d_ns;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < ns; i += gridDim.x * blockDim.x) {
        const Row r = make_row(src, tgt, nrm, best, i, ns, thresh);
        keep[i] = r.valid ? 1 : 0;
#pragma unroll
        for (int u = 0; u < 6; ++u) A[6 * (int64_t)i + u] = r.valid ? r.a[u] : 0.0f;
        bvec[i] = r.valid ? r.b : 0.0f;
    }
}

// adjoint of linearize
__global__ void linearize_bwd_k(const float *__restrict__ src, const int32_t *__restrict__ d_ns,
                                const float *__restrict__ tgt, const float *__restrict__ nrm,
                                const unsigned long long *__restrict__ best, float thresh,
                                const float *__restrict__ gout /*43: Hbar36 | gbar6 | ebar*/, float *__restrict__ g_src,
                                float *__restrict__ g_tgt, float *__restrict__ g_nrm) {
    const int ns = *d_ns;
    __shared__ float G[43];
    if (threadIdx.x < 43) G[threadIdx.x] = gout[threadIdx.x];
    __syncthreads();
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < ns; i += gridDim.x * blockDim.x) {
        const Row r = make_row(src, tgt, nrm, best, i, ns, thresh);
        if (!r.valid) {
            if (g_src) st3(g_src, i, f3{0, 0, 0});
            continue;
        }
        const uint32_t j = (uint32_t)(best[i] & 0xffffffffu);
        const f3 s = ld3(src, i), d = ld3(tgt, j), n = ld3(nrm, j);
        float ab[6];
#pragma unroll
        for (int u = 0; u < 6; ++u) {
            float v = G[36 + u] * r.b;
#pragma unroll
            for (int w = 0; w < 6; ++w) v += (G[6 * u + w] + G[6 * w + u]) * r.a[w];
            ab[u] = v;
        }
        float bb = 2.0f * G[42] * r.b;
#pragma unroll
        for (int u = 0; u < 6; ++u) bb += G[36 + u] * r.a[u];
        // a = [n ; s x n]
        const f3 an{ab[0], ab[1], ab[2]}, ac{ab[3], ab[4], ab[5]};
        // c = s x n: s_bar = n x c_bar ; n_bar += c_bar x s
        f3 sb{n.y * ac.z - n.z * ac.y, n.z * ac.x - n.x * ac.z, n.x * ac.y - n.y * ac.x};
        f3 nb{an.x + (ac.y * s.z - ac.z * s.y), an.y + (ac.z * s.x - ac.x * s.z), an.z + (ac.x * s.y - ac.y * s.x)};
        // b = n.(d - s)
        sb.x -= bb * n.x; sb.y -= bb * n.y; sb.z -= bb * n.z;
        nb.x += bb * (d.x - s.x); nb.y += bb * (d.y - s.y); nb.z += bb * (d.z - s.z);
        if (g_src) st3(g_src, i, sb);
        if (g_tgt) {
            atomicAdd(g_tgt + 3 * (int64_t)j, bb * n.x);
            atomicAdd(g_tgt + 3 * (int64_t)j + 1, bb * n.y);
            atomicAdd(g_tgt + 3 * (int64_t)j + 2, bb * n.z);
        }
        if (g_nrm) {
            atomicAdd(g_nrm + 3 * (int64_t)j, nb.x);
            atomicAdd(g_nrm + 3 * (int64_t)j + 1, nb.y);
            atomicAdd(g_nrm + 3 * (int64_t)j + 2, nb.z);
        }
    }
}

__global__ void transform_k(const float *__restrict__ pts, const int32_t *__restrict__ d_n, const float *__restrict__ T,
                            float *__restrict__ out) {
    const int n = *d_n;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) st3(out, i, xform(T, ld3(pts, i)));
}

// ------------------------------------------------------------------ X: O(1) algebra on one lane
// x = (H + damp I)^-1 g.  H, g arrive in fp32 and the damping is added in fp32 like the reference
// (odometry/icputils.py:86-87); the 6x6 system itself is solved in fp64 with partial pivoting, which
// removes the solver's own rounding from the parity budget (the reference inverts in fp32 LAPACK).
// The augmented matrix lives in caller-provided memory (LDS in the kernels: dynamic indexing there costs neither
// registers nor scratch -- this rare path must not inflate the register budget of the association kernel).
__device__ __noinline__ void solve6_lu(const float *H, const float *g, float damp, float *x, double *Mbuf /* 42 */) {
    double (*M)[7] = reinterpret_cast<double (*)[7]>(Mbuf);
    // every loop stays a loop (#pragma nounroll): this is the rare path, it must stay small in registers
#pragma nounroll
    for (int i = 0; i < 6; ++i) {
#pragma nounroll
        for (int j = 0; j < 6; ++j) M[i][j] = (double)(i == j ? H[6 * i + j] + damp : H[6 * i + j]);
        M[i][6] = (double)g[i];
    }
#pragma nounroll
    for (int c = 0; c < 6; ++c) {
        int p = c;
        double big = fabs(M[c][c]);
#pragma nounroll
        for (int r = c + 1; r < 6; ++r)
            if (fabs(M[r][c]) > big) { big = fabs(M[r][c]); p = r; }
        if (p != c) {
#pragma nounroll
            for (int k = 0; k < 7; ++k) { const double t = M[c][k]; M[c][k] = M[p][k]; M[p][k] = t; }
        }
        const double piv = M[c][c];
#pragma nounroll
        for (int r = c + 1; r < 6; ++r) {
            const double f = M[r][c] / piv;
#pragma nounroll
            for (int k = c; k < 7; ++k) M[r][k] -= f * M[c][k];
        }
    }
#pragma nounroll
    for (int r = 5; r >= 0; --r) {  // the solution overwrites the right-hand side column
        double v = M[r][6];
#pragma nounroll
        for (int k = r + 1; k < 6; ++k) v -= M[r][k] * M[k][6];
        M[r][6] = v / M[r][r];
    }
#pragma nounroll
    for (int i = 0; i < 6; ++i) x[i] = (float)M[i][6];
}

// H + damp I is symmetric positive definite in every sane case (H = A^T A, damp > 0): fully unrolled
// fp64 LDL^T in registers (~0.5 us on one lane); anything else falls back to the pivoted elimination.
__device__ void solve6(const float *H, const float *g, float damp, float *x, double *lu_buf) {
    double A[6][6], d[6], y[6];
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j < 6; ++j) A[i][j] = (double)(i == j ? H[6 * i + j] + damp : H[6 * i + j]);
    bool ok = true;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        double dj = A[j][j];
#pragma unroll
        for (int k = 0; k < 6; ++k)
            if (k < j) dj -= A[j][k] * A[j][k] * d[k];
        d[j] = dj;
        ok = ok && (dj > 0.0) && (dj < 1e300);
        const double inv = 1.0 / dj;
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            if (i > j) {
                double v = A[i][j];
#pragma unroll
                for (int k = 0; k < 6; ++k)
                    if (k < j) v -= A[i][k] * A[j][k] * d[k];
                A[i][j] = v * inv;  // L[i][j]
            }
        }
    }
    if (!ok) {
        solve6_lu(H, g, damp, x, lu_buf);
        return;
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) {  // L y = g
        double v = (double)g[i];
#pragma unroll
        for (int k = 0; k < 6; ++k)
            if (k < i) v -= A[i][k] * y[k];
        y[i] = v;
    }
#pragma unroll
    for (int i = 5; i >= 0; --i) {  // L^T x = D^-1 y
        double v = y[i] / d[i];
#pragma unroll
        for (int k = 0; k < 6; ++k)
            if (k > i) v -= A[k][i] * y[k];
        y[i] = v;
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) x[i] = (float)y[i];
}

// reference geometry/se3utils.py:77-115 (xi = [v ; omega]); small-angle branch uses V = I + w^ (sic)
__device__ __noinline__ void se3_exp_dev(const float *xi, float *T) {
    const float v0 = xi[0], v1 = xi[1], v2 = xi[2], w0 = xi[3], w1 = xi[4], w2 = xi[5];
    float Wh[9] = {0.0f, -w2, w1, w2, 0.0f, -w0, -w1, w0, 0.0f};
    const float th = sqrtf(__fmaf_rn(w2, w2, __fmaf_rn(w1, w1, w0 * w0)));
    float R[9], V[9];
    if (th < 1e-6f) {
        for (int i = 0; i < 9; ++i) { R[i] = ((i % 4 == 0) ? 1.0f : 0.0f) + Wh[i]; V[i] = R[i]; }
    } else {
        const float s = sinf(th), c = cosf(th);
        float W2[9];
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j)
                W2[3 * i + j] = dot3_fma(Wh[3 * i], Wh[3 * i + 1], Wh[3 * i + 2], Wh[j], Wh[3 + j], Wh[6 + j]);
        const float A = s / th, Bc = (1.0f - c) / (th * th), C = (th - s) / (th * th * th);
        for (int i = 0; i < 9; ++i) {
            const float e = (i % 4 == 0) ? 1.0f : 0.0f;
            R[i] = (e + A * Wh[i]) + Bc * W2[i];
            V[i] = (e + Bc * Wh[i]) + C * W2[i];
        }
    }
    for (int i = 0; i < 3; ++i) {
        T[4 * i] = R[3 * i]; T[4 * i + 1] = R[3 * i + 1]; T[4 * i + 2] = R[3 * i + 2];
        T[4 * i + 3] = dot3_fma(V[3 * i], V[3 * i + 1], V[3 * i + 2], v0, v1, v2);
    }
    T[12] = 0.0f; T[13] = 0.0f; T[14] = 0.0f; T[15] = 1.0f;
}

enum StepMode { STEP_ADOPT = 0, STEP_LM = 1, STEP_GRAD_B = 2 };

// Tape record of one step (REC_WORDS floats): the IcpState BEFORE the step, then what the step saw.
// The state after step j is the state before step j+1, so record j+1's head doubles as "after j".
constexpr int REC_STATE = 0;     // sizeof(IcpState)/4 words
constexpr int REC_LIN = 96;      // 44 floats: H|g|e|cnt of the cloud the preceding association wrote
constexpr int REC_SLOT = 140;    // slot that association wrote
constexpr int REC_MODE = 141;
constexpr int REC_ACCEPT = 142;
constexpr int REC_WORDS = 160;

struct GradParams {
    // formed in double on the host like the reference's Python scalars, rounded once:
    // lambda_min = 1/lambda_max, range = lambda_max - lambda_min, inv_nu = 1/nu
    float lambda_min, range, B, B2, inv_nu;
};


// x = (H + damp I)^-1 g by ONE WAVE: Gauss-Jordan on the augmented 6x7 system in fp64, element (i, k) in
// lane 8 i + k, rows / columns exchanged with lane permutes.  Takes ~0.5 us like a fully unrolled
// single-lane factorisation but needs a handful of VGPRs instead of ~100, which is what lets the step live in
// the association kernel without costing it its occupancy.  H + damp I is symmetric positive definite in
// every sane case (H = A^T A, damp > 0): no pivoting; a pivot that is not a positive finite number hands the
// system to the pivoted elimination below (one lane, matrix in LDS).  All 64 lanes must call this.
__device__ __forceinline__ double shfl_d(double v, int src) {
    const int lo = __shfl(__double2loint(v), src, kWave), hi = __shfl(__double2hiint(v), src, kWave);
    return __hiloint2double(hi, lo);
}
__device__ void solve6_wave(const float *H, const float *g, float damp, float *x, double *lu_buf) {
    const int lane = threadIdx.x & 63, i = lane >> 3, k = lane & 7;
    double a = 0.0;
    if (i < 6 && k < 6) a = (double)(i == k ? H[6 * i + k] + damp : H[6 * i + k]);
    if (i < 6 && k == 6) a = (double)g[i];
    bool ok = true;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        const double piv = shfl_d(a, 8 * j + j);
        ok = ok && (piv > 0.0) && (piv < 1e300);
        const double pr = shfl_d(a, 8 * j + k);                 // pivot row, my column
        const double col = shfl_d(a, 8 * (i < 6 ? i : 0) + j);  // my row, pivot column
        const double inv = 1.0 / piv;
        a = (i == j) ? pr * inv : a - (col * inv) * pr;
    }
    if (i < 6 && k == 6) x[i] = (float)a;
    if (!ok && lane == 0) solve6_lu(H, g, damp, x, lu_buf);     // `ok` is wave-uniform: every lane saw the same pivots
}

// The O(1) step by ONE wave (S and acc in LDS).  The work is spread
// over its lanes where the data is wide -- expanding the 29 sums to H | g | e | n, adopting them, T = dT . T, the
// tape / trace records -- so that the serial part is a handful of scalars:
//   STEP_ADOPT : the look-ahead cloud becomes the current one unconditionally (initial cloud; gradICP's
//                re-linearisation)                          -> solve ; dT = exp(xi)
//   STEP_LM    : look-ahead cloud: accept (adopt, damp/2, T = dT T) or reject (damp*2) -> solve ; dT
//   STEP_GRAD_B: look-ahead error -> damp, sigma ; dT = exp(sigma xi) ; T = dT T ; look-ahead discarded
// `solve` = false for a loop's very last step, whose xi / dT nothing consumes.
__device__ __forceinline__ float expand_elem(const float *acc, int t) {  // element t of the 44 from the 29 sums
    if (t < 36) {
        int u = t / 6, v = t % 6;
        if (u > v) { const int w = u; u = v; v = w; }
        return acc[u * 6 - (u * (u - 1)) / 2 + (v - u)];
    }
    if (t < 42) return acc[21 + (t - 36)];
    return t == 42 ? acc[27] : acc[28];
}
// LDS hand-offs inside ONE wave: its LDS operations execute in issue order, so all that is needed is that the
// compiler keeps them in program order.
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// Called by the block's FIRST WAVE only (all 64 lanes of it): no block barrier inside, so the other waves are free to
// do something else meanwhile (the grid search's staging); the caller synchronises the block afterwards.
__device__ __forceinline__ void step_wave0(IcpState *S, const float *acc, int mode, GradParams gp, float *trace, float *out_T,
                                           int look_slot, float *rec, double *lu_buf, bool solve) {
    __shared__ float lin[44];
    __shared__ float sx[6];
    const int t = threadIdx.x;
    if (t < 44) lin[t] = expand_elem(acc, t);
    wave_sync();
    const float err = S->cur[42], new_err = lin[42];
    const bool lm_accept = new_err < err;
    const bool adopt = mode == STEP_ADOPT || (mode == STEP_LM && lm_accept);
    if (rec) {  // tape: what the look-ahead launch measured, where it wrote, what this step is
        if (t < 44) rec[REC_LIN + t] = lin[t];
        if (t == 44) {
            rec[REC_SLOT] = (float)look_slot;
            rec[REC_MODE] = (float)mode;
            rec[REC_ACCEPT] = (mode == STEP_LM) ? (lm_accept ? 1.0f : 0.0f) : 1.0f;
        }
    }
    if (trace && mode != STEP_ADOPT) {
        float *tr = trace + 48 * S->it;
        if (t < 42) tr[t] = S->cur[t];
        if (t == 42) {
            tr[42] = err; tr[43] = new_err; tr[44] = S->damp; tr[45] = (mode == STEP_LM && !lm_accept) ? 0.0f : 1.0f;
            tr[46] = S->cur[43]; tr[47] = 0.0f;
        }
    }
    if (mode == STEP_GRAD_B) {  // the gates and the damped step: a few scalars, one lane
        if (t == 0) {
            float diff = new_err - err;
            diff = fminf(fmaxf(diff, -70.0f), 70.0f);
            const float damp_new = gp.lambda_min + gp.range / (1.0f + expf((-gp.B) * diff));
            S->damp = S->damp * damp_new;
            const float sig = 1.0f / powf(1.0f + expf((-gp.B2) * diff), gp.inv_nu);
            for (int i = 0; i < 6; ++i) sx[i] = sig * S->xi[i];
            se3_exp_dev(sx, S->dT);
        }
        wave_sync();
    }
    // T = dT . T  (accepted LM step, every gradLM step): one lane per element, torch.mm's fma chain over k
    const bool mul_T = mode == STEP_GRAD_B || (mode == STEP_LM && lm_accept);
    float new_T = 0.0f;
    if (mul_T && t < 16) {
        const int i = t >> 2, j = t & 3;
        float v = S->dT[4 * i] * S->T[j];
        v = __fmaf_rn(S->dT[4 * i + 1], S->T[4 + j], v);
        v = __fmaf_rn(S->dT[4 * i + 2], S->T[8 + j], v);
        v = __fmaf_rn(S->dT[4 * i + 3], S->T[12 + j], v);
        new_T = v;
    }
    wave_sync();  // the old state has been read
    if (adopt && t < 44) S->cur[t] = lin[t];
    if (mul_T && t < 16) S->T[t] = new_T;
    if (t == 0) {
        if (mode != STEP_ADOPT) S->b_first = S->b_cur;  // the neighbour array the iteration's first solve used
        if (adopt) {
            S->p_cur = look_slot >= 0 ? look_slot : 1 - S->p_cur;
            S->b_cur = look_slot >= 0 ? look_slot : 1 - S->b_cur;
        }
        if (mode == STEP_ADOPT) S->b_first = S->b_cur;
        if (mode == STEP_LM) S->damp = lm_accept ? S->damp / 2.0f : S->damp * 2.0f;
        if (mode != STEP_ADOPT) S->it += 1;
    }
    wave_sync();
    GS_STAMP(13);  // (diagnostic build, wave 0: decision taken, state updated)
    if (out_T && t < 16) out_T[t] = S->T[t];
    if (solve && mode != STEP_GRAD_B) {  // wave-uniform
        solve6_wave(S->cur, S->cur + 36, S->damp, S->xi, lu_buf);
        wave_sync();
        GS_STAMP(14);  // (solved)
        if (t == 0) se3_exp_dev(S->xi, S->dT);
    }
    wave_sync();
}

__global__ __launch_bounds__(1024) void icp_step_k(IcpState *__restrict__ Sg, const float *__restrict__ partials, int nblocks,
                                                  int mode, GradParams gp, float *__restrict__ trace /* or NULL */,
                                                  float *__restrict__ out_T, int look_slot,
                                                  float *__restrict__ rec /* this step's tape record or NULL */, int solve,
                                                  const float *__restrict__ compose_right, float *__restrict__ compose_out) {
    __shared__ float acc[NACC];
    __shared__ double lu_sm[42];
    __shared__ IcpState st;  // work on an LDS copy: ~200 dependent accesses at LDS, not HBM, latency
    constexpr int kWords = sizeof(IcpState) / 4;
#ifdef GS_DIAG_STAMPS
    if (g_diag && threadIdx.x == 0) g_diag[0] = wall_clock64();
#endif
    if (threadIdx.x < kWords) reinterpret_cast<int *>(&st)[threadIdx.x] = reinterpret_cast<const int *>(Sg)[threadIdx.x];
    reduce_partials(partials, nblocks, acc);  // ends with a barrier: st and acc are visible
    // (from the global copy: wave 0 is about to change the LDS one)
    if (rec && threadIdx.x < kWords) reinterpret_cast<int *>(rec)[REC_STATE + threadIdx.x] = reinterpret_cast<const int *>(Sg)[threadIdx.x];
#ifdef GS_DIAG_STAMPS
    if (g_diag && threadIdx.x == 0) g_diag[1] = wall_clock64();
#endif
    if (threadIdx.x < 64) step_wave0(&st, acc, mode, gp, trace, out_T, look_slot, rec, lu_sm, solve != 0);
#ifdef GS_DIAG_STAMPS
    if (g_diag && threadIdx.x == 0) g_diag[2] = wall_clock64();
#endif
    __syncthreads();
    if (threadIdx.x < kWords) {
        const int v = reinterpret_cast<const int *>(&st)[threadIdx.x];
        reinterpret_cast<int *>(Sg)[threadIdx.x] = v;
        if (rec) reinterpret_cast<int *>(rec)[REC_WORDS + REC_STATE + threadIdx.x] = v;  // head of the next record = state after
    }
    if (compose_out && threadIdx.x == 0) compose44(st.T, compose_right, compose_out);  // e.g. T . previous pose
}

// Association launch of the loops, with the PRECEDING step folded into its prologue.
// A tiny dependent kernel costs ~4.5 us of stream time on this part however little it computes, so the loop's
// O(1) step (reduce the previous launch's partial sums, LM / gradLM decision, 6x6 solve, exp) is not a launch
// of its own: every block of the next association recomputes it from the previous launch's outputs (S_in,
// partials_in -- complete and visible at kernel start, no inter-block hand-off inside a launch) into LDS, and
// block 0 alone publishes the new state (S_out, tape record, trace, out_T).  State and partial sums are
// double-buffered across launches so that no block reads what another block of the same launch writes.
// Then: in = (first ? user source : pts[p_cur]) transformed by dT, out = pts[out_slot], NN -> best[out_slot]
// (out_slot < 0: the other one of the two ping-pong slots).  Seed: the current cloud's NN of the same source index
// when there is one, else the sampled seed pass.
//
// GRID (all search hints given): the association is a GRID SEARCH WITH A GEOMETRIC PROOF.
//   window : the target is bucketed by ds-grid pixel of the camera it was selected with (scan order, hints.pix_start).
//            Every lane examines ALL targets of the 3 x 3 pixels around the pixel its point projects to (three
//            contiguous slot ranges, widened to whole chunks), staged through LDS by coalesced loads issued before the
//            folded step, so they cost no time.  The lanes of a tile move together, so their windows lie in at most
//            four row bands, each one contiguous slot range (row-major pixels), which share a pool of POOL staged points.
//   proof  : every target OUTSIDE the window projects at least 2 ds - 0.5 image pixels from the window's centre pixel,
//            i.e. lies beyond one of four planes through the camera centre; the point's distance to the nearest of those
//            planes bounds its distance to all of them from below (cam_bound2).  A window best strictly inside that
//            bound IS the nearest neighbour, tie-break included, and no box is touched: nothing is carried from launch
//            to launch, the first association of a loop is proven like every other.  Lanes that fail (no map point
//            within centimetres: new image regions, depth edges) take the exact chunk-box search, restricted to them.
// The result is the brute-force scan's in every case; only the cost differs (~100 candidates per point at ten targets
// per pixel instead of ~800 and no box tests).
// What stays the same for every launch of one loop lives in the workspace (written once by icp_prepare_k), not in
// the kernel arguments: at ~100 scalar registers a 1024-thread block no longer shares its CU with a second one (the
// hardware admits floor(800 / (ceil(sgpr / 16) 16 + 16)) waves per SIMD: 8 up to 80 SGPRs, 7 from 82 on -- whatever
// the compiler's occupancy estimate says), and the association kernel lives on that second block.
struct LoopConst {
    const float *user_src, *tgt, *nrm, *boxes, *sboxes;
    const int32_t *d_ns, *d_nt;
    float *trace, *out_T;
    gs_icp_hints hints;
    GradParams gp;
    float thresh;
    int ns, nt;            // *d_ns, *d_nt as icp_prepare_k found them (one dependent load less at every kernel start)
    int cert_off;          // measurements only (GS_CERT_OFF=1): never trust a proof -> every association searches exactly
    int tile_points;       // source points per block (lanes 0 .. tile_points - 1 of every wave hold one each): 64, or what
                           // gs_set_tile_points forces (tests)
    int grid_variant;         // this loop launches knn1_loop_k<true> (for the loop counters only)
    // the camera the targets were bucketed with (hints.cam_pose / cam_K as icp_prepare_k read them): world -> camera as
    // project_point (gs_project.hpp) applies it, and the pinhole constants.  cam_ok = 0: K is not a plain pinhole
    // matrix (skew, a projective third row ...) -> no geometric proof, every association searches exactly.
    CamK cam;
    int cam_ok;
    int32_t *cells;           // (2, cells_stride): the ds-grid pixel every point of the cloud an association wrote projects to,
    int cells_stride;         // by launch parity -- where the NEXT launch centres its windows (knn1_loop_k)
};

// world point -> camera coordinates of the bucketing camera, with project_point's arithmetic (gs_project.hpp)
__device__ __forceinline__ f3 cam_point(const CamK &k, const f3 p) {
    return f3{dot3_fma(p.x, p.y, p.z, k.R[0], k.R[3], k.R[6]) + k.T[0],
              dot3_fma(p.x, p.y, p.z, k.R[1], k.R[4], k.R[7]) + k.T[1],
              dot3_fma(p.x, p.y, p.z, k.R[2], k.R[5], k.R[8]) + k.T[2]};
}
// the ds-grid pixel (row-major id) whose centre is nearest to the projection of p (clamped into the grid; any value is
// safe: the proof below is evaluated against whatever centre was chosen)
__device__ __forceinline__ int cam_cell(const CamK &k, const f3 p) {
    const f3 q = cam_point(k, p);
    const float zs = (q.z != 0.0f) ? q.z : 1.0f;
    const float ds = (float)k.ds;
    const float u = ((k.fx * q.x + k.cx * q.z) / zs) / ds, v = ((k.fy * q.y + k.cy * q.z) / zs) / ds;
    const int cc = (int)fminf(fmaxf(rintf(u), 0.0f), (float)(k.Wd - 1));
    const int cr = (int)fminf(fmaxf(rintf(v), 0.0f), (float)(k.Hd - 1));
    return cr * k.Wd + cc;
}
// GEOMETRIC PROOF.  Squared lower bound on the distance from s to every target OUTSIDE the (2R+1)^2 grid pixels around
// `centre`.  A target sits in grid pixel (r, c) iff its projection (u, v) rounds to the image pixel (r ds, c ds), so
// |u - c ds| <= 0.5 and |v - r ds| <= 0.5 (+ ~1e-3 of fp32 error in the bucketing's own projection).  A target outside the
// window therefore has u >= U+ = (cc + R + 1) ds - 0.5, or u <= U- = (cc - R - 1) ds + 0.5, or the same in v.  With
// z > 0 (only points in front of the camera are targets), u >= U+ means fx x + (cx - U+) z >= 0: a half-space whose
// boundary plane passes through the camera centre -- and likewise for the other three sides.  The distance from s to a
// half-space it is not in is the distance to its plane; the minimum over the (up to four) sides that can hold targets
// at all -- beyond the image border there are none -- bounds the distance to every outside target from below.  Rigid
// transforms preserve distances, so the bound is evaluated in camera coordinates.  Margins: 0.52 instead of 0.5 px,
// 0.2 % + 10 um off the bound (the plane normals are normalised with the hardware's 1-ulp reciprocal square root):
// orders of magnitude above fp32 rounding of the terms.  0 = no proof.
__device__ __forceinline__ float cam_bound2(const CamK &k, const f3 s, const int centre, const int R) {
    const f3 q = cam_point(k, s);
    const int cr = centre / k.Wd, cc = centre - cr * k.Wd;
    const float ds = (float)k.ds;
    const float hw = (float)(R + 1) * ds - 0.52f;
    const float uc = (float)cc * ds, vc = (float)cr * ds;
    float L = INFINITY;
    if (cc + R + 1 < k.Wd) { const float a = k.cx - (uc + hw); L = fminf(L, -(k.fx * q.x + a * q.z) * __builtin_amdgcn_rsqf(k.fx * k.fx + a * a)); }
    if (cc - R - 1 >= 0) { const float a = k.cx - (uc - hw); L = fminf(L, (k.fx * q.x + a * q.z) * __builtin_amdgcn_rsqf(k.fx * k.fx + a * a)); }
    if (cr + R + 1 < k.Hd) { const float a = k.cy - (vc + hw); L = fminf(L, -(k.fy * q.y + a * q.z) * __builtin_amdgcn_rsqf(k.fy * k.fy + a * a)); }
    if (cr - R - 1 >= 0) { const float a = k.cy - (vc - hw); L = fminf(L, (k.fy * q.y + a * q.z) * __builtin_amdgcn_rsqf(k.fy * k.fy + a * a)); }
    L = L * 0.998f - 1e-5f;
    return L > 0.0f ? L * L : 0.0f;  // (NaN compares false: no proof)
}

template <bool GRID, int NL>
// (argument order: what the first batch of requests needs comes first -- the leading sixteen dwords of the kernel arguments
// can be preloaded into SGPRs with the dispatch, -amdgpu-kernarg-preload-count in the Makefile)
__global__ __launch_bounds__(KNN_BT, 8) void knn1_loop_k(const LoopConst *__restrict__ C, const IcpState *__restrict__ S_in,
                                                         const float *__restrict__ partials_in, const int32_t *__restrict__ pix_ws,
                                                         const int32_t *__restrict__ cells_in, int cap, int tile_points,
                                                         int phase /* first | launch parity << 1 */, int step_mode, int look_slot,
                                                         int nblocks_in, IcpState *__restrict__ S_out, float *__restrict__ rec, int out_slot,
                                                         LoopBufs B, float *__restrict__ partials /* gridDim.x x NACC */,
                                                         const float *__restrict__ user_src) {
    const int first = phase & 1, par = phase >> 1;
    __shared__ KnnShared sh;
    __shared__ IcpState st_sm;
    __shared__ float acc_sm[NACC];
    __shared__ double lu_sm[42];
    constexpr int kWords = sizeof(IcpState) / 4;
    GS_STAMP(6);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int tile0 = blockIdx.x * tile_points;
    const int i = tile0 + lane;
    constexpr bool grid = GRID;  // (the host launches this variant only with complete hints and camera)
    // ---- Everything whose address follows from the kernel's ARGUMENTS is requested here, in one go, before anything is
    // waited for: the state, the partial rows of the folded step, and (GRID) what the staging waves need first -- the
    // lane's own pixel (the workspace's copy of hints.src_pix: pix_ws), the pixel its point projected to in the previous
    // launch (cells_in), wave 2's seed keys and its copy of the camera constants.  Every trip to memory at kernel start costs
    // 1.5-2 us (the data was written by other XCDs' CUs).  Until round 3's last session these requests stood behind the
    // loop constants (C->ns for the bounds, C->hints.* / C->cells for the addresses: a trip of their own), the state's load
    // was waited for on the spot (another, in waves 0 and 1), and the pointers taken from the constants made FLAT loads,
    // which every later wait for a scalar load also waits for (the compiler's s_waitcnt vmcnt(0) lgkmcnt(0)): four trips in
    // a row before the first window centre was known (phase stamps: 3.2 us after kernel entry).  Now the indices are
    // clamped to the arrays' capacity (cap: all of them are the workspace's own, sized by it) instead of tested against ns.
    // No branch stands between these requests and nothing is tested on them before all are out: the compiler places a load
    // where its scheduling region first needs it, and sinks a load below a branch whose other side does not use it -- with the
    // block's early exit tested first, every request stood behind the trip for ns again.  Hence: addresses nobody needs are
    // clamped to something harmless instead of branched around; ns / nt come by a VECTOR load (lane & 1 picks) in the same
    // batch instead of the scalar load the compiler would issue only where the exit test wants it; and the empty asm below
    // names every requested value, so that none of the requests can move past it.
    __shared__ unsigned int rp_cnt;  // GRID: waves whose part of the row sums (and of the state) is in LDS (rp_finish_wave0)
    if constexpr (GRID) {
        // rp_cnt = 0 must be visible to every wave before any of them counts itself in: wave 0 waits for its LDS write, and
        // all meet at a RAW barrier (no fence needed, nothing else is in flight yet; at kernel start the waves of a block
        // arrive within ~0.1 us of each other)
        if (threadIdx.x == 0) { rp_cnt = 0; sh.plan_ready = 0; }
        if (wave == 0) __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0)
        __builtin_amdgcn_s_barrier();
    }
    const int ic = min(i, cap - 1);
    const int st_w = reinterpret_cast<const int *>(S_in)[min((int)threadIdx.x, kWords - 1)];
    float rp_a[RP_LOADS];
    // (consumed only if a step is folded in: then the launch has <= 512 blocks, one round; ten loads cover the 300 rows of a
    // 160 x 120 frame -- every instruction here is executed by sixteen waves on four SIMDs)
    // (NL: the host instantiates the ten-load form for launches of <= 320 blocks)
    // GRID: waves 1 and 2 (the planner and the lanes' wave: the critical path of the prologue) sum no rows; their four row
    // groups are the second duty of waves 14 and 15, which otherwise only wait for the plan.  (Same groups, same order
    // inside each: the sums do not change.)
    const int rp_g = (GRID && (wave == 1 || wave == 2)) ? -1 : (int)(threadIdx.x >> 5);
    const int rp_g2 = (GRID && wave >= 14) ? (int)(threadIdx.x >> 5) - 26 : -1;
    float rp_b[RP_LOADS];
#pragma unroll
    for (int u = 0; u < RP_LOADS; ++u) { rp_a[u] = 0.0f; rp_b[u] = 0.0f; }
    if (rp_g >= 0) rp_issue_padded<NL>(partials_in, rp_g, rp_a);  // (wave-uniform branches around loads only: nothing is waited for inside)
    if (rp_g2 >= 0) rp_issue_padded<NL>(partials_in, rp_g2, rp_b);
    static_assert(offsetof(LoopConst, nt) == offsetof(LoopConst, ns) + 4, "ns | nt are read as a pair");
    const int nn = reinterpret_cast<const int *>(&C->ns)[lane & 1];
    int e_h = 0, e_c = -1, e_cam = 0;
    unsigned long long e_k0 = 0, e_k1 = 0;
    if constexpr (GRID) {
        e_h = pix_ws[ic];
        e_c = cells_in[ic];
        // wave 2's seeds for either outcome of the step: outside tape mode the two neighbour arrays are slots 0 and 1 (which
        // of them is current is decided below, when the state has arrived); tape mode names the slots in the state
        const int ik = wave == 2 ? ic : 0;
        e_k0 = B.N(0)[ik]; e_k1 = B.N(1)[ik];
        e_cam = reinterpret_cast<const int *>(&C->cam)[min(lane, (int)(sizeof(CamK) / 4) - 1)];
        asm volatile("" ::"v"(e_h), "v"(e_c), "v"(e_cam), "v"((unsigned)e_k0), "v"((unsigned)(e_k0 >> 32)),
                     "v"((unsigned)e_k1), "v"((unsigned)(e_k1 >> 32)));
    }
    if constexpr (GRID)
        asm volatile("" ::"v"(rp_b[0]), "v"(rp_b[1]), "v"(rp_b[2]), "v"(rp_b[3]), "v"(rp_b[4]), "v"(rp_b[5]), "v"(rp_b[6]), "v"(rp_b[7]), "v"(rp_b[8]),
                     "v"(rp_b[9]), "v"(rp_b[10]), "v"(rp_b[11]), "v"(rp_b[12]), "v"(rp_b[13]), "v"(rp_b[14]), "v"(rp_b[15]));
    asm volatile("" ::"v"(st_w), "v"(nn), "v"(rp_a[0]), "v"(rp_a[1]), "v"(rp_a[2]), "v"(rp_a[3]), "v"(rp_a[4]), "v"(rp_a[5]), "v"(rp_a[6]),
                 "v"(rp_a[7]), "v"(rp_a[8]), "v"(rp_a[9]), "v"(rp_a[10]), "v"(rp_a[11]), "v"(rp_a[12]), "v"(rp_a[13]), "v"(rp_a[14]),
                 "v"(rp_a[15]));
    GS_STAMP(8);  // (diagnostic build: the first batch has arrived)
    const int ns = __builtin_amdgcn_readlane(nn, 0), nt = __builtin_amdgcn_readlane(nn, 1);
    const bool ok = lane < tile_points && i < ns;
    const bool tile_live = tile0 < ns && nt > 0;
    // The launch covers the cloud's CAPACITY; blocks beyond its actual size leave at once.  Their partial rows are zeros
    // (added behind every thread's live rows by the next launch: x + 0 = x).  (Block 0 publishes the state: it always stays.)
    if (tile0 >= ns && blockIdx.x != 0) {
        if (threadIdx.x < NACC) partials[blockIdx.x * NACC + threadIdx.x] = 0.0f;
        return;
    }
    f3 e_pp{0.0f, 0.0f, 0.0f};
    unsigned long long e_ka = 0, e_kb = 0;
    if (GRID && grid && tile_live && wave != 0 && ok) {
        if (first && C->cam_ok) e_pp = ld3(user_src, i);  // (the first launch: no step is folded into it)
        if (wave == 2 && !first) {
            const int ba = S_in->b_cur;
            if (look_slot < 0) { e_ka = ba == 0 ? e_k0 : e_k1; e_kb = ba == 0 ? e_k1 : e_k0; }
            else { e_ka = B.N(ba)[i]; e_kb = B.N(look_slot)[i]; }
        }
    }

    // ---- the O(1) step is wave 0's; GRID: the other fifteen waves meanwhile work out the window of every lane and the
    // row bands of the tile, stage the bands' targets into LDS and fetch the seed for either outcome of the step.
    // Nothing of that depends on the step, so it costs the association no time.
    if (threadIdx.x < kWords) reinterpret_cast<int *>(&st_sm)[threadIdx.x] = st_w;
    if (step_mode >= 0) {
        if constexpr (GRID) {
            float v = 0.0f, v2 = 0.0f;
            if (rp_g >= 0) v = rp_sum_padded<NL>(rp_a);
            if (rp_g2 >= 0) v2 = rp_sum_padded<NL>(rp_b);
            rp_finish_wave0(v, rp_g, v2, rp_g2, acc_sm, &rp_cnt);  // wave 0 leaves it with acc_sm and every wave's st_sm words visible TO IT
        } else {
            rp_finish(rp_sum_padded<NL>(rp_a), acc_sm);  // ends with a barrier: st_sm and acc_sm are visible
        }
        GS_STAMP(9);  // (diagnostic build: the rows are summed)
        // (the record takes the state BEFORE the step from the global copy: wave 0 is about to change the LDS one)
        if (blockIdx.x == 0 && rec && threadIdx.x < kWords) reinterpret_cast<int *>(rec)[REC_STATE + threadIdx.x] = st_w;
    }
    if (wave == 0) {
        const bool pub = blockIdx.x == 0;  // the one block whose copy of the new state is published
        if (step_mode >= 0)
            step_wave0(&st_sm, acc_sm, step_mode, C->gp, pub ? C->trace : nullptr, pub ? C->out_T : nullptr, look_slot, pub ? rec : nullptr, lu_sm, true);
        GS_STAMP(10);  // (wave 0: the step is done)
    } else if (!grid && tile_live && !first && wave == 1) {
        // chunk-box search: the seed (the previous neighbour's target point) for either outcome of the step, fetched
        // while wave 0 computes it -- two dependent loads less on the association's critical path
        if (ok) {
            const int ba = S_in->b_cur, bb = look_slot >= 0 ? look_slot : 1 - ba;
            const unsigned long long ka = B.N(ba)[i], kb = B.N(bb)[i];
            // (one of the two arrays may never have been written -- the outcome that cannot happen: clamp as unsigned)
            const int sa = (int)min((uint32_t)(ka & 0xffffffffu), (uint32_t)(nt - 1)), sb = (int)min((uint32_t)(kb & 0xffffffffu), (uint32_t)(nt - 1));
            const f3 qa = ld3(C->tgt, sa), qb = ld3(C->tgt, sb);
            *reinterpret_cast<float4 *>(sh.seed[0][lane]) = make_float4(qa.x, qa.y, qa.z, __int_as_float(sa));
            *reinterpret_cast<float4 *>(sh.seed[1][lane]) = make_float4(qb.x, qb.y, qb.z, __int_as_float(sb));
        }
    } else if (grid && tile_live) {
        // Three roles.  Wave 1 PLANS: every lane's window centre, the displacement of the tile's majority, the tile's row
        // bands and (one trip) their slot ranges -> LDS, then a flag.  Wave 2 prepares the LANES: the seeds for either
        // outcome of the step and each lane's own window rows (its loads leave at once; the rows are packed against the plan
        // when it is there).  The other thirteen sleep until the plan is in LDS; then all fifteen stage the bands' targets.
        // Every staging wave used to derive the same plan for itself -- ~200 instructions x 15 waves on four SIMDs: the
        // issue slots, not the memory trips, were what the phase stamps showed between "first batch arrived" and "centre
        // known" (1.6 us, r04a) -- and wave 1 carried the seeds and rows on top of the plan.
        const int Wd = C->hints.grid_w, nc = C->hints.grid_w * C->hints.grid_h;
        constexpr int R = 1;
        int bbase[WBANDS], pstart[WBANDS], bcnt[WBANDS], used = 0;  // per band: first slot, pool offset (INT_MAX: not in the pool), length; pool fill
#pragma unroll
        for (int q = 0; q < WBANDS; ++q) { bbase[q] = 0; pstart[q] = 0x7fffffff; bcnt[q] = 0; }
        // Window centre: the grid pixel the point projects to.  The point itself is only known once the step (wave 0,
        // concurrently) has produced dT -- but it is within millimetres of the cloud the PREVIOUS launch wrote, whatever
        // the step decides, and that launch left the pixel of every point it wrote in C->cells (by launch parity: one
        // load at an address known at kernel start; first launch: the caller's cloud under the initial transform,
        // exactly).  The centre only selects which window is examined; the proof below is evaluated for the point's
        // actual position against it.
        const int h = ok ? min(max(e_h, 0), nc - 1) : 0;
        int c = h;
        if (wave <= 2 && ok && C->cam_ok) c = first ? cam_cell(C->cam, xform(S_in->dT, e_pp)) : min(max(e_c, 0), nc - 1);
        int row_lo[WROWS], row_hi[WROWS];
        float4 sd[2] = {make_float4(0.0f, 0.0f, 0.0f, 0.0f), make_float4(0.0f, 0.0f, 0.0f, 0.0f)};
#pragma unroll
        for (int r = 0; r < WROWS; ++r) { row_lo[r] = 0; row_hi[r] = 0; }
        if (wave == 2) {
            // the first / one-past-last slots of the lane's three window rows, and the seeds (the step leaves b_cur as it is or
            // moves it to the look-ahead's array): requested now, used when the plan is there
#pragma unroll
            for (int r = 0; r < WROWS; ++r) {
                const int g = c + (r - 1) * Wd;
                row_lo[r] = C->hints.pix_start[min(max(g - 1, 0), nc - 1)];
                row_hi[r] = C->hints.pix_start[min(max(g + 1, 0), nc - 1) + 1];
            }
            if (ok) {
                int sj[2];
                if (first) {
                    const int slot = min(max(C->hints.pix_start[h], 0), nt - 1);
                    sj[0] = sj[1] = min(max(C->hints.scan_orig[slot], 0), nt - 1);
                } else {
                    // (one of the two arrays may never have been written -- the outcome that cannot happen: clamp as unsigned)
                    sj[0] = (int)min((uint32_t)(e_ka & 0xffffffffu), (uint32_t)(nt - 1));
                    sj[1] = (int)min((uint32_t)(e_kb & 0xffffffffu), (uint32_t)(nt - 1));
                }
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const f3 q = ld3(C->tgt, sj[u]);
                    sd[u] = make_float4(q.x, q.y, q.z, __int_as_float(sj[u]));
                }
            }
        }
        if (wave == 1) {
            // The lanes of a tile move together: their centres are their own pixels (consecutive in row-major order, also
            // across a row end) plus nearly the same displacement.  Relative to the tile's smallest displacement a lane
            // sits up to a pixel further along the row and / or one row further down (rel); what remains is contiguous
            // in row-major order again, so every row band of the tile is ONE slot range.
            // (wave reductions leave uniform values in vector registers: move them, and all that follows, to scalars)
            // The reference displacement is the MAJORITY's: a lane whose neighbour is far away (no map point near it) has a
            // centre anywhere, and taking the plain minimum would let one such lane cost the whole tile its windows.  Up to
            // three candidates (the first lanes not yet represented); supporters = lanes within a row and three columns.
            const int dsp = c - h;
            auto near = [&](int r) { return abs(r) <= 3 || abs(r - Wd) <= 3 || abs(r + Wd) <= 3; };
            const unsigned long long okm = __ballot(ok);
            unsigned long long pool = okm, sup = 0;
            for (int tries = 0; tries < 3 && pool; ++tries) {
                const int cand = __builtin_amdgcn_readlane(dsp, __builtin_ctzll(pool));
                const unsigned long long m = __ballot(ok && near(dsp - cand));
                if (__popcll(m) > __popcll(sup)) sup = m;
                if (2 * __popcll(m) >= __popcll(okm)) break;
                pool &= ~m;
            }
            GS_STAMP(13);
            const bool mine = ok && ((sup >> lane) & 1);
            const int dmin = __builtin_amdgcn_readfirstlane(wave_min_i(mine ? dsp : 0x7fffffff));
            const int e = mine ? dsp - dmin : 0;
            int rel = mine ? (e >= Wd / 2) + (e >= Wd + Wd / 2) : 2;  // rel > 1: no window (a lane that does not move with its tile)
            const bool in = mine && rel <= 1 && abs(e - rel * Wd) <= 6;
            if (!in) rel = 2;
            const int beta = c - rel * Wd;
            const int bmin = __builtin_amdgcn_readfirstlane(wave_min_i(in ? beta : 0x7fffffff));
            const int bmax = __builtin_amdgcn_readfirstlane(wave_max_i(in ? beta : (int)0x80000000));
            const bool two_rows = __any(in && rel == 1);
            // Band kk covers the pixels [bmin + (kk - R) Wd - R, bmax + (kk - R) Wd + R], kk = 0 .. 2 R (+ 1 if the lanes sit in
            // two rows), R = 1 (radius 2 = five rows, six bands was measured: ~1 us per launch more on a dense target, nothing
            // gained on a sparse one).  All first-slot loads are issued before any is used: taken one band after the other
            // they were four dependent trips through the scalar cache (2.4 us, r03h).
            int boff[WBANDS];
            {
                int lo_raw[WBANDS], hi_raw[WBANDS];
                bool want[WBANDS];
#pragma unroll
                for (int kk = 0; kk < WBANDS; ++kk) {
                    const int a = bmin + (kk - R) * Wd - R, b = bmax + (kk - R) * Wd + R;
                    want[kk] = kk <= 2 * R + (two_rows ? 1 : 0) && bmax >= bmin && b >= 0 && a <= nc - 1;
                    lo_raw[kk] = C->hints.pix_start[min(max(a, 0), nc - 1)];
                    hi_raw[kk] = C->hints.pix_start[min(max(b, 0), nc - 1) + 1];
                }
                sh.centre[lane] = c;
                sh.wflag[lane] = rel;  // (provisional: wave 2 completes it)
#pragma unroll
                for (int kk = 0; kk < WBANDS; ++kk) {
                    bbase[kk] = 0; bcnt[kk] = 0; boff[kk] = used;
                    const int lo = min(max(lo_raw[kk], 0), nt) & ~(CHUNK - 1);
                    const int hi = min((min(max(hi_raw[kk], 0), nt) + CHUNK - 1) & ~(CHUNK - 1), nt);
                    if (want[kk] && hi > lo) {
                        bbase[kk] = lo; bcnt[kk] = hi - lo;
                        // a band the pool has no room for is read from memory by the lanes themselves (slower, but the
                        // window stays complete and with it the proof): pool offset -1
                        if (used + (hi - lo) <= POOL) used += hi - lo; else boff[kk] = -1;
                    }
                }
            }
            // the plan: LDS, then the flag (release)
            if (lane < WBANDS) {
                int bb = bbase[0], bo = boff[0], bn = bcnt[0];
#pragma unroll
                for (int q = 1; q < WBANDS; ++q) { bb = lane == q ? bbase[q] : bb; bo = lane == q ? boff[q] : bo; bn = lane == q ? bcnt[q] : bn; }
                sh.band[lane] = bb; sh.band[WBANDS + lane] = bo;
                sh.plan[lane] = (bo >= 0 && bn > 0) ? bo : 0x7fffffff;
                sh.plan[WBANDS + lane] = bn;
            }
            if (lane == 0) { sh.band[2 * WBANDS] = used; sh.cnt = 0; }
            sh.key[lane] = KEY_NONE;
            if (lane == 0) __hip_atomic_store(&sh.plan_ready, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
#pragma unroll
            for (int q = 0; q < WBANDS; ++q) pstart[q] = (boff[q] >= 0 && bcnt[q] > 0) ? boff[q] : 0x7fffffff;
        } else {
            while (__hip_atomic_load(&sh.plan_ready, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) == 0u) __builtin_amdgcn_s_sleep(2);
            const int pl = sh.band[min(lane, 2 * WBANDS)], ps = sh.plan[min(lane, 2 * WBANDS - 1)];
#pragma unroll
            for (int q = 0; q < WBANDS; ++q) {
                bbase[q] = __builtin_amdgcn_readlane(pl, q);
                pstart[q] = __builtin_amdgcn_readlane(ps, q);
                bcnt[q] = __builtin_amdgcn_readlane(ps, WBANDS + q);
            }
            used = __builtin_amdgcn_readlane(pl, 2 * WBANDS);
        }
        GS_STAMP(14);
        // staging loads first (they are the long ones), the per-lane rows behind them
        constexpr int ST = KNN_BT - 64, NR = (POOL + ST - 1) / ST;
        float4 sreg[NR];
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const int e = (int)threadIdx.x - 64 + ST * r;
            if (e < used) {
                int bb = 0, bo = 0;  // the staged band that holds pool element e: the last one that starts at or before it
#pragma unroll
                for (int q = 0; q < WBANDS; ++q) {
                    const bool here = e >= pstart[q];
                    bb = here ? bbase[q] : bb; bo = here ? pstart[q] : bo;
                }
                const int slot = bb + e - bo;
                const f3 q3 = ld3(C->hints.scan_points, slot);
                sreg[r] = make_float4(q3.x, q3.y, q3.z, __int_as_float(C->hints.scan_orig[slot]));
            }
        }
        if (wave == 2) {
            const int rel = sh.wflag[lane];  // (the planner's; its centre is this wave's own c: same arithmetic on the same words)
            const bool in = rel <= 1;
            bool full = in;
#pragma unroll
            for (int r = 0; r < WROWS; ++r) {
                int packed = 0;
                const int g = c + (r - R) * Wd;
                if (in && r <= 2 * R && g + R >= 0 && g - R <= nc - 1) {
                    const int lo = min(max(row_lo[r], 0), nt) & ~(CHUNK - 1);
                    const int hi = min((min(max(row_hi[r], 0), nt) + CHUNK - 1) & ~(CHUNK - 1), nt);
                    if (hi > lo) {
                        int bb = bbase[0], bn = bcnt[0];
#pragma unroll
                        for (int q = 1; q < WBANDS; ++q) { bb = (rel + r) == q ? bbase[q] : bb; bn = (rel + r) == q ? bcnt[q] : bn; }
                        // inside the staged band, or not examined (then the lane has no certificate: `full`)
                        // (LaneWin packs the chunk count in 9 bits: a staged band is at most POOL / CHUNK = 256 chunks, but a
                        // band read from memory has no such bound -- a longer row stays unpacked and the lane uncertified)
                        if (lo >= bb && hi <= bb + bn && (hi - lo + CHUNK - 1) / CHUNK <= 511) packed = LaneWin::pack(lo, hi - lo); else full = false;
                    }
                }
                sh.win[r][lane] = packed;
            }
            sh.wflag[lane] = min(rel, 2) | (full ? 4 : 0) | (R << 3);
            if (lane < (int)(sizeof(CamK) / 4)) reinterpret_cast<int *>(&sh.cam)[lane] = e_cam;
            *reinterpret_cast<float4 *>(sh.seed[0][lane]) = sd[0];
            *reinterpret_cast<float4 *>(sh.seed[1][lane]) = sd[1];
        }
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const int e = (int)threadIdx.x - 64 + ST * r;
            if (e < used) *reinterpret_cast<float4 *>(&sh.u.stage[4 * e]) = sreg[r];
        }
        GS_STAMP(15);
    }
    __syncthreads();
    if (step_mode >= 0 && blockIdx.x == 0 && threadIdx.x < kWords) {
        const int v = reinterpret_cast<const int *>(&st_sm)[threadIdx.x];
        reinterpret_cast<int *>(S_out)[threadIdx.x] = v;
        if (rec) reinterpret_cast<int *>(rec)[REC_WORDS + REC_STATE + threadIdx.x] = v;
    }
    GS_STAMP(7);
    const IcpState *S = &st_sm;
    if (tile0 >= ns) {  // empty tile: its partial row must still be defined
        if (threadIdx.x < NACC) partials[blockIdx.x * NACC + threadIdx.x] = 0.0f;
        return;
    }
    const int p_cur = S->p_cur, b_cur = S->b_cur;
    const float *in = first ? user_src : B.P(p_cur);
    float *out = B.P(out_slot >= 0 ? out_slot : 1 - p_cur);
    unsigned long long *best = B.N(out_slot >= 0 ? out_slot : 1 - b_cur);
    f3 s{0.0f, 0.0f, 0.0f};
    if (ok) {
        s = xform(S->dT, ld3(in, i));
        if (wave == 0) st3(out, i, s);
    }
    if (nt <= 0) {
        if (ok && wave == 0) best[i] = KEY_NONE;
        if (threadIdx.x < NACC) partials[blockIdx.x * NACC + threadIdx.x] = 0.0f;
        return;
    }
    unsigned long long key;
    bool need = false;
    int rel = 2;  // grid search: the lane's centre row relative to the tile's first (0 / 1), | 4 = window fully staged
    if (grid) {
        GS_STAMP(0);
        rel = sh.wflag[lane];
        // the seed: one real candidate per lane, fetched for either outcome of the step
        if (wave == 0) {
            unsigned long long k0 = KEY_NONE;
            if (ok) {
                const int u = (first || b_cur == S_in->b_cur) ? 0 : 1;
                const float4 q = *reinterpret_cast<const float4 *>(sh.seed[u][lane]);
                k0 = pack_key(dist2(s, q.x, q.y, q.z), __float_as_int(q.w));
            }
            atomicMin(&sh.key[lane], k0);
        }
        {   // window: every wave takes every 16th slot of the lane's row ranges
            float bd = INFINITY;
            int bi = 0x7fffffff;
#pragma unroll
            for (int r = 0; r < WROWS; ++r) {
                const int wr = sh.win[r][lane];
                const int kk = (rel & 3) + r, wn = LaneWin::len(wr, nt);
                const int po = sh.band[WBANDS + min(kk, WBANDS - 1)];
                const float *row = sh.u.stage + 4 * (max(po, 0) + LaneWin::lo(wr) - sh.band[min(kk, WBANDS - 1)]);
                for (int p = wave; p < wn; p += KNN_NW) {
                    float4 q;
                    if (po >= 0) {
                        q = *reinterpret_cast<const float4 *>(row + 4 * p);
                    } else {  // band not staged: straight from memory
                        const int slot = LaneWin::lo(wr) + p;
                        const f3 g3 = ld3(C->hints.scan_points, slot);
                        q = make_float4(g3.x, g3.y, g3.z, __int_as_float(C->hints.scan_orig[slot]));
                    }
                    const float d = dist2(s, q.x, q.y, q.z);
                    const int jj = __float_as_int(q.w);
                    const bool better = (d < bd) | ((d == bd) & (jj < bi));
                    bd = better ? d : bd;
                    bi = better ? jj : bi;
                }
            }
            if (ok && bd < INFINITY) atomicMin(&sh.key[lane], pack_key(bd, bi));
        }
        __syncthreads();
        GS_STAMP(1);
        float bd;
        int bi;
        key_unpack(sh.key[lane], bd, bi);
        // proof: every target outside the window is at least sqrt(cam_bound2) away (see there); the window was examined
        // completely (rel & 4), so a best strictly inside the bound IS the nearest neighbour, tie-break included
        const float bound2 = (ok && C->cam_ok) ? cam_bound2(sh.cam, s, sh.centre[lane], rel >> 3) : 0.0f;
        const bool proven = !C->cert_off & ((rel & 4) != 0) & (bd * 1.0001f < bound2);
        need = ok & !proven;
        const unsigned long long need_mask = __ballot(need);  // the same 64 lanes in every wave: block-uniform
        GS_COUNT(12, (unsigned long long)__popcll(need_mask));
        bool tile_search = __popcll(need_mask) > 6;
        if (!tile_search && need_mask) {
            tile_search = !knn_point_search(sh, s, need_mask, C->hints.scan_points, C->hints.scan_orig, C->boxes, C->sboxes, nt);  // ends with a barrier
        }
        if (tile_search) {
            tile_box(sh, s, need);
            __syncthreads();
            knn_prune_search<true>(sh, s, ok, need, C->hints.scan_points, C->hints.scan_orig, C->boxes, C->sboxes, nt);  // ends with a barrier
        }
        GS_STAMP(2);
        key = ok ? sh.key[lane] : KEY_NONE;
        GS_STAMP(3);
        // diagnostic build: which CU the block ran on (slot 5 as knn_prune_search writes it; a proven tile never gets there)
        GS_COUNT(5, ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) << 16) |
                        ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 48));
    } else {
        int sj = -1;
        if (!first) {  // seeded from what wave 1 fetched during the step (knn_tile's -2: keys already in LDS)
            sj = -2;
            if (wave == 0) {
                unsigned long long k0 = KEY_NONE;
                if (ok) {
                    const float4 q = *reinterpret_cast<const float4 *>(sh.seed[b_cur == S_in->b_cur ? 0 : 1][lane]);
                    k0 = pack_key(dist2(s, q.x, q.y, q.z), __float_as_int(q.w));
                }
                sh.key[lane] = k0;
            }
        }
        const bool window_seed = first && C->hints.scan_points && C->hints.src_pix && C->hints.pix_start && C->hints.grid_w > 0;
        if (window_seed) sj = -2;  // seeded by knn_window_seed below (block-uniform decision)
        const float *scan = C->hints.scan_points ? C->hints.scan_points : C->tgt;
        const int32_t *scan_orig = C->hints.scan_points ? C->hints.scan_orig : nullptr;
        if (window_seed) knn_window_seed(sh, s, ok, i, C->hints, nt);
        key = knn_tile(sh, s, ok, sj, C->tgt, scan, scan_orig, C->boxes, C->sboxes, nt);
    }
    // linearise this tile straight away (J fused into K's epilogue): 29 sums over the tile's 64 points,
    // reduced through LDS by the whole block in a fixed order (two short stages instead of 29 butterflies)
    if (wave == 0) {
        if (ok) best[i] = key;
        if (grid && ok && C->cam_ok) C->cells[par * C->cells_stride + i] = cam_cell(sh.cam, s);  // where the next launch looks
        float acc[NACC];
#pragma unroll
        for (int k = 0; k < NACC; ++k) acc[k] = 0.0f;
        const Row r = make_row_from(s, ok, key, C->tgt, C->nrm, C->thresh);
        if (r.valid) accumulate_row(r, acc);
#pragma unroll
        for (int k = 0; k < NACC; ++k) sh.u.a.rows[k][lane] = acc[k];
    }
    __syncthreads();
    if (threadIdx.x < NACC * 16) {
        const int k = threadIdx.x >> 4, p4 = (threadIdx.x & 15) * 4;
        sh.u.a.part[k][threadIdx.x & 15] = ((sh.u.a.rows[k][p4] + sh.u.a.rows[k][p4 + 1]) + sh.u.a.rows[k][p4 + 2]) + sh.u.a.rows[k][p4 + 3];
    }
    __syncthreads();
    if (threadIdx.x < NACC) {
        float v = 0.0f;
#pragma unroll
        for (int q = 0; q < 16; ++q) v += sh.u.a.part[threadIdx.x][q];
        partials[blockIdx.x * NACC + threadIdx.x] = v;
    }
}

__global__ void icp_init_state_k(IcpState *S, const float *__restrict__ init_T /* NULL = identity */, float damp) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        for (int i = 0; i < 16; ++i) {
            const float v = init_T ? init_T[i] : ((i % 5 == 0) ? 1.0f : 0.0f);
            S->T[i] = v; S->dT[i] = v;
        }
        for (int i = 0; i < 44; ++i) S->cur[i] = 0.0f;
        S->damp = damp;
        S->p_cur = 1;  // the first association writes pts[0] / best[0]
        S->b_cur = 1; S->b_first = 0;
        S->it = 0;
    }
}

// one launch for the loop's preparations: initial state (one lane), the target's chunk boxes and, per SUPER = 64
// chunks (one block), their common box -- the second level the point-serial search consults first
__global__ __launch_bounds__(SUPER * CHUNK) void icp_prepare_k(IcpState *S, const float *__restrict__ init_T, float damp,
                                                              const float *__restrict__ tgt, const int32_t *__restrict__ d_nt,
                                                              float *__restrict__ boxes, float *__restrict__ sboxes,
                                                              LoopConst lc, LoopConst *__restrict__ lc_out, float *__restrict__ part0,
                                                              float *__restrict__ part1, int rows_written, int rows_read) {
    static_assert(SUPER * CHUNK == 1024, "one block per super-box");
    // rows the loop's launches read at kernel start but never write (knn1_loop_k's unmasked row sums): zeros
    for (int q = rows_written * NACC + blockIdx.x * blockDim.x + threadIdx.x; q < rows_read * NACC; q += gridDim.x * blockDim.x) {
        part0[q] = 0.0f; part1[q] = 0.0f;
    }
    __shared__ float wb[16][6];
    if (blockIdx.x == 0 && threadIdx.x < sizeof(LoopConst) / 4) {  // the loop's constants, for its association launches
        int v = reinterpret_cast<const int *>(&lc)[threadIdx.x];
        if (threadIdx.x == offsetof(LoopConst, ns) / 4) v = *lc.d_ns;
        if (threadIdx.x == offsetof(LoopConst, nt) / 4) v = *lc.d_nt;
        if (threadIdx.x == offsetof(LoopConst, tile_points) / 4) {  // (any one thread: the loop counters of gs_loop_counts)
            atomicAdd(&g_loop_counts[0], 1u);
            if (lc.grid_variant) atomicAdd(&g_loop_counts[1], 1u);
            if (lc.tile_points != 64) atomicAdd(&g_loop_counts[2], 1u);
        }
        reinterpret_cast<int *>(lc_out)[threadIdx.x] = v;
    }
    // the loop's own copy of hints.src_pix, defined up to the cloud's capacity: the association kernel requests it at kernel
    // start by an index clamped to the capacity, before it knows ns (knn1_loop_k)
    if (lc.cells) {
        const int n = *lc.d_ns;
        int32_t *pix = lc.cells + 2 * (size_t)lc.cells_stride;
        for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < lc.cells_stride; i += gridDim.x * blockDim.x)
            pix[i] = (i < n && lc.hints.src_pix) ? lc.hints.src_pix[i] : 0;
    }
    if (blockIdx.x == 0) {  // the bucketing camera, after the plain copy above (same words)
        __syncthreads();
        if (threadIdx.x == 0 && lc.hints.cam_pose && lc.hints.cam_K && lc.hints.ds > 0) {
            const float *T = lc.hints.cam_pose, *K = lc.hints.cam_K;
            const Cam c = make_cam(T, K);
            for (int q = 0; q < 9; ++q) lc_out->cam.R[q] = c.R[q];
            for (int q = 0; q < 3; ++q) lc_out->cam.T[q] = c.tinv[q];
            lc_out->cam.fx = K[0]; lc_out->cam.fy = K[5]; lc_out->cam.cx = K[2]; lc_out->cam.cy = K[6];
            lc_out->cam.ds = lc.hints.ds; lc_out->cam.Wd = lc.hints.grid_w; lc_out->cam.Hd = lc.hints.grid_h;
            // project_point divides (K row 0 / 1) . [x y z 1] by (K row 2) . [x y z 1]: the proof's planes assume the
            // plain pinhole form u = fx x / z + cx, v = fy y / z + cy
            const bool pinhole = K[1] == 0.0f && K[3] == 0.0f && K[4] == 0.0f && K[7] == 0.0f && K[8] == 0.0f && K[9] == 0.0f &&
                                 K[10] == 1.0f && K[11] == 0.0f && K[0] != 0.0f && K[5] != 0.0f;
            lc_out->cam_ok = pinhole ? 1 : 0;
        }
    }
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        for (int i = 0; i < 16; ++i) {
            const float v = init_T ? init_T[i] : ((i % 5 == 0) ? 1.0f : 0.0f);
            S->T[i] = v; S->dT[i] = v;
        }
        for (int i = 0; i < 44; ++i) S->cur[i] = 0.0f;
        for (int i = 0; i < 6; ++i) S->xi[i] = 0.0f;
        S->damp = damp;
        S->p_cur = 1;  // the first association writes pts[0] / best[0]
        S->b_cur = 1; S->b_first = 0;
        S->it = 0;
    }
    const int nt = *d_nt;
    const int j = blockIdx.x * (SUPER * CHUNK) + threadIdx.x;
    if (blockIdx.x * (SUPER * CHUNK) >= nt) return;
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    if (j < nt) {
        const f3 p = ld3(tgt, j);
        lo[0] = hi[0] = p.x; lo[1] = hi[1] = p.y; lo[2] = hi[2] = p.z;
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
#pragma unroll
        for (int off = CHUNK / 2; off > 0; off >>= 1) {
            lo[a] = fminf(lo[a], __shfl_xor(lo[a], off, kWave));
            hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], off, kWave));
        }
    }
    if ((threadIdx.x % CHUNK) == 0 && j < nt) {
        float *b = boxes + 6 * (int64_t)(j / CHUNK);
        b[0] = lo[0]; b[1] = lo[1]; b[2] = lo[2]; b[3] = hi[0]; b[4] = hi[1]; b[5] = hi[2];
    }
    // the block's box: finish the wave reduction, then the sixteen waves through LDS
#pragma unroll
    for (int a = 0; a < 3; ++a) {
#pragma unroll
        for (int off = 32; off >= CHUNK; off >>= 1) {
            lo[a] = fminf(lo[a], __shfl_xor(lo[a], off, kWave));
            hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], off, kWave));
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { wb[wave][0] = lo[0]; wb[wave][1] = lo[1]; wb[wave][2] = lo[2]; wb[wave][3] = hi[0]; wb[wave][4] = hi[1]; wb[wave][5] = hi[2]; }
    __syncthreads();
    if (threadIdx.x < 6) {
        float v = wb[0][threadIdx.x];
        for (int w = 1; w < 16; ++w) v = threadIdx.x < 3 ? fminf(v, wb[w][threadIdx.x]) : fmaxf(v, wb[w][threadIdx.x]);
        sboxes[6 * (int64_t)blockIdx.x + threadIdx.x] = v;
    }
}

__global__ void copy_best_last_k(const IcpState *__restrict__ S, LoopBufs B, const int32_t *__restrict__ d_ns,
                                 unsigned long long *__restrict__ out) {
    const unsigned long long *src = B.N(S->b_first);
    const int ns = *d_ns;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < ns; i += gridDim.x * blockDim.x) out[i] = src[i];
}

// ------------------------------------------------------------------ launch geometry
static inline int knn_nsplit_brute(int max_ns, int max_nt) {
    const int tiles = cdiv(max_ns, KNN_T);
    int ns = cdiv(2048, tiles);
    const int cap = cdiv(max_nt, 64);
    if (ns > cap) ns = cap;
    if (ns < 1) ns = 1;
    if (ns > 4096) ns = 4096;
    return ns;
}
static inline int lin_blocks(int max_ns) {
    int nb = cdiv(max_ns, LIN_T);
    if (nb > LIN_MAXB) nb = LIN_MAXB;
    if (nb < 1) nb = 1;
    return nb;
}
static inline size_t boxes_bytes(int max_nt) { return align_up((size_t)cdiv(max_nt > 0 ? max_nt : 1, CHUNK) * 6 * 4, 256); }

// ------------------------------------------------------------------ optional per-kernel timing
// bench.py asks for the average duration of the two hot kernels of the loop, measured with HIP events
// on the stream they are launched on.  Off by default (no events, no overhead).
struct Prof {
    bool on = false;
    std::vector<hipEvent_t> ev[2][2];  // [tag][start|stop]
    size_t used[2] = {0, 0};
    double total_ms[2] = {0.0, 0.0};
    long count[2] = {0, 0};
};
static Prof g_prof;
static inline void prof_mark(int tag, int which, hipStream_t st) {
    if (!g_prof.on) return;
    auto &v = g_prof.ev[tag][which];
    const size_t i = g_prof.used[tag];
    if (i >= v.size()) {
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) return;
        v.push_back(e);
    }
    (void)hipEventRecord(v[i], st);
    if (which == 1) g_prof.used[tag] = i + 1;
}

bool profiling_enabled() { return g_prof.on; }

static int g_grid_mode = getenv("GS_GRID_MODE") ? atoi(getenv("GS_GRID_MODE")) : 1;  // gs_set_grid_search (environment: measurements)
static int g_tile_points = 0;  // gs_set_tile_points (0 = automatic)

// Source points per block of the loops' association kernel (knn1_loop_k): 64 -- lane = point.  Round 2 cut a dense
// target's cloud into 38-point tiles (every CU two equal tiles: the chunk-box search there was VALU-issue bound and a
// launch lasted as long as its doubled-up CUs); with the grid search's geometric proof the launch is latency-bound at every
// density and the tile size no longer matters (200 frames, MI355X: 1 864 / 1 842 / 1 838 frames/s for 64 / the round-2 rule /
// 38, profiles/r03d), so the rule is gone and with it the surplus blocks it launched while a map grew dense.  The
// setting remains for tests (gs_set_tile_points): the tile size fixes the order of the 29-term sums, so results of
// different settings agree to rounding, nearest neighbours exactly.
constexpr int TILE_MIN = 32;
static inline int loop_tile_points() {
    static const int env = getenv("GS_TILE_POINTS") ? atoi(getenv("GS_TILE_POINTS")) : 0;
    const int forced = g_tile_points ? g_tile_points : env;
    return (forced >= TILE_MIN && forced <= 64) ? forced : 64;
}
int icp_config_stamp() { return g_grid_mode | (g_tile_points << 4); }  // part of slam.hip's graph-cache key
static inline int loop_blocks_max(int max_ns) { return cdiv(max_ns, TILE_MIN); }  // workspace: whatever the tile size
// rows of a partial-sum buffer: the association kernel's prologue reads rows 0 .. 32 RP_LOADS - 1 without testing them against
// the launch's row count (icp_prepare_k zeroes the rows the loop's launches do not write), four bytes at a time up to twelve
// bytes past a row's end: one spare row
static inline int partial_rows_alloc(int max_ns) { return std::max(loop_blocks_max(max_ns), 32 * RP_LOADS) + 1; }

struct IcpWs {
    IcpState *S[2];      // double-buffered across launches (see knn1_loop_k)
    LoopBufs B;
    float *partials[2];
    float *boxes, *sboxes;  // chunk boxes, and one box per SUPER chunks
    LoopConst *lc;       // the loop's constants (knn1_loop_k reads them from here, not from its arguments)
    int32_t *cells;      // grid search: (2, max_ns) ds-grid pixel of every point an association wrote, by launch parity
};
static inline size_t icp_ws_layout(int max_ns, int max_nt, void *ws, IcpWs *out) {
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += align_up(bytes, 256); return o; };
    const size_t oS = take(sizeof(IcpState)), oS1 = take(sizeof(IcpState));
    const size_t oP0 = take((size_t)max_ns * 12), oP1 = take((size_t)max_ns * 12);
    const size_t oB0 = take((size_t)max_ns * 8), oB1 = take((size_t)max_ns * 8);
    const size_t oPart = take((size_t)partial_rows_alloc(max_ns) * NACC * 4), oPart1 = take((size_t)partial_rows_alloc(max_ns) * NACC * 4);
    const size_t oBox = take(boxes_bytes(max_nt)), oSBox = take((size_t)cdiv(max_nt > 0 ? max_nt : 1, 1024) * 6 * 4);
    const size_t oLc = take(sizeof(LoopConst)), oCells = take((size_t)max_ns * 12);  // cells by launch parity (2 planes) | the loop's copy of hints.src_pix
    if (ws && out) {
        char *p = (char *)ws;
        out->S[0] = (IcpState *)(p + oS); out->S[1] = (IcpState *)(p + oS1);
        out->B.pts = (float *)(p + oP0);
        out->B.pts_stride = (int64_t)(oP1 - oP0) / 4;
        out->B.best = (unsigned long long *)(p + oB0);
        out->B.best_stride = (int64_t)(oB1 - oB0) / 8;
        out->partials[0] = (float *)(p + oPart); out->partials[1] = (float *)(p + oPart1);
        out->boxes = (float *)(p + oBox);
        out->sboxes = (float *)(p + oSBox);
        out->lc = (LoopConst *)(p + oLc);
        out->cells = (int32_t *)(p + oCells);
    }
    return off;
}

// ------------------------------------------------------------------ tape (autograd)
// [records: (steps + 1) x REC_WORDS floats][cloud slots][nearest-neighbour slots]; one slot per association
// launch, one record per step launch.  LM: numiters + 1 of each; gradLM: 2 numiters of each.
struct Tape {
    float *rec;
    LoopBufs B;
    int nslots;
};
static inline int tape_launches(bool grad, int numiters) { return grad ? 2 * numiters : numiters + 1; }
static inline size_t tape_layout(bool grad, int max_ns, int numiters, void *tape, Tape *out) {
    const int n = tape_launches(grad, numiters > 0 ? numiters : 1);
    const size_t rec_b = align_up((size_t)(n + 1) * REC_WORDS * 4, 256);
    const size_t pts_b = align_up((size_t)max_ns * 12, 256), best_b = align_up((size_t)max_ns * 8, 256);
    if (tape && out) {
        char *p = (char *)tape;
        out->rec = (float *)p;
        out->B.pts = (float *)(p + rec_b);
        out->B.pts_stride = (int64_t)pts_b / 4;
        out->B.best = (unsigned long long *)(p + rec_b + (size_t)n * pts_b);
        out->B.best_stride = (int64_t)best_b / 8;
        out->nslots = n;
    }
    return rec_b + (size_t)n * (pts_b + best_b);
}

static int icp_run(bool grad, const float *src, const int32_t *d_ns, int max_ns, const float *tgt, const float *nrm,
                   const int32_t *d_nt, int max_nt, const float *init_T, int numiters, float damp, float thresh,
                   GradParams gp, const gs_icp_hints *hints_in, float *out_T, uint64_t *best_last, float *trace, void *ws,
                   size_t ws_bytes, hipStream_t st, const char *name, void *tape = nullptr, size_t tape_bytes = 0,
                   const float *compose_right = nullptr, float *compose_out = nullptr) {
    gs_icp_hints hints{nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0, nullptr, nullptr, 0};
    if (hints_in) hints = *hints_in;
    GS_REQUIRE(!hints.scan_points || hints.scan_orig, "%s: hints.scan_points needs hints.scan_orig", name);
    GS_REQUIRE(src && d_ns && tgt && nrm && d_nt && out_T, "%s: NULL argument", name);  // init_T NULL = identity
    GS_REQUIRE(max_ns > 0 && max_nt > 0 && numiters >= 0, "%s: bad sizes max_ns=%d max_nt=%d numiters=%d", name, max_ns, max_nt, numiters);
    if (!ws || ws_bytes < icp_ws_layout(max_ns, max_nt, nullptr, nullptr)) {
        set_error("%s: workspace too small (%zu < %zu)", name, ws_bytes, icp_ws_layout(max_ns, max_nt, nullptr, nullptr));
        return GS_ERR_WORKSPACE_TOO_SMALL;
    }
    if (numiters == 0) {
        if (init_T) {
            GS_HIP(hipMemcpyAsync(out_T, init_T, 64, hipMemcpyDeviceToDevice, st), name);
        } else {
            hipLaunchKernelGGL(icp_init_state_k, dim3(1), dim3(64), 0, st, (IcpState *)ws, init_T, damp);
            GS_HIP(hipMemcpyAsync(out_T, ws, 64, hipMemcpyDeviceToDevice, st), name);  // IcpState starts with T
        }
        return GS_OK;
    }
    IcpWs w;
    icp_ws_layout(max_ns, max_nt, ws, &w);
    Tape tp{nullptr, {}, 0};
    if (tape) {
        if (tape_bytes < tape_layout(grad, max_ns, numiters, nullptr, nullptr)) {
            set_error("%s: tape too small (%zu < %zu)", name, tape_bytes, tape_layout(grad, max_ns, numiters, nullptr, nullptr));
            return GS_ERR_WORKSPACE_TOO_SMALL;
        }
        tape_layout(grad, max_ns, numiters, tape, &tp);
        w.B = tp.B;  // the tape is the loop's working storage
    }
    int n_assoc = 0, n_step = 0;
    const int tile_points = loop_tile_points();
    const dim3 kgrid(cdiv(max_ns, tile_points));
    const int lb = (int)kgrid.x;  // one partial row per tile, written by the association kernel
    const int fb = min(cdiv(max_ns, 256), 256);

    static const int cert_off = getenv("GS_CERT_OFF") != nullptr;
    // all hints and the camera given: grid search with its geometric proof (knn1_loop_k<true>), at every density;
    // GS_NO_GRID_SEARCH=1 / gs_set_grid_search(0) keep the chunk-box search for every association (same results; for
    // A/B measurements and tests)
    static const bool grid_off = getenv("GS_NO_GRID_SEARCH") != nullptr;
    const bool grid_search = !grid_off && g_grid_mode != 0 && hints.scan_points && hints.scan_orig && hints.src_pix && hints.pix_start &&
                             hints.cam_pose && hints.cam_K && hints.ds > 0 && hints.grid_w > 0 && hints.grid_h > 0;
    LoopConst lc{src, tgt, nrm, w.boxes, w.sboxes, d_ns, d_nt, trace, out_T, hints, gp, thresh, 0, 0, cert_off, tile_points,
                 grid_search ? 1 : 0, CamK{}, 0, w.cells, max_ns};
    hipLaunchKernelGGL(icp_prepare_k, dim3(cdiv(max_nt, SUPER * CHUNK)), dim3(SUPER * CHUNK), 0, st, w.S[0], init_T, damp,
                       hints.scan_points ? hints.scan_points : tgt, d_nt, w.boxes, w.sboxes, lc, w.lc, w.partials[0], w.partials[1], lb,
                       std::max(lb, 32 * RP_LOADS));
    GS_LAUNCH_CHECK(name);
    // The loop as a sequence  A S A S ... A S  (A = association + linearise launch, S = O(1) step on A's sums).
    // Every S but the last runs in the prologue of the A that follows it; state and partial sums alternate
    // between two buffers from launch to launch.
    int cur = 0;            // buffer the next launch READS its state / the previous sums from
    int pending = -1;       // step waiting to be folded into the next association
    int pending_slot = -1;  // slot the association before that step wrote (tape mode)
    // Folding pays while the association's blocks all run at once (latency-bound regime: the step's ~4 us ride in
    // every block instead of a ~4.5 us launch).  With more tiles than the chip holds (2 x 1024-thread blocks per CU)
    // the kernel is throughput-bound and 4 us of redundant work in EVERY block costs more than one small launch:
    // then each step is a launch of its own again (measured at 78 k source points: 1225 blocks).
    const bool fold = (int)kgrid.x <= 2 * 256;
    auto assoc = [&](int first) {
        if (!fold && pending >= 0) {  // stand-alone step, state updated in place
            hipLaunchKernelGGL(icp_step_k, dim3(1), dim3(1024), 0, st, w.S[cur], w.partials[cur], lb, pending, gp, trace, out_T,
                               pending_slot, tape ? tp.rec + (size_t)n_step * REC_WORDS : nullptr, 1, (const float *)nullptr,
                               (float *)nullptr);
            ++n_step;
            pending = -1;
        }
        const int nxt = pending >= 0 ? 1 - cur : cur;  // a folded step publishes the new state to the other buffer
        prof_mark(0, 0, st);
        float *rec_p = (tape && pending >= 0) ? tp.rec + (size_t)n_step * REC_WORDS : nullptr;
        auto launch = [&](auto kernel) {
            hipLaunchKernelGGL(kernel, kgrid, dim3(KNN_BT), 0, st, (const LoopConst *)w.lc, (const IcpState *)w.S[cur], (const float *)w.partials[cur],
                               (const int32_t *)(w.cells + 2 * (size_t)max_ns), (const int32_t *)(w.cells + (size_t)(1 - (n_assoc & 1)) * max_ns),
                               max_ns, tile_points, first | ((n_assoc & 1) << 1), pending, pending_slot, lb, w.S[nxt], rec_p,
                               tape ? n_assoc : -1, w.B, w.partials[nxt], src);
        };
        if (grid_search && lb <= 32 * RP_FEW) launch(knn1_loop_k<true, RP_FEW>);
        else if (grid_search) launch(knn1_loop_k<true, RP_LOADS>);
        else launch(knn1_loop_k<false, RP_LOADS>);
        prof_mark(0, 1, st);
        if (pending >= 0) ++n_step;
        cur = nxt;
        pending = -1;
        ++n_assoc;
    };
    auto step = [&](int mode) {  // deferred: folded into the next association, or launched by finish()
        pending = mode;
        pending_slot = tape ? n_assoc - 1 : -1;
    };
    auto finish = [&]() {  // the loop's last step has no association behind it
        hipLaunchKernelGGL(icp_step_k, dim3(1), dim3(1024), 0, st, w.S[cur], w.partials[cur], lb, pending, gp, trace, out_T,
                           pending_slot, tape ? tp.rec + (size_t)n_step * REC_WORDS : nullptr, 0, compose_right, compose_out);
        ++n_step;
        pending = -1;
    };
    // NB with a folded step the sums it reduces are those of the launch before: partials[cur] at that time.
    // assoc() above passes w.partials[cur] (read) and w.partials[nxt] (write); without a pending step nothing is
    // read and cur == nxt is harmless.
    assoc(1);
    step(STEP_ADOPT);
    if (!grad) {
        // numiters + 1 associations instead of the reference's 2 x numiters: an accepted look-ahead
        // IS the next iteration's first linearisation, a rejected one leaves it unchanged.
        for (int it = 0; it < numiters; ++it) {
            assoc(0);
            step(STEP_LM);
        }
    } else {
        for (int it = 0; it < numiters; ++it) {
            assoc(0);            // look-ahead: pts[p_cur] . exp(xi)
            step(STEP_GRAD_B);   // dT = exp(sigma xi); look-ahead discarded
            if (it + 1 < numiters) {
                assoc(0);        // new current cloud: pts[p_cur] . exp(sigma xi)
                step(STEP_ADOPT);
            }
        }
    }
    finish();
    GS_LAUNCH_CHECK(name);
    if (best_last) {
        hipLaunchKernelGGL(copy_best_last_k, dim3(fb), dim3(256), 0, st, w.S[cur], w.B, d_ns, (unsigned long long *)best_last);
        GS_LAUNCH_CHECK(name);
    }
    return GS_OK;
}


// ------------------------------------------------------------------ reverse pass of the taped loops
// Walks the tape backwards entirely on the device (accept/reject is read from the records, so rejected LM
// iterations cost two empty launches and no host round trip).  Per iteration:
//   S  small_k  : adjoints of T' = dT T, dT = exp(xi), xi = (H + damp I)^-1 g, and of the gradLM gates
//   B  look_k   : gradLM only -- adjoint of the look-ahead error
//   C  lin_k    : gP_i <- R^T gP_i + adjoint of the linearisation (H, g, e) at s_i ; and, for the step that
//                 produced s from its predecessor cloud q (s = dT q):  sum_i gP_i (x) q_i , sum_i gP_i
//                 -- what the S kernel of that earlier step needs as the adjoint of dT
// gP (ns,3) is updated in place, target / normal adjoints accumulate with float atomics.
constexpr int BWD_T = 256;
constexpr int BWD_MAXB = 512;

struct BwdState {
    float gT[16];    // adjoint of the accumulated transform
    float G[44];     // Hbar(36) | gbar(6) | ebar | pad : what lin_k applies
    float gdT[12];   // adjoint of the top 3 rows of the step being unwound (row-major 3x4)
    float R2[9];     // rotation by which lin_k pulls gP back (the step that produced the cloud gP belongs to)
    float R1[9];     // gradLM: rotation of the look-ahead step
    float gxi[6];
    float g_new_err, g_err, gdamp;
    int active;      // 0: rejected LM iteration, nothing to do
    int src_slot, nn_slot, look_slot;
    int prev_slot;   // slot of the cloud src_slot was derived from (-1: the caller's source cloud)
};

__device__ __forceinline__ const IcpState *rec_state(const float *rec) { return reinterpret_cast<const IcpState *>(rec + REC_STATE); }

// adjoint of one linearised point: returns s_bar, scatters d_bar / n_bar
__device__ __forceinline__ f3 lin_point_bwd(const float *G, const Row &r, const f3 s, const uint32_t j, const float *tgt,
                                            const float *nrm, float *g_tgt, float *g_nrm) {
    const f3 d = ld3(tgt, j), n = ld3(nrm, j);
    float ab[6];
#pragma unroll
    for (int u = 0; u < 6; ++u) {
        float v = G[36 + u] * r.b;
#pragma unroll
        for (int w = 0; w < 6; ++w) v += (G[6 * u + w] + G[6 * w + u]) * r.a[w];
        ab[u] = v;
    }
    float bb = 2.0f * G[42] * r.b;
#pragma unroll
    for (int u = 0; u < 6; ++u) bb += G[36 + u] * r.a[u];
    const f3 an{ab[0], ab[1], ab[2]}, ac{ab[3], ab[4], ab[5]};
    f3 sb{n.y * ac.z - n.z * ac.y, n.z * ac.x - n.x * ac.z, n.x * ac.y - n.y * ac.x};
    f3 nb{an.x + (ac.y * s.z - ac.z * s.y), an.y + (ac.z * s.x - ac.x * s.z), an.z + (ac.x * s.y - ac.y * s.x)};
    sb.x -= bb * n.x; sb.y -= bb * n.y; sb.z -= bb * n.z;
    nb.x += bb * (d.x - s.x); nb.y += bb * (d.y - s.y); nb.z += bb * (d.z - s.z);
    if (g_tgt) {
        atomicAdd(g_tgt + 3 * (int64_t)j, bb * n.x);
        atomicAdd(g_tgt + 3 * (int64_t)j + 1, bb * n.y);
        atomicAdd(g_tgt + 3 * (int64_t)j + 2, bb * n.z);
    }
    if (g_nrm) {
        atomicAdd(g_nrm + 3 * (int64_t)j, nb.x);
        atomicAdd(g_nrm + 3 * (int64_t)j + 1, nb.y);
        atomicAdd(g_nrm + 3 * (int64_t)j + 2, nb.z);
    }
    return sb;
}

// 12 running sums of a block -> partials[blockIdx.x][12] (fixed order: wave butterflies, then waves in order)
__device__ __forceinline__ void block_store12(float *acc, float *__restrict__ partials) {
    __shared__ float wsum[BWD_T / 64][12];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 12; ++k) {
        const float v = wave_sum(acc[k]);
        if (lane == 0) wsum[wave][k] = v;
    }
    __syncthreads();
    if (threadIdx.x < 12) {
        float v = 0.0f;
        for (int w = 0; w < BWD_T / 64; ++w) v += wsum[w][threadIdx.x];
        partials[blockIdx.x * 12 + threadIdx.x] = v;
    }
}
__device__ __forceinline__ void acc_outer(float *acc, const f3 g, const f3 s) {
    acc[0] += g.x * s.x; acc[1] += g.x * s.y; acc[2] += g.x * s.z; acc[3] += g.x;
    acc[4] += g.y * s.x; acc[5] += g.y * s.y; acc[6] += g.y * s.z; acc[7] += g.y;
    acc[8] += g.z * s.x; acc[9] += g.z * s.y; acc[10] += g.z * s.z; acc[11] += g.z;
}
__device__ __forceinline__ f3 rot_t(const float *R, const f3 g) {  // R^T g, R row-major 3x3
    return f3{R[0] * g.x + R[3] * g.y + R[6] * g.z, R[1] * g.x + R[4] * g.y + R[7] * g.z, R[2] * g.x + R[5] * g.y + R[8] * g.z};
}

// ---- O(1) adjoints, fp64 on one lane
// adjoint of T = se3_exp(xi) (se3_exp_dev above, both branches) given gT (top 3 rows, row-major 3x4)
__device__ void se3_exp_bwd(const float *xi, const double *gT, double *gxi) {
    const double v[3] = {xi[0], xi[1], xi[2]}, w[3] = {xi[3], xi[4], xi[5]};
    const double Wh[9] = {0.0, -w[2], w[1], w[2], 0.0, -w[0], -w[1], w[0], 0.0};
    const float thf = sqrtf(__fmaf_rn(xi[5], xi[5], __fmaf_rn(xi[4], xi[4], xi[3] * xi[3])));  // the branch the forward took
    double gR[9], gV[9], gt[3] = {gT[3], gT[7], gT[11]};
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) { gR[3 * i + j] = gT[4 * i + j]; gV[3 * i + j] = gt[i] * v[j]; }
    double V[9], gWh[9], gw[3] = {0.0, 0.0, 0.0};
    if (thf < 1e-6f) {
        for (int i = 0; i < 9; ++i) { V[i] = ((i % 4 == 0) ? 1.0 : 0.0) + Wh[i]; gWh[i] = gR[i] + gV[i]; }
    } else {
        const double th = sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
        const double s = sin(th), c = cos(th);
        double W2[9];
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) W2[3 * i + j] = Wh[3 * i] * Wh[j] + Wh[3 * i + 1] * Wh[3 + j] + Wh[3 * i + 2] * Wh[6 + j];
        const double th2 = th * th, th3 = th2 * th, th4 = th2 * th2;
        const double A = s / th, Bc = (1.0 - c) / th2, C = (th - s) / th3;
        double gA = 0.0, gB = 0.0, gC = 0.0, gW2[9];
        for (int i = 0; i < 9; ++i) {
            V[i] = ((i % 4 == 0) ? 1.0 : 0.0) + Bc * Wh[i] + C * W2[i];
            gA += gR[i] * Wh[i];
            gB += gR[i] * W2[i] + gV[i] * Wh[i];
            gC += gV[i] * W2[i];
            gWh[i] = A * gR[i] + Bc * gV[i];
            gW2[i] = Bc * gR[i] + C * gV[i];
        }
        for (int i = 0; i < 3; ++i)      // W2 = Wh Wh : gWh += gW2 Wh^T + Wh^T gW2
            for (int j = 0; j < 3; ++j) {
                double a = 0.0;
                for (int k = 0; k < 3; ++k) a += gW2[3 * i + k] * Wh[3 * j + k] + Wh[3 * k + i] * gW2[3 * k + j];
                gWh[3 * i + j] += a;
            }
        const double dA = (c * th - s) / th2, dB = (s * th - 2.0 * (1.0 - c)) / th3, dC = ((1.0 - c) * th - 3.0 * (th - s)) / th4;
        const double gth = gA * dA + gB * dB + gC * dC;
        for (int k = 0; k < 3; ++k) gw[k] = gth * w[k] / th;
    }
    gw[0] += gWh[7] - gWh[5];
    gw[1] += gWh[2] - gWh[6];
    gw[2] += gWh[3] - gWh[1];
    for (int j = 0; j < 3; ++j) gxi[j] = V[j] * gt[0] + V[3 + j] * gt[1] + V[6 + j] * gt[2];  // V^T gt
    gxi[3] = gw[0]; gxi[4] = gw[1]; gxi[5] = gw[2];
}

// sum of the 12-wide partial rows by a 256-thread block: 16 groups stride over the rows, then 12 threads
// add the 16 group sums in order (nblocks <= BWD_MAXB)
__device__ __forceinline__ void reduce12(const float *__restrict__ partials, int nblocks, float *out_sm) {
    __shared__ float stage[12][17];
    const int k = threadIdx.x & 15, g = threadIdx.x >> 4;
    if (k < 12) {
        float v = 0.0f;
        for (int b = g; b < nblocks; b += 16) v += partials[b * 12 + k];
        stage[k][g] = v;
    }
    __syncthreads();
    if (threadIdx.x < 12) {
        float v = 0.0f;
#pragma unroll
        for (int q = 0; q < 16; ++q) v += stage[threadIdx.x][q];
        out_sm[threadIdx.x] = v;
    }
    __syncthreads();
}

__device__ __forceinline__ void top3_times_Tt(const float *gTn, const float *T, double *out12) {  // (gTn . T^T) rows 0..2
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 4; ++j) {
            double a = 0.0;
            for (int k = 0; k < 4; ++k) a += (double)gTn[4 * i + k] * (double)T[4 * j + k];
            out12[4 * i + j] = a;
        }
}
__device__ __forceinline__ void pull_gT(float *gT, const float *dT) {  // gT <- dT^T gT
    float r[16];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            double a = 0.0;
            for (int k = 0; k < 4; ++k) a += (double)dT[4 * k + i] * (double)gT[4 * k + j];
            r[4 * i + j] = (float)a;
        }
    for (int i = 0; i < 16; ++i) gT[i] = r[i];
}
// xi = (H + damp I)^-1 g : given gxi -> G (Hbar, gbar), returns damp_bar
__device__ double solve_bwd(const float *H, float damp, const float *xi, const double *gxi, float *G) {
    float gx[6], y[6];
    double lu[42];
    for (int i = 0; i < 6; ++i) gx[i] = (float)gxi[i];
    solve6(H, gx, damp, y, lu);  // M symmetric: M^-T = M^-1
    double gd = 0.0;
    for (int i = 0; i < 6; ++i) {
        for (int j = 0; j < 6; ++j) G[6 * i + j] = -y[i] * xi[j];
        G[36 + i] = y[i];
        gd -= (double)y[i] * (double)xi[i];
    }
    return gd;
}

// ---- the O(1) steps of the reverse pass, as device functions on a state in LDS (one lane; fp64 where the forward's
// fp32 value would lose the gradient).  They run FOLDED into the prologue of the wide kernel that follows them, the way
// the forward folds its step into the next association: every block recomputes the step from the previous launch's
// outputs (state and partial sums: complete and visible at kernel start), block 0 alone publishes the new state; state
// and partial sums alternate between two buffers from launch to launch.  Per gradLM iteration that is two launches
// instead of four (S1 + look + S2 + lin were 7.3 + 11.4 + 5.7 + 11.6 us, profiles/r03n_fwd_bwd200_kernel_stats.csv).
enum BwdFold { FOLD_G1 = 1, FOLD_G2 = 2, FOLD_LM = 3 };

// S for one LM iteration (record = the STEP_LM record of that iteration)
__device__ void small_lm(BwdState *Sb, const float *rec, const float *__restrict__ rec_global, const float *sums, int iter) {
    const IcpState *S = rec_state(rec);
    if (rec[REC_ACCEPT] == 0.0f) { Sb->active = 0; return; }
    double gdT[12], gxi[6];
    top3_times_Tt(Sb->gT, S->T, gdT);
    for (int k = 0; k < 12; ++k) gdT[k] += (double)sums[k];
    pull_gT(Sb->gT, S->dT);
    se3_exp_bwd(S->xi, gdT, gxi);
    solve_bwd(S->cur, S->damp, S->xi, gxi, Sb->G);
    Sb->G[42] = 0.0f; Sb->G[43] = 0.0f;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) Sb->R2[3 * i + j] = S->dT[4 * i + j];
    Sb->src_slot = S->p_cur; Sb->nn_slot = S->b_cur; Sb->look_slot = (int)rec[REC_SLOT];
    // the cloud of this iteration was made by the closest earlier accepted iteration (from ITS cloud)
    int prev = -1;
    for (int j = iter - 1; j >= 0 && prev < 0; --j) {
        const float *rj = rec_global - (size_t)(iter - j) * REC_WORDS;  // (earlier records: not in the LDS copy)
        if (rj[REC_ACCEPT] != 0.0f) prev = rec_state(rj)->p_cur;
    }
    Sb->prev_slot = prev;
    Sb->active = 1;
}

// S1 for one gradLM iteration (record = its STEP_GRAD_B record; the next record's head = the state after)
__device__ void small_g1(BwdState *Sb, const float *rec, const float *sums, GradParams gp, int prev_slot) {
    const IcpState *S = rec_state(rec), *Sn = rec_state(rec + REC_WORDS);
    const float err = S->cur[42], new_err = rec[REC_LIN + 42];
    const float raw = new_err - err;
    const float diff = fminf(fmaxf(raw, -70.0f), 70.0f);
    const bool pass = raw >= -70.0f && raw <= 70.0f;  // clamp passes the adjoint inside the range (torch.clamp)
    const double eB = exp(-(double)gp.B * diff), eB2 = exp(-(double)gp.B2 * diff);
    const double F = (double)gp.lambda_min + (double)gp.range / (1.0 + eB);
    const double dF = (double)gp.range * (double)gp.B * eB / ((1.0 + eB) * (1.0 + eB));
    const double sig = pow(1.0 + eB2, -(double)gp.inv_nu);
    const double dsig = (double)gp.inv_nu * (double)gp.B2 * eB2 * pow(1.0 + eB2, -(double)gp.inv_nu - 1.0);
    float sx[6];
    for (int i = 0; i < 6; ++i) sx[i] = (float)sig * S->xi[i];
    double gdT2[12], gsx[6];
    top3_times_Tt(Sb->gT, S->T, gdT2);
    for (int k = 0; k < 12; ++k) gdT2[k] += (double)sums[k];
    pull_gT(Sb->gT, Sn->dT);
    se3_exp_bwd(sx, gdT2, gsx);
    double g_s = 0.0;
    for (int i = 0; i < 6; ++i) { g_s += gsx[i] * (double)S->xi[i]; Sb->gxi[i] = (float)(sig * gsx[i]); }
    const double gdamp_next = Sb->gdamp;
    const double g_diff = pass ? gdamp_next * (double)S->damp * dF + g_s * dsig : 0.0;
    Sb->gdamp = (float)(gdamp_next * F);
    Sb->g_new_err = (float)g_diff;
    Sb->g_err = (float)(-g_diff);
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) { Sb->R2[3 * i + j] = Sn->dT[4 * i + j]; Sb->R1[3 * i + j] = S->dT[4 * i + j]; }
    Sb->src_slot = S->p_cur; Sb->nn_slot = S->b_cur; Sb->look_slot = (int)rec[REC_SLOT];
    Sb->prev_slot = prev_slot;
    Sb->active = 1;
}

// S2: look-ahead step dT1 = exp(xi) -> xi ; then the solve
__device__ void small_g2(BwdState *Sb, const float *rec, const float *sums) {
    const IcpState *S = rec_state(rec);
    double gdT1[12], gxi[6];
    for (int k = 0; k < 12; ++k) gdT1[k] = (double)sums[k];
    se3_exp_bwd(S->xi, gdT1, gxi);
    for (int i = 0; i < 6; ++i) gxi[i] += (double)Sb->gxi[i];
    const double gd = solve_bwd(S->cur, S->damp, S->xi, gxi, Sb->G);
    Sb->gdamp = (float)((double)Sb->gdamp + gd);
    Sb->G[42] = Sb->g_err; Sb->G[43] = 0.0f;
}

// prologue of the wide kernels: the folded small step on an LDS copy of the state; ends with a barrier.
// Everything the step reads -- the state, its tape record with the head of the next one, the partial rows -- is requested in
// ONE batch at kernel start and handed over through LDS: the step runs on one lane, and every global word it used to
// fetch for itself (the record's state, sums, flags: a dozen dependent round trips) is an LDS read now; the rows' loads
// used to follow each other through a four-deep loop (same order of summation as reduce12: the sums do not change).
__device__ __forceinline__ void bwd_fold(BwdState &sb, const BwdState *__restrict__ Sb_in, BwdState *__restrict__ Sb_out, int fold,
                                         const float *__restrict__ rec, const float *__restrict__ partials_in, int nblocks,
                                         GradParams gp, int arg) {
    __shared__ float sums[12];
    __shared__ float rec_sm[2 * REC_WORDS];
    __shared__ float stage[12][17];
    constexpr int kWords = sizeof(BwdState) / 4, RU = 8;
    static_assert(kWords <= BWD_T, "state copied by one pass of the block");
    static_assert(BWD_T < 2 * REC_WORDS && 2 * REC_WORDS <= 2 * BWD_T, "record pair copied by two loads per thread");
    const int t = threadIdx.x, k = t & 15, g = t >> 4, kc = min(k, 11), last = max(nblocks - 1, 0);
    const int sw = reinterpret_cast<const int *>(Sb_in)[min(t, kWords - 1)];
    const float r0 = rec[t], r1 = rec[BWD_T + min(t, 2 * REC_WORDS - BWD_T - 1)];
    float a[RU];
#pragma unroll
    for (int u = 0; u < RU; ++u) a[u] = partials_in[min(g + 16 * u, last) * 12 + kc];
    if (t < kWords) reinterpret_cast<int *>(&sb)[t] = sw;
    rec_sm[t] = r0;
    if (t < 2 * REC_WORDS - BWD_T) rec_sm[BWD_T + t] = r1;
    float v = 0.0f;
#pragma unroll
    for (int u = 0; u < RU; ++u) v += (g + 16 * u < nblocks) ? a[u] : 0.0f;
    for (int b0 = g + 16 * RU; b0 < nblocks; b0 += 16 * RU) {  // (more than 128 rows: further rounds)
#pragma unroll
        for (int u = 0; u < RU; ++u) a[u] = partials_in[min(b0 + 16 * u, last) * 12 + kc];
#pragma unroll
        for (int u = 0; u < RU; ++u) v += (b0 + 16 * u < nblocks) ? a[u] : 0.0f;
    }
    if (k < 12) stage[k][g] = v;
    __syncthreads();
    if (t < 12) {
        float q = 0.0f;
#pragma unroll
        for (int j = 0; j < 16; ++j) q += stage[t][j];
        sums[t] = q;
    }
    __syncthreads();  // sb, rec_sm and sums are visible
    if (t == 0) {
        if (fold == FOLD_G1) small_g1(&sb, rec_sm, sums, gp, arg);
        else if (fold == FOLD_G2) small_g2(&sb, rec_sm, sums);
        else small_lm(&sb, rec_sm, rec, sums, arg);
    }
    __syncthreads();
    if (blockIdx.x == 0 && t < kWords) reinterpret_cast<int *>(Sb_out)[t] = reinterpret_cast<const int *>(&sb)[t];
}

// B (gradLM), with S1 folded in: adjoint of new_err = e(look, NN(look)); gP_i <- R2^T gP_i + R1^T glook_i ; sums glook (x) s
__global__ __launch_bounds__(BWD_T) void bwd_look_k(const BwdState *__restrict__ Sb_in, BwdState *__restrict__ Sb_out,
                                                    const float *__restrict__ rec, const float *__restrict__ partials_in, int nblocks,
                                                    GradParams gp, int prev_slot, LoopBufs B, const int32_t *__restrict__ d_ns,
                                                    const float *__restrict__ tgt, const float *__restrict__ nrm, float thresh,
                                                    float *__restrict__ gP, float *__restrict__ g_tgt, float *__restrict__ g_nrm,
                                                    float *__restrict__ partials) {
    __shared__ BwdState sb;
    __shared__ float G[44];
    bwd_fold(sb, Sb_in, Sb_out, FOLD_G1, rec, partials_in, nblocks, gp, prev_slot);
    if (threadIdx.x < 44) G[threadIdx.x] = (threadIdx.x == 42) ? sb.g_new_err : 0.0f;
    __syncthreads();
    const float *R = sb.R2, *R1 = sb.R1;
    float acc[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) acc[k] = 0.0f;
    const float *src = B.P(sb.src_slot), *look = B.P(sb.look_slot);
    const unsigned long long *nn = B.N(sb.look_slot);
    const int ns = *d_ns;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < ns; i += gridDim.x * blockDim.x) {
        const Row r = make_row(look, tgt, nrm, nn, i, ns, thresh);
        f3 gl{0.0f, 0.0f, 0.0f};
        if (r.valid) gl = lin_point_bwd(G, r, ld3(look, i), (uint32_t)(nn[i] & 0xffffffffu), tgt, nrm, g_tgt, g_nrm);
        acc_outer(acc, gl, ld3(src, i));
        const f3 a = rot_t(R, ld3(gP, i)), b = rot_t(R1, gl);
        st3(gP, i, f3{a.x + b.x, a.y + b.y, a.z + b.z});
    }
    block_store12(acc, partials);
}

// C, with S2 (gradLM) or S (LM) folded in: gP_i <- (rotate ? R2^T gP_i : gP_i) + adjoint of (H, g, e) at the iteration's
// source cloud; sums of gP (x) predecessor cloud for the small step of the iteration that made this cloud
__global__ __launch_bounds__(BWD_T) void bwd_lin_k(const BwdState *__restrict__ Sb_in, BwdState *__restrict__ Sb_out, int fold,
                                                   const float *__restrict__ rec, const float *__restrict__ partials_in, int nblocks,
                                                   int iter, int rotate, LoopBufs B, const float *__restrict__ user_src,
                                                   const int32_t *__restrict__ d_ns, const float *__restrict__ tgt,
                                                   const float *__restrict__ nrm, float thresh, float *__restrict__ gP,
                                                   float *__restrict__ g_tgt, float *__restrict__ g_nrm,
                                                   float *__restrict__ partials) {
    __shared__ BwdState sb;
    bwd_fold(sb, Sb_in, Sb_out, fold, rec, partials_in, nblocks, GradParams{}, iter);
    if (!sb.active) {  // rejected LM iteration: gP stays as it is, the pending sums are handed on unchanged
        if (threadIdx.x < 12) partials[blockIdx.x * 12 + threadIdx.x] = partials_in[blockIdx.x * 12 + threadIdx.x];
        return;
    }
    const float *G = sb.G, *R = sb.R2;
    const float *src = B.P(sb.src_slot);
    const float *prev = sb.prev_slot >= 0 ? B.P(sb.prev_slot) : user_src;
    const unsigned long long *nn = B.N(sb.nn_slot);
    const int ns = *d_ns;
    float acc[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) acc[k] = 0.0f;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < ns; i += gridDim.x * blockDim.x) {
        const Row r = make_row(src, tgt, nrm, nn, i, ns, thresh);
        f3 g = ld3(gP, i);
        if (rotate) g = rot_t(R, g);
        if (r.valid) {
            const f3 sb_ = lin_point_bwd(G, r, ld3(src, i), (uint32_t)(nn[i] & 0xffffffffu), tgt, nrm, g_tgt, g_nrm);
            g.x += sb_.x; g.y += sb_.y; g.z += sb_.z;
        }
        st3(gP, i, g);
        acc_outer(acc, g, ld3(prev, i));
    }
    block_store12(acc, partials);
}

struct BwdWs {
    BwdState *S[2];       // double-buffered across launches (bwd_fold)
    float *gP, *partials[2];
};
static inline size_t bwd_ws_layout(int max_ns, void *ws, BwdWs *out) {
    const size_t sS = align_up(sizeof(BwdState), 256), sG = align_up((size_t)max_ns * 12, 256), sP = align_up((size_t)BWD_MAXB * 12 * 4, 256);
    if (ws && out) {
        char *p = (char *)ws;
        out->S[0] = (BwdState *)p; out->S[1] = (BwdState *)(p + sS);
        out->gP = (float *)(p + 2 * sS);
        out->partials[0] = (float *)(p + 2 * sS + sG); out->partials[1] = (float *)(p + 2 * sS + sG + sP);
    }
    return 2 * sS + sG + 2 * sP;
}

__global__ void zero_rows_k(float *__restrict__ a, float *__restrict__ b, const int32_t *__restrict__ d_n, int cap) {
    const int n = 3 * min(*d_n, cap);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        if (a) a[i] = 0.0f;
        if (b) b[i] = 0.0f;
    }
}

// one launch for the reverse pass's preparations: gP and the first partial rows zeroed (nothing depends on the final
// cloud), the target / normal adjoints zeroed over the rows that exist (max_nt may be a generous capacity), the state set
__global__ void bwd_begin_k(BwdState *Sb, const float *__restrict__ grad_T, float *__restrict__ gP, int n_gp, float *__restrict__ partials,
                            int n_part, float *__restrict__ g_tgt, float *__restrict__ g_nrm, const int32_t *__restrict__ d_nt, int cap) {
    const int tid = blockIdx.x * blockDim.x + threadIdx.x, stride = gridDim.x * blockDim.x;
    for (int i = tid; i < n_gp; i += stride) gP[i] = 0.0f;
    for (int i = tid; i < n_part; i += stride) partials[i] = 0.0f;
    const int n = 3 * min(*d_nt, cap);
    for (int i = tid; i < n; i += stride) {
        if (g_tgt) g_tgt[i] = 0.0f;
        if (g_nrm) g_nrm[i] = 0.0f;
    }
    if (blockIdx.x == 0) {
        if (threadIdx.x < 16) Sb->gT[threadIdx.x] = grad_T[threadIdx.x];
        if (threadIdx.x == 0) { Sb->gdamp = 0.0f; Sb->active = 0; }
    }
}

// last: through src0 = init_T . user_src; block 0 also adds the last partial sums up into the adjoint of init_T (the T
// chain starts at init_T and src0 = init_T . user_src)
__global__ __launch_bounds__(BWD_T) void bwd_finish_k(const float *__restrict__ init_T, const int32_t *__restrict__ d_ns,
                                                      const float *__restrict__ gP, float *__restrict__ g_src,
                                                      const BwdState *__restrict__ Sb, const float *__restrict__ partials, int nblocks,
                                                      float *__restrict__ g_init_T) {
    __shared__ float R[9];
    __shared__ float sums[12];
    if (threadIdx.x < 9) R[threadIdx.x] = init_T[4 * (threadIdx.x / 3) + threadIdx.x % 3];
    __syncthreads();
    const int ns = *d_ns;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < ns; i += gridDim.x * blockDim.x) st3(g_src, i, rot_t(R, ld3(gP, i)));
    if (blockIdx.x == 0) {  // (block-uniform)
        reduce12(partials, nblocks, sums);
        if (threadIdx.x < 16) g_init_T[threadIdx.x] = Sb->gT[threadIdx.x] + (threadIdx.x < 12 ? sums[threadIdx.x] : 0.0f);
    }
}

static int icp_backward_run(bool grad, const float *src, const int32_t *d_ns, int max_ns, const float *tgt, const float *nrm,
                            const int32_t *d_nt, int max_nt, const float *init_T, int numiters, float thresh, GradParams gp, const void *tape,
                            size_t tape_bytes, const float *grad_T, float *g_src, float *g_tgt, float *g_nrm, float *g_init_T,
                            void *ws, size_t ws_bytes, hipStream_t st) {
    const char *name = "gs_icp_backward";
    GS_REQUIRE(src && d_ns && tgt && nrm && d_nt && init_T && tape && grad_T && g_src && g_init_T, "%s: NULL argument", name);
    GS_REQUIRE(max_ns > 0 && max_nt > 0 && numiters >= 0, "%s: bad sizes", name);
    if (!ws || ws_bytes < bwd_ws_layout(max_ns, nullptr, nullptr)) {
        set_error("%s: workspace too small", name);
        return GS_ERR_WORKSPACE_TOO_SMALL;
    }
    GS_REQUIRE(tape_bytes >= tape_layout(grad, max_ns, numiters, nullptr, nullptr), "%s: tape too small", name);
    Tape tp;
    tape_layout(grad, max_ns, numiters, (void *)tape, &tp);
    BwdWs w;
    bwd_ws_layout(max_ns, ws, &w);
    const int nb = min(cdiv(max_ns, BWD_T), BWD_MAXB);
    int cur = 0;  // buffer the next launch READS its state / the previous sums from
    hipLaunchKernelGGL(bwd_begin_k, dim3(min(cdiv(3 * max(max_nt, max_ns), 256), 1024)), dim3(256), 0, st, w.S[0], grad_T, w.gP, 3 * max_ns,
                       w.partials[0], nb * 12, g_tgt, g_nrm, d_nt, max_nt);
    for (int k = numiters - 1; k >= 0; --k) {
        if (!grad) {
            const float *rec = tp.rec + (size_t)(1 + k) * REC_WORDS;
            hipLaunchKernelGGL(bwd_lin_k, dim3(nb), dim3(BWD_T), 0, st, (const BwdState *)w.S[cur], w.S[1 - cur], (int)FOLD_LM, rec,
                               (const float *)w.partials[cur], nb, k, 1, tp.B, src, d_ns, tgt, nrm, thresh, w.gP, g_tgt, g_nrm,
                               w.partials[1 - cur]);
            cur = 1 - cur;
        } else {
            const float *rec = tp.rec + (size_t)(1 + 2 * k) * REC_WORDS;
            // slots of the gradLM loop are fixed: cloud k lives in slot 0 (k = 0) or 2k
            hipLaunchKernelGGL(bwd_look_k, dim3(nb), dim3(BWD_T), 0, st, (const BwdState *)w.S[cur], w.S[1 - cur], rec,
                               (const float *)w.partials[cur], nb, gp, k == 0 ? -1 : (k == 1 ? 0 : 2 * (k - 1)), tp.B, d_ns, tgt, nrm, thresh,
                               w.gP, g_tgt, g_nrm, w.partials[1 - cur]);
            cur = 1 - cur;
            hipLaunchKernelGGL(bwd_lin_k, dim3(nb), dim3(BWD_T), 0, st, (const BwdState *)w.S[cur], w.S[1 - cur], (int)FOLD_G2, rec,
                               (const float *)w.partials[cur], nb, 0, 0, tp.B, src, d_ns, tgt, nrm, thresh, w.gP, g_tgt, g_nrm,
                               w.partials[1 - cur]);
            cur = 1 - cur;
        }
    }
    hipLaunchKernelGGL(bwd_finish_k, dim3(nb), dim3(BWD_T), 0, st, init_T, d_ns, (const float *)w.gP, g_src, (const BwdState *)w.S[cur],
                       (const float *)w.partials[cur], nb, g_init_T);
    GS_LAUNCH_CHECK(name);
    return GS_OK;
}

int icp_localize_run(int grad_lm, const float *src, const int32_t *d_ns, int max_ns, const float *tgt, const float *nrm,
                     const int32_t *d_nt, int max_nt, int numiters, float damp, float thresh, float lambda_max, float Bp,
                     float B2, float nu, const gs_icp_hints *hints, float *out_T, void *ws, size_t ws_bytes, hipStream_t st,
                     void *tape, size_t tape_bytes, const float *compose_right, float *compose_out) {
    const GradParams gp = grad_lm ? GradParams{(float)(1.0 / (double)lambda_max), (float)((double)lambda_max - 1.0 / (double)lambda_max),
                                               Bp, B2, (float)(1.0 / (double)nu)}
                                  : GradParams{0.5f, 1.5f, 1.0f, 1.0f, 0.005f};
    return icp_run(grad_lm != 0, src, d_ns, max_ns, tgt, nrm, d_nt, max_nt, nullptr, numiters, damp, thresh, gp, hints, out_T, nullptr,
                   nullptr, ws, ws_bytes, st, "gs_slam_localize/icp", tape, tape_bytes, compose_right, compose_out);
}

}  // namespace gs

using namespace gs;

extern "C" {

#ifdef GS_DIAG_STAMPS
int gs_diag_set_buffer(void *p) {
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_diag), &p, sizeof(p));
}
#endif

void gs_set_grid_search(int on) { g_grid_mode = on; }
void gs_set_tile_points(int n) { g_tile_points = n; }
int gs_icp_launch_geometry(int max_ns, int have_hints, int *blocks, int *tile_points_dense, int *partial_rows) {
    GS_REQUIRE(max_ns > 0, "gs_icp_launch_geometry: max_ns must be positive");
    (void)have_hints;
    const int tp = loop_tile_points();
    if (blocks) *blocks = cdiv(max_ns, tp);
    if (tile_points_dense) *tile_points_dense = tp;
    if (partial_rows) *partial_rows = partial_rows_alloc(max_ns);
    return GS_OK;
}

int gs_loop_counts(unsigned int *out4, int reset) {
    GS_REQUIRE(out4, "gs_loop_counts: NULL argument");
    GS_HIP(hipMemcpyFromSymbol(out4, HIP_SYMBOL(g_loop_counts), 16), "gs_loop_counts");  // synchronises with the device
    if (reset) {
        const unsigned int z[4] = {0, 0, 0, 0};
        GS_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_loop_counts), z, 16), "gs_loop_counts/reset");
    }
    return GS_OK;
}

void gs_profile_enable(int on) {
    g_prof.on = on != 0;
    for (int t = 0; t < 2; ++t) { g_prof.used[t] = 0; g_prof.total_ms[t] = 0.0; g_prof.count[t] = 0; }
}

// Fold the events recorded so far into the totals (the caller must have synchronised the stream)
// and return, for tag 0 (association kernel) / 1 (linearise kernel), launches and total milliseconds.
int gs_profile_read(int tag, long *launches, double *total_ms) {
    GS_REQUIRE(tag == 0 || tag == 1, "gs_profile_read: tag must be 0 or 1");
    for (int t = 0; t < 2; ++t) {
        for (size_t i = 0; i < g_prof.used[t]; ++i) {
            float ms = 0.0f;
            if (hipEventElapsedTime(&ms, g_prof.ev[t][0][i], g_prof.ev[t][1][i]) == hipSuccess) {
                g_prof.total_ms[t] += ms;
                g_prof.count[t] += 1;
            }
        }
        g_prof.used[t] = 0;
    }
    if (launches) *launches = g_prof.count[tag];
    if (total_ms) *total_ms = g_prof.total_ms[tag];
    return GS_OK;
}

size_t gs_knn1_ws_bytes(int max_nt) { return boxes_bytes(max_nt); }

int gs_knn1(const float *src, const int32_t *d_ns, int max_ns, const float *tgt, const int32_t *d_nt, int max_nt,
            uint64_t *best, void *ws, size_t ws_bytes, gs_stream_t stream) {
    GS_REQUIRE(src && d_ns && tgt && d_nt && best, "gs_knn1: NULL argument");
    GS_REQUIRE(max_ns >= 0 && max_nt >= 0, "gs_knn1: negative size");
    if (max_ns == 0) return GS_OK;
    hipStream_t st = (hipStream_t)stream;
    if (max_nt == 0) {
        hipLaunchKernelGGL(fill_u64_k, dim3(min(cdiv(max_ns, 256), 1024)), dim3(256), 0, st, (unsigned long long *)best, max_ns, KEY_NONE);
        GS_LAUNCH_CHECK("gs_knn1/fill");
        return GS_OK;
    }
    if (!ws || ws_bytes < boxes_bytes(max_nt)) {
        set_error("gs_knn1: workspace too small (%zu < %zu)", ws_bytes, boxes_bytes(max_nt));
        return GS_ERR_WORKSPACE_TOO_SMALL;
    }
    hipLaunchKernelGGL(tgt_boxes_k, dim3(cdiv(max_nt, 64)), dim3(64), 0, st, tgt, d_nt, (float *)ws);
    GS_LAUNCH_CHECK("gs_knn1/boxes");
    hipLaunchKernelGGL(knn1_box_k, dim3(cdiv(max_ns, 64)), dim3(KNN_BT), 0, st, src, d_ns, tgt, (const float *)ws, d_nt,
                       (unsigned long long *)best);
    GS_LAUNCH_CHECK("gs_knn1");
    return GS_OK;
}

int gs_knn1_bruteforce(const float *src, const int32_t *d_ns, int max_ns, const float *tgt, const int32_t *d_nt,
                       int max_nt, uint64_t *best, gs_stream_t stream) {
    GS_REQUIRE(src && d_ns && tgt && d_nt && best, "gs_knn1_bruteforce: NULL argument");
    GS_REQUIRE(max_ns >= 0 && max_nt >= 0, "gs_knn1_bruteforce: negative size");
    if (max_ns == 0) return GS_OK;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(fill_u64_k, dim3(min(cdiv(max_ns, 256), 1024)), dim3(256), 0, st, (unsigned long long *)best, max_ns, KEY_NONE);
    GS_LAUNCH_CHECK("gs_knn1_bruteforce/fill");
    if (max_nt == 0) return GS_OK;
    const int nsplit = knn_nsplit_brute(max_ns, max_nt);
    hipLaunchKernelGGL(knn1_brute_k, dim3(cdiv(max_ns, KNN_T), nsplit), dim3(KNN_T), 0, st, src, d_ns, tgt, d_nt, nsplit,
                       (unsigned long long *)best);
    GS_LAUNCH_CHECK("gs_knn1_bruteforce");
    return GS_OK;
}

int gs_knn1_unpack(const uint64_t *best, const int32_t *d_ns, int max_ns, float *dist2, int64_t *idx, gs_stream_t stream) {
    GS_REQUIRE(best && d_ns && max_ns >= 0, "gs_knn1_unpack: bad arguments");
    if (max_ns == 0) return GS_OK;
    hipLaunchKernelGGL(knn_unpack_k, dim3(min(cdiv(max_ns, 256), 1024)), dim3(256), 0, (hipStream_t)stream,
                       (const unsigned long long *)best, d_ns, dist2, idx);
    GS_LAUNCH_CHECK("gs_knn1_unpack");
    return GS_OK;
}

size_t gs_icp_linearize_ws_bytes(int max_ns) { (void)max_ns; return align_up((size_t)LIN_MAXB * NACC * 4, 256); }

int gs_icp_linearize(const float *src, const int32_t *d_ns, int max_ns, const float *tgt, const float *tgt_normals,
                     const uint64_t *best, float dist_thresh, float *out44, void *ws, size_t ws_bytes,
                     gs_stream_t stream) {
    GS_REQUIRE(src && d_ns && tgt && tgt_normals && best && out44, "gs_icp_linearize: NULL argument");
    GS_REQUIRE(max_ns >= 0, "gs_icp_linearize: negative size");
    if (!ws || ws_bytes < gs_icp_linearize_ws_bytes(max_ns)) {
        set_error("gs_icp_linearize: workspace too small");
        return GS_ERR_WORKSPACE_TOO_SMALL;
    }
    hipStream_t st = (hipStream_t)stream;
    const int nb = lin_blocks(max_ns);  // capped at LIN_MAXB: large clouds grid-stride
    hipLaunchKernelGGL(linearize_k, dim3(nb), dim3(LIN_T), 0, st, src, d_ns, tgt, tgt_normals,
                       (const unsigned long long *)best, dist_thresh, (float *)ws);
    GS_LAUNCH_CHECK("gs_icp_linearize");
    hipLaunchKernelGGL(finalize44_k, dim3(1), dim3(1024), 0, st, (const float *)ws, nb, out44);
    GS_LAUNCH_CHECK("gs_icp_linearize/finalize");
    return GS_OK;
}

int gs_icp_rows(const float *src, const int32_t *d_ns, int max_ns, const float *tgt, const float *tgt_normals,
                const uint64_t *best, float dist_thresh, float *A, float *b, uint8_t *keep, gs_stream_t stream) {
    GS_REQUIRE(src && d_ns && tgt && tgt_normals && best && A && b && keep, "gs_icp_rows: NULL argument");
    if (max_ns <= 0) return GS_OK;
    hipLaunchKernelGGL(icp_rows_k, dim3(min(cdiv(max_ns, 256), 2048)), dim3(256), 0, (hipStream_t)stream, src, d_ns, tgt,
                       tgt_normals, (const unsigned long long *)best, dist_thresh, A, b, keep);
    GS_LAUNCH_CHECK("gs_icp_rows");
    return GS_OK;
}

int gs_icp_linearize_backward(const float *src, const int32_t *d_ns, int max_ns, const float *tgt,
                              const float *tgt_normals, const uint64_t *best, float dist_thresh, const float *g_out43,
                              float *g_src, float *g_tgt, float *g_normals, gs_stream_t stream) {
    GS_REQUIRE(src && d_ns && tgt && tgt_normals && best && g_out43, "gs_icp_linearize_backward: NULL argument");
    if (max_ns <= 0) return GS_OK;
    hipLaunchKernelGGL(linearize_bwd_k, dim3(min(cdiv(max_ns, 256), 2048)), dim3(256), 0, (hipStream_t)stream, src, d_ns,
                       tgt, tgt_normals, (const unsigned long long *)best, dist_thresh, g_out43, g_src, g_tgt, g_normals);
    GS_LAUNCH_CHECK("gs_icp_linearize_backward");
    return GS_OK;
}

int gs_transform_points(const float *pts, const int32_t *d_n, int max_n, const float *T, float *out, gs_stream_t stream) {
    GS_REQUIRE(pts && d_n && T && out && max_n >= 0, "gs_transform_points: bad arguments");
    if (max_n == 0) return GS_OK;
    hipLaunchKernelGGL(transform_k, dim3(min(cdiv(max_n, 256), 2048)), dim3(256), 0, (hipStream_t)stream, pts, d_n, T, out);
    GS_LAUNCH_CHECK("gs_transform_points");
    return GS_OK;
}

size_t gs_icp_ws_bytes(int max_ns, int max_nt) {
    return icp_ws_layout(max_ns > 0 ? max_ns : 1, max_nt > 0 ? max_nt : 1, nullptr, nullptr);
}

int gs_icp_point_to_plane(const float *src, const int32_t *d_ns, int max_ns, const float *tgt, const float *tgt_normals,
                          const int32_t *d_nt, int max_nt, const float *init_T, int numiters, float damp,
                          float dist_thresh, const gs_icp_hints *hints, float *out_T, uint64_t *best_last, float *trace,
                          void *ws, size_t ws_bytes, gs_stream_t stream) {
    return icp_run(false, src, d_ns, max_ns, tgt, tgt_normals, d_nt, max_nt, init_T, numiters, damp, dist_thresh,
                   GradParams{0.5f, 1.5f, 1.0f, 1.0f, 0.005f}, hints, out_T, best_last, trace, ws, ws_bytes,
                   (hipStream_t)stream, "gs_icp_point_to_plane");
}

int gs_icp_point_to_plane_grad(const float *src, const int32_t *d_ns, int max_ns, const float *tgt,
                               const float *tgt_normals, const int32_t *d_nt, int max_nt, const float *init_T,
                               int numiters, float damp, float dist_thresh, float lambda_max, float B, float B2, float nu,
                               const gs_icp_hints *hints, float *out_T, uint64_t *best_last, float *trace, void *ws,
                               size_t ws_bytes, gs_stream_t stream) {
    return icp_run(true, src, d_ns, max_ns, tgt, tgt_normals, d_nt, max_nt, init_T, numiters, damp, dist_thresh,
                   GradParams{(float)(1.0 / (double)lambda_max), (float)((double)lambda_max - 1.0 / (double)lambda_max), B, B2,
                              (float)(1.0 / (double)nu)},
                   hints, out_T, best_last, trace, ws, ws_bytes, (hipStream_t)stream, "gs_icp_point_to_plane_grad");
}


static inline GradParams make_grad_params(float lambda_max, float B, float B2, float nu) {
    return GradParams{(float)(1.0 / (double)lambda_max), (float)((double)lambda_max - 1.0 / (double)lambda_max), B, B2,
                      (float)(1.0 / (double)nu)};
}

size_t gs_icp_tape_bytes(int max_ns, int numiters, int grad_lm) {
    return tape_layout(grad_lm != 0, max_ns > 0 ? max_ns : 1, numiters, nullptr, nullptr);
}

int gs_icp_point_to_plane_taped(const float *src, const int32_t *d_ns, int max_ns, const float *tgt, const float *tgt_normals,
                                const int32_t *d_nt, int max_nt, const float *init_T, int numiters, float damp,
                                float dist_thresh, int grad_lm, float lambda_max, float B, float B2, float nu,
                                const gs_icp_hints *hints, float *out_T, uint64_t *best_last, void *tape, size_t tape_bytes,
                                void *ws, size_t ws_bytes, gs_stream_t stream) {
    GS_REQUIRE(tape, "gs_icp_point_to_plane_taped: NULL tape");
    return icp_run(grad_lm != 0, src, d_ns, max_ns, tgt, tgt_normals, d_nt, max_nt, init_T, numiters, damp, dist_thresh,
                   grad_lm ? make_grad_params(lambda_max, B, B2, nu) : GradParams{0.5f, 1.5f, 1.0f, 1.0f, 0.005f}, hints, out_T,
                   best_last, nullptr, ws, ws_bytes, (hipStream_t)stream, "gs_icp_point_to_plane_taped", tape, tape_bytes);
}

size_t gs_icp_backward_ws_bytes(int max_ns) { return bwd_ws_layout(max_ns > 0 ? max_ns : 1, nullptr, nullptr); }

int gs_icp_point_to_plane_backward(const float *src, const int32_t *d_ns, int max_ns, const float *tgt, const float *tgt_normals,
                                   const int32_t *d_nt, int max_nt, const float *init_T, int numiters, float dist_thresh, int grad_lm,
                                   float lambda_max, float B, float B2, float nu, const void *tape, size_t tape_bytes,
                                   const float *grad_T, float *grad_src, float *grad_tgt, float *grad_normals,
                                   float *grad_init_T, void *ws, size_t ws_bytes, gs_stream_t stream) {
    if (numiters == 0) {  // T = init_T, nothing else depends on the inputs
        GS_REQUIRE(grad_T && grad_src && grad_init_T && d_nt && max_ns > 0 && max_nt > 0, "gs_icp_point_to_plane_backward: bad arguments");
        hipStream_t st = (hipStream_t)stream;
        GS_HIP(hipMemsetAsync(grad_src, 0, (size_t)max_ns * 12, st), "gs_icp_point_to_plane_backward");
        if (grad_tgt || grad_normals)
            hipLaunchKernelGGL(zero_rows_k, dim3(min(cdiv(3 * max_nt, 256), 1024)), dim3(256), 0, st, grad_tgt, grad_normals, d_nt, max_nt);
        GS_HIP(hipMemcpyAsync(grad_init_T, grad_T, 64, hipMemcpyDeviceToDevice, st), "gs_icp_point_to_plane_backward");
        return GS_OK;
    }
    return icp_backward_run(grad_lm != 0, src, d_ns, max_ns, tgt, tgt_normals, d_nt, max_nt, init_T, numiters, dist_thresh,
                            grad_lm ? make_grad_params(lambda_max, B, B2, nu) : GradParams{0.5f, 1.5f, 1.0f, 1.0f, 0.005f}, tape,
                            tape_bytes, grad_T, grad_src, grad_tgt, grad_normals, grad_init_T, ws, ws_bytes, (hipStream_t)stream);
}

}  // extern "C"
