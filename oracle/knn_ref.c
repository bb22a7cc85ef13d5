/* oracle/knn_ref.c -- TEST INFRASTRUCTURE (see oracle/__init__.py).
 *
 * CPU restatement of the K=1 nearest-neighbour search the reference obtains from the
 * third-party dependency chamferdist==1.0.0 (requirements.txt:2; call site
 * gradslam/odometry/icputils.py:200-201).  chamferdist's source is NOT in the reference tree;
 * its knn_points is the pytorch3d CPU kernel, whose published contract is restated here:
 *
 *   for every source point i: scan target points j = 0..Nt-1 in ascending order,
 *   d = ((sx-tx)^2 + (sy-ty)^2) + (sz-tz)^2 accumulated in that order in fp32 (no FMA),
 *   keep (d, j) iff d < best  (strict: the lowest index wins ties);
 *   output the SQUARED distance (fp32) and the index (int64).
 *
 * Per-point indices are parity-unpinned by the reference's own tests (they only pin the
 * converged pose).  Compile with -ffp-contract=off so the compiler cannot fuse the mul/add.
 */
#include <stdint.h>
#include <math.h>

void knn1_ref(const float *src, int64_t ns, const float *tgt, int64_t nt,
              float *out_d2, int64_t *out_idx)
{
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < ns; ++i) {
        const float sx = src[3 * i], sy = src[3 * i + 1], sz = src[3 * i + 2];
        float best = INFINITY;
        int64_t bj = 0;
        for (int64_t j = 0; j < nt; ++j) {
            const float dx = sx - tgt[3 * j];
            const float dy = sy - tgt[3 * j + 1];
            const float dz = sz - tgt[3 * j + 2];
            const float d = (dx * dx + dy * dy) + dz * dz;
            if (d < best) { best = d; bj = j; }
        }
        out_d2[i] = best;
        out_idx[i] = bj;
    }
}

/* Double-precision checker used by the tests to bound the fp32 result independently:
 * returns the exact (fp64) squared distance to the nearest target. */
void knn1_ref_f64(const float *src, int64_t ns, const float *tgt, int64_t nt, double *out_d2)
{
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < ns; ++i) {
        const double sx = src[3 * i], sy = src[3 * i + 1], sz = src[3 * i + 2];
        double best = INFINITY;
        for (int64_t j = 0; j < nt; ++j) {
            const double dx = sx - tgt[3 * j], dy = sy - tgt[3 * j + 1], dz = sz - tgt[3 * j + 2];
            const double d = dx * dx + dy * dy + dz * dz;
            if (d < best) best = d;
        }
        out_d2[i] = best;
    }
}

/* The same scan, sixteen source points at a time (gcc vector extensions: every lane runs the loop above on its own
 * point -- same operations in the same order, targets ascending, strict '<' -- so the result is bit-identical to
 * knn1_ref; tests/test_oracle_golden.py checks that).  Used where the serial loop is too slow to be a checker:
 * the 640x480 golden generators (tools/gen_golden_c3.py) and the full-size parity tests. */
typedef float vf __attribute__((vector_size(32)));
typedef int32_t vi __attribute__((vector_size(32)));
#define KW 8

__attribute__((target_clones("avx2", "default")))
void knn1_ref_wide(const float *src, int64_t ns, const float *tgt, int64_t nt,
                   float *out_d2, int64_t *out_idx)
{
    const int64_t nblk = (ns + 2 * KW - 1) / (2 * KW);
#pragma omp parallel for schedule(dynamic, 4)
    for (int64_t blk = 0; blk < nblk; ++blk) {
        vf sx[2], sy[2], sz[2], best[2];
        vi bj[2];
        for (int h = 0; h < 2; ++h)
            for (int l = 0; l < KW; ++l) {
                int64_t i = blk * 2 * KW + h * KW + l;
                if (i >= ns) i = ns - 1;
                sx[h][l] = src[3 * i]; sy[h][l] = src[3 * i + 1]; sz[h][l] = src[3 * i + 2];
                best[h][l] = INFINITY; bj[h][l] = 0;
            }
        for (int64_t j = 0; j < nt; ++j) {
            const float tx = tgt[3 * j], ty = tgt[3 * j + 1], tz = tgt[3 * j + 2];
            const vi jj = {(int32_t)j, (int32_t)j, (int32_t)j, (int32_t)j, (int32_t)j, (int32_t)j, (int32_t)j, (int32_t)j};
            for (int h = 0; h < 2; ++h) {
                const vf dx = sx[h] - tx, dy = sy[h] - ty, dz = sz[h] - tz;
                const vf d = (dx * dx + dy * dy) + dz * dz;
                const vi m = d < best[h];
                best[h] = (vf)(((vi)d & m) | ((vi)best[h] & ~m));
                bj[h] = (jj & m) | (bj[h] & ~m);
            }
        }
        for (int h = 0; h < 2; ++h)
            for (int l = 0; l < KW; ++l) {
                const int64_t i = blk * 2 * KW + h * KW + l;
                if (i < ns) { out_d2[i] = best[h][l]; out_idx[i] = bj[h][l]; }
            }
    }
}
