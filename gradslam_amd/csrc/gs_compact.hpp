// gs_compact.hpp -- order-preserving (stable) stream compaction skeleton.
//
// The reference selects rows with boolean-mask indexing (x[mask]) whose output order is the input
// order; map-point order and (h,w) row-major append order are user-visible, so the compaction
// must be stable.  Launches: per-block counts -> [single-block scan] -> ordered write.  The
// predicate is re-evaluated in the write pass (cheap, inputs are L2/MALL resident) instead of
// materialising flags.  No inter-workgroup hand-off inside a launch, so no coherence protocol.
// Up to kSelfScanBlocks blocks the scan is not a launch of its own: a dependent launch costs ~4.5 us of
// stream time on this part however small, whereas every write block adding up the counts of the blocks in
// front of it (<= 1024 L2-resident ints, one coalesced read + a block reduction) costs well under 1 us.
#pragma once

#include <type_traits>

#include "gs_common.hpp"

namespace gs {

// a Writer may also provide skip(i), called for every row the predicate rejects (used by the adjoint of a
// compaction, which must write zeros there)
template <class W, class = void>
struct writer_has_skip : std::false_type {};
template <class W>
struct writer_has_skip<W, std::void_t<decltype(&W::skip)>> : std::true_type {};
// a Pred may provide block_init(pass, bid, nb), run once by every block of both passes (pass 0 = count, 1 = write; bid / nb =
// the block's index and the number of blocks of THIS compaction, which need not be the launch's grid) before
// the first item and followed by a barrier: per-block constants go to LDS there, and small side jobs ride along,
// instead of costing a preparation launch
template <class P, class = void>
struct pred_has_block_init : std::false_type {};
template <class P>
struct pred_has_block_init<P, std::void_t<decltype(&P::block_init)>> : std::true_type {};

// a Pred may split itself into fetch(i) -> Item (the loads; i is always a valid index) and test(item, i): the kernels then
// FETCH all of a thread's items before testing any, so that the memory round trips of its kCI items overlap instead of
// running back to back (a predicate with an early-out before its load serialises them: 8 dependent trips per thread for
// the projection).  A Writer may then provide put(item, i, pos) to reuse what the predicate fetched.
template <class P, class = void>
struct pred_has_fetch : std::false_type {};
template <class P>
struct pred_has_fetch<P, std::void_t<typename P::Item>> : std::true_type {};
template <class W, class = void>
struct writer_has_put : std::false_type {};
template <class W>
struct writer_has_put<W, std::void_t<decltype(&W::put)>> : std::true_type {};

constexpr int kCT = 256;               // threads per block
constexpr int kCI = 4;                 // consecutive items per thread
constexpr int kCB = kCT * kCI;         // items per block

static inline int compact_blocks(int64_t n) { return n > 0 ? cdiv(n, kCB) : 1; }
// workspace: block_counts[nb] + block_offsets[nb]
static inline size_t compact_ws_bytes(int64_t n) { return align_up(2 * sizeof(int) * (size_t)compact_blocks(n), 256); }

// `flags` (optional, one byte per thread = its kCI verdicts): the write pass then reads the verdicts instead of evaluating the
// predicate again -- for predicates that cost more to evaluate than a byte costs to move (the projection: two IEEE
// divisions per point; the write pass of a 1.7 M-point map took 18 us re-projecting, 12 B per point re-read).
// (bodies as device functions of a VIRTUAL block index / block count: compact_count2_k / compact_write2_k below run two
// independent compactions in one launch each, the second behind the first in the grid)
template <class Pred>
__device__ __forceinline__ void compact_count_body(int64_t n, const Pred &pred, int *__restrict__ block_counts,
                                                   unsigned char *__restrict__ flags, int bid, int nb) {
    __shared__ int sm[kCT / 64];
    const int64_t base = (int64_t)bid * kCB + (int64_t)threadIdx.x * kCI;
    int c = 0;
    if constexpr (pred_has_fetch<Pred>::value) {
        typename Pred::Item it[kCI];  // (fetched before the block's preparation: its loads overlap these)
#pragma unroll
        for (int k = 0; k < kCI; ++k) it[k] = pred.fetch(base + k < n ? base + k : 0);
        if constexpr (pred_has_block_init<Pred>::value) {
            pred.block_init(0, bid, nb);
            __syncthreads();
        }
        int bits = 0;
#pragma unroll
        for (int k = 0; k < kCI; ++k)
            if (base + k < n && pred.test(it[k], base + k)) { ++c; bits |= 1 << k; }
        if (flags) flags[(int64_t)bid * kCT + threadIdx.x] = (unsigned char)bits;
    } else {
        if constexpr (pred_has_block_init<Pred>::value) {
            pred.block_init(0, bid, nb);
            __syncthreads();
        }
        int bits = 0;
#pragma unroll
        for (int k = 0; k < kCI; ++k) {
            const int64_t i = base + k;
            if (i < n && pred(i)) { ++c; bits |= 1 << k; }
        }
        if (flags) flags[(int64_t)bid * kCT + threadIdx.x] = (unsigned char)bits;
    }
    c = wave_sum_i(c);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) {
        int s = 0;
        for (int w = 0; w < kCT / 64; ++w) s += sm[w];
        block_counts[bid] = s;
    }
}
template <class Pred>
__global__ __launch_bounds__(kCT) void compact_count_k(int64_t n, Pred pred, int *__restrict__ block_counts,
                                                       unsigned char *__restrict__ flags = nullptr) {
    compact_count_body(n, pred, block_counts, flags, (int)blockIdx.x, (int)gridDim.x);
}

// single block; out_total[0] = sum, optionally added to *accum_base first (for segmented use)
static __global__ __launch_bounds__(1024) void compact_scan_k(const int *__restrict__ block_counts, int nblocks,
                                                       int *__restrict__ block_offsets, int *__restrict__ out_total) {
    __shared__ int sm[1024 / 64 + 1];
    __shared__ int carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int start = 0; start < nblocks; start += 1024) {
        const int i = start + threadIdx.x;
        const int v = i < nblocks ? block_counts[i] : 0;
        int total;
        const int ex = block_excl_scan<1024>(v, sm, &total);
        if (i < nblocks) block_offsets[i] = carry + ex;
        __syncthreads();
        if (threadIdx.x == 0) carry += total;
        __syncthreads();
    }
    if (threadIdx.x == 0 && out_total) out_total[0] = carry;
}

constexpr int kSelfScanBlocks = 1024;

// sum of counts[0 .. blockIdx.x) by the whole block; the last block also publishes the grand total
__device__ __forceinline__ int self_scan_offset(const int *__restrict__ block_counts, int *__restrict__ out_total, int bid, int nb) {
    __shared__ int red[kCT / 64];
    __shared__ int result;
    int mine = 0, all = 0;
    const bool last = bid == nb - 1;
    for (int j = threadIdx.x; j < nb; j += kCT) {
        const int c = (j < bid || last) ? block_counts[j] : 0;
        if (j < bid) mine += c;
        all += c;
    }
    mine = wave_sum_i(mine);
    all = wave_sum_i(all);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mine;
    __syncthreads();
    if (threadIdx.x == 0) {
        int s = 0;
        for (int w = 0; w < kCT / 64; ++w) s += red[w];
        result = s;
    }
    __syncthreads();
    const int off = result;
    if (last && out_total) {  // block-uniform branch
        __syncthreads();
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = all;
        __syncthreads();
        if (threadIdx.x == 0) {
            int s = 0;
            for (int w = 0; w < kCT / 64; ++w) s += red[w];
            out_total[0] = s;
        }
    }
    return off;
}

template <class Pred, class Writer, bool SelfScan>
__device__ __forceinline__ void compact_write_body(int64_t n, const Pred &pred, const Writer &writer,
                                                   const int *__restrict__ block_offsets /* SelfScan: block COUNTS */,
                                                   int *__restrict__ out_total, const unsigned char *__restrict__ flags, int bid, int nb) {
    __shared__ int sm[kCT / 64 + 1];
    const int my_bits = flags ? flags[(int64_t)bid * kCT + threadIdx.x] : 0;  // (requested before everything else)
    if constexpr (pred_has_block_init<Pred>::value) {
        pred.block_init(1, bid, nb);
        __syncthreads();
    }
    const int block_base = SelfScan ? self_scan_offset(block_offsets, out_total, bid, nb) : block_offsets[bid];
    const int64_t base = (int64_t)bid * kCB + (int64_t)threadIdx.x * kCI;
    bool f[kCI];
    int c = 0;
    if (flags) {  // the count pass's verdicts: no predicate, no fetch; the writer loads what it needs for the rows it writes
#pragma unroll
        for (int k = 0; k < kCI; ++k) { f[k] = (my_bits >> k) & 1; c += f[k] ? 1 : 0; }
        int total;
        int pos = block_base + block_excl_scan<kCT>(c, sm, &total);
#pragma unroll
        for (int k = 0; k < kCI; ++k) {
            if (f[k]) {
                writer(base + k, (int64_t)pos);
                ++pos;
            } else if constexpr (writer_has_skip<Writer>::value) {
                if (base + k < n) writer.skip(base + k);
            }
        }
        return;
    }
    if constexpr (pred_has_fetch<Pred>::value) {
        typename Pred::Item it[kCI];
#pragma unroll
        for (int k = 0; k < kCI; ++k) it[k] = pred.fetch(base + k < n ? base + k : 0);
#pragma unroll
        for (int k = 0; k < kCI; ++k) {
            f[k] = (base + k < n) && pred.test(it[k], base + k);
            c += f[k] ? 1 : 0;
        }
        int total;
        int pos = block_base + block_excl_scan<kCT>(c, sm, &total);
#pragma unroll
        for (int k = 0; k < kCI; ++k) {
            if (f[k]) {
                if constexpr (writer_has_put<Writer>::value) writer.put(it[k], base + k, (int64_t)pos);
                else writer(base + k, (int64_t)pos);
                ++pos;
            } else if constexpr (writer_has_skip<Writer>::value) {
                if (base + k < n) writer.skip(base + k);
            }
        }
        return;
    }
#pragma unroll
    for (int k = 0; k < kCI; ++k) {
        const int64_t i = base + k;
        f[k] = (i < n) && pred(i);
        c += f[k] ? 1 : 0;
    }
    int total;
    int pos = block_base + block_excl_scan<kCT>(c, sm, &total);
#pragma unroll
    for (int k = 0; k < kCI; ++k) {
        if (f[k]) {
            writer(base + k, (int64_t)pos);
            ++pos;
        } else if constexpr (writer_has_skip<Writer>::value) {
            if (base + k < n) writer.skip(base + k);
        }
    }
}
template <class Pred, class Writer, bool SelfScan = false>
__global__ __launch_bounds__(kCT) void compact_write_k(int64_t n, Pred pred, Writer writer,
                                                       const int *__restrict__ block_offsets /* SelfScan: block COUNTS */,
                                                       int *__restrict__ out_total = nullptr,
                                                       const unsigned char *__restrict__ flags = nullptr) {
    compact_write_body<Pred, Writer, SelfScan>(n, pred, writer, block_offsets, out_total, flags, (int)blockIdx.x, (int)gridDim.x);
}

// Two independent compactions in ONE launch per pass: blocks [0, nbA) run A, blocks [nbA, nbA + nbB) run B.  For a small job
// whose inputs are ready at the same time as a large one's (the ds-grid source cloud of a frame next to the projection of
// the map): its two launches (~5 us each, latency) disappear into the large job's.
template <class PA, class PB>
__global__ __launch_bounds__(kCT) void compact_count2_k(int64_t nA, PA pa, int *__restrict__ countsA, unsigned char *__restrict__ flagsA, int nbA,
                                                        int64_t nB, PB pb, int *__restrict__ countsB, int nbB) {
    if ((int)blockIdx.x < nbA) compact_count_body(nA, pa, countsA, flagsA, (int)blockIdx.x, nbA);
    else compact_count_body(nB, pb, countsB, (unsigned char *)nullptr, (int)blockIdx.x - nbA, nbB);
}
template <class PA, class WA, bool SelfScanA, class PB, class WB>
__global__ __launch_bounds__(kCT) void compact_write2_k(int64_t nA, PA pa, WA wa, const int *__restrict__ offsetsA, int *__restrict__ totalA,
                                                        const unsigned char *__restrict__ flagsA, int nbA, int64_t nB, PB pb, WB wb,
                                                        const int *__restrict__ countsB, int *__restrict__ totalB, int nbB) {
    if ((int)blockIdx.x < nbA) compact_write_body<PA, WA, SelfScanA>(nA, pa, wa, offsetsA, totalA, flagsA, (int)blockIdx.x, nbA);
    else compact_write_body<PB, WB, true>(nB, pb, wb, countsB, totalB, (const unsigned char *)nullptr, (int)blockIdx.x - nbA, nbB);
}

// Enqueue the launches.  ws must hold compact_ws_bytes(n).
// (A one-block, one-launch variant for small inputs was measured and rejected: 19 200 candidates on a single CU
// take ~60 us of dependent gathers -- two launches spread over 19 blocks take ~9.)
static inline size_t compact_flags_bytes(int64_t n) { return align_up((size_t)compact_blocks(n) * kCT, 256); }  // optional verdict bytes

template <class Pred, class Writer>
static inline int compact_launch(int64_t n, Pred pred, Writer writer, int *d_out_count, void *ws,
                                 hipStream_t st, const char *name, unsigned char *flags = nullptr /* compact_flags_bytes(n), or NULL */) {
    const int nb = compact_blocks(n);
    int *counts = (int *)ws;
    int *offsets = counts + nb;
    hipLaunchKernelGGL((compact_count_k<Pred>), dim3(nb), dim3(kCT), 0, st, n, pred, counts, flags);
    GS_LAUNCH_CHECK(name);
    if (nb <= kSelfScanBlocks) {
        hipLaunchKernelGGL((compact_write_k<Pred, Writer, true>), dim3(nb), dim3(kCT), 0, st, n, pred, writer, counts, d_out_count,
                           (const unsigned char *)flags);
        GS_LAUNCH_CHECK(name);
        return GS_OK;
    }
    hipLaunchKernelGGL(compact_scan_k, dim3(1), dim3(1024), 0, st, counts, nb, offsets, d_out_count);
    GS_LAUNCH_CHECK(name);
    hipLaunchKernelGGL((compact_write_k<Pred, Writer, false>), dim3(nb), dim3(kCT), 0, st, n, pred, writer, offsets,
                       (int *)nullptr, (const unsigned char *)flags);
    GS_LAUNCH_CHECK(name);
    return GS_OK;
}

// compact_launch for job A (with optional verdict bytes) and a small job B (at most kSelfScanBlocks blocks) side by side;
// wsA / wsB must hold compact_ws_bytes(nA) / compact_ws_bytes(nB)
template <class PA, class WA, class PB, class WB>
static inline int compact_launch2(int64_t nA, PA pa, WA wa, int *d_countA, void *wsA, unsigned char *flagsA, int64_t nB, PB pb, WB wb,
                                  int *d_countB, void *wsB, hipStream_t st, const char *name) {
    const int nbA = compact_blocks(nA), nbB = compact_blocks(nB);
    if (nbB > kSelfScanBlocks) {
        set_error("%s: second compaction too large for the fused form", name);
        return GS_ERR_INVALID_ARG;
    }
    int *countsA = (int *)wsA, *offsetsA = countsA + nbA, *countsB = (int *)wsB;
    hipLaunchKernelGGL((compact_count2_k<PA, PB>), dim3(nbA + nbB), dim3(kCT), 0, st, nA, pa, countsA, flagsA, nbA, nB, pb, countsB, nbB);
    GS_LAUNCH_CHECK(name);
    if (nbA <= kSelfScanBlocks) {
        hipLaunchKernelGGL((compact_write2_k<PA, WA, true, PB, WB>), dim3(nbA + nbB), dim3(kCT), 0, st, nA, pa, wa, (const int *)countsA, d_countA,
                           (const unsigned char *)flagsA, nbA, nB, pb, wb, (const int *)countsB, d_countB, nbB);
    } else {
        hipLaunchKernelGGL(compact_scan_k, dim3(1), dim3(1024), 0, st, countsA, nbA, offsetsA, d_countA);
        GS_LAUNCH_CHECK(name);
        hipLaunchKernelGGL((compact_write2_k<PA, WA, false, PB, WB>), dim3(nbA + nbB), dim3(kCT), 0, st, nA, pa, wa, (const int *)offsetsA,
                           (int *)nullptr, (const unsigned char *)flagsA, nbA, nB, pb, wb, (const int *)countsB, d_countB, nbB);
    }
    GS_LAUNCH_CHECK(name);
    return GS_OK;
}

// MultiWriter that appends: selected rows go behind the *base rows a destination already holds (device-side
// count), rows that would not fit into `cap` are dropped (the caller sizes cap so that this never happens and
// checks the overflow flag)
// one row of `w` 32-bit words from src row i to dst row j; the common widths are spelled out so that the three
// words of an xyz / rgb row travel as one dwordx3 access instead of three dword accesses
__device__ __forceinline__ void copy_row(const uint32_t *__restrict__ src, uint32_t *__restrict__ dst, int64_t i, int64_t j, int w) {
    if (w == 3) {
        const uint32_t a = src[3 * i], b = src[3 * i + 1], c = src[3 * i + 2];
        dst[3 * j] = a; dst[3 * j + 1] = b; dst[3 * j + 2] = c;
    } else if (w == 1) {
        dst[j] = src[i];
    } else {
        for (int k = 0; k < w; ++k) dst[j * w + k] = src[i * w + k];
    }
}

struct AppendWriter {
    const uint32_t *src[4];
    uint32_t *out[4];
    int words[4];
    int n_arrays;
    const int32_t *base;
    int cap;
    __device__ void operator()(int64_t i, int64_t pos) const {
        const int64_t at = (int64_t)(*base) + pos;
        if (at >= cap) return;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            if (a >= n_arrays) break;
            copy_row(src[a], out[a], i, at, words[a]);
        }
    }
};

static __global__ void append_count_k(int32_t *__restrict__ count, const int *__restrict__ total, int cap,
                               int32_t *__restrict__ appended, int32_t *__restrict__ overflow) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        const int64_t want = (int64_t)*count + *total;
        const int now = (int)(want > cap ? cap : want);
        if (appended) *appended = now - *count;
        if (overflow && want > cap) *overflow = 1;
        *count = now;
    }
}

}  // namespace gs
