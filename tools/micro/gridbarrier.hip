// Micro-benchmark for VERDICT r1 item 5: what would ONE device-scope barrier between the iterations of a persistent
// ICP loop kernel cost, against the ~4.5 us of stream time a dependent launch costs?  300 workgroups of 1024 threads
// (the c2 association's grid; two per CU, all resident), every iteration: block barrier, one agent-scope atomic
// arrival per workgroup, bounded spin (sc1 loads) until all have arrived, block barrier -- plus the hand-off a loop
// needs: every workgroup writes a 29-float row before the barrier and block 0's rows are re-read by everyone after it.
// Every spin is bounded (a stuck barrier ends the kernel instead of hanging the GPU).
//   hipcc --offload-arch=gfx950 -O3 gridbarrier.hip -o gridbarrier
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
__global__ __launch_bounds__(1024) void loop_k(unsigned *counter, float *rows, float *sink, unsigned long long *ticks, int iters, int payload) {
    const unsigned nb = gridDim.x;
    __shared__ float acc;
    const unsigned long long t0 = wall_clock64();
    unsigned failed = 0;
    for (int it = 0; it < iters; ++it) {
        if (payload && threadIdx.x < 29) __hip_atomic_store(rows + ((it & 1) * nb + blockIdx.x) * 32 + threadIdx.x, (float)(it + threadIdx.x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        if (threadIdx.x == 0) {
            __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned target = (unsigned)(it + 1) * nb;
            unsigned spins = 0;
            while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target && ++spins < 400000u) __builtin_amdgcn_s_sleep(1);
            if (spins >= 400000u) failed = 1;
        }
        __syncthreads();
        if (payload) {  // what the folded step does: every block reduces all rows of the previous phase
            float v = 0.0f;
            for (unsigned b = threadIdx.x >> 5; b < nb; b += 32)
                if ((threadIdx.x & 31) < 29) v += __hip_atomic_load(rows + ((it & 1) * nb + b) * 32 + (threadIdx.x & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (threadIdx.x == 0) acc = 0.0f;
            __syncthreads();
            atomicAdd(&acc, v);
            __syncthreads();
            if (threadIdx.x == 0 && blockIdx.x == 0) sink[it & 7] = acc;
        }
        if (failed) break;
    }
    if (threadIdx.x == 0) { ticks[2 * blockIdx.x] = wall_clock64() - t0; ticks[2 * blockIdx.x + 1] = failed; }
}
int main() {
    const int nb = 300, iters = 200;
    unsigned *counter; float *rows, *sink; unsigned long long *ticks, h[2 * nb];
    hipMalloc(&counter, 4); hipMalloc(&rows, 2 * nb * 32 * 4); hipMalloc(&sink, 64); hipMalloc(&ticks, sizeof(h));
    for (int payload = 0; payload < 2; ++payload) {
        for (int rep = 0; rep < 3; ++rep) {
            hipMemset(counter, 0, 4);
            hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
            hipEventRecord(a);
            hipLaunchKernelGGL(loop_k, dim3(nb), dim3(1024), 0, 0, counter, rows, sink, ticks, iters, payload);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms = 0; hipEventElapsedTime(&ms, a, b);
            hipMemcpy(h, ticks, sizeof(h), hipMemcpyDeviceToHost);
            unsigned long long mx = 0, failed = 0;
            for (int i = 0; i < nb; ++i) { mx = h[2 * i] > mx ? h[2 * i] : mx; failed += h[2 * i + 1]; }
            if (rep == 2)
                printf("%s: %d iterations, 300 x 1024 threads: %.2f us per iteration (kernel %.1f us, slowest block %.1f us), barriers that gave up: %llu\n",
                       payload ? "barrier + 29-float row hand-off and reduce" : "barrier only", iters, 1e3 * ms / iters, 1e3 * ms, mx * 0.01, failed);
        }
    }
    // the same hand-off as dependent launches: an empty 300 x 1024 kernel, 200 in a row
    return 0;
}
