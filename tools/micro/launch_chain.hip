// launch_chain.hip -- what does ONE dependent launch cost on a stream, and what changes it?
//
// Question behind it ("tile size" / section 8): the same 200-frame PointFusion run is 8-13 % faster
// under `rocprofv3 --kernel-trace` than clean -- ~2.8 us per launch that is neither kernel time nor a gap in the trace --
// and it is not the hipGraph replay (measured slower).  The profiler rewrites every AQL dispatch packet (its own
// completion signal, profiling timestamps).  This program times a chain of N dependent launches of a kernel shaped
// like the association launch (G blocks x 1024 threads, each spinning for `us` microseconds) in the forms the library
// could use itself:
//   plain    hipLaunchKernelGGL back to back
//   events   hipExtLaunchKernelGGL with a start / stop event pair on every launch (a completion signal per dispatch:
//            what the profiler's packets carry)
//   record   hipLaunchKernelGGL + hipEventRecord after every launch (a barrier packet with a signal behind every dispatch)
//   graph    the chain captured once, replayed as a hipGraph
//   two      launches alternating between two streams, ordered by events (the dependency as an explicit wait)
// Output: microseconds per launch for each form (host clock around the whole chain, queue drained before and after),
// and the sum of the kernels' own durations where events give it.
//
// Build: hipcc --offload-arch=gfx950 -O3 launch_chain.hip -o launch_chain ; run: ./launch_chain [G=300] [us=12] [N=220]
// NOT yet run on the GPU box (written after the round's GPU minutes were spent).
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                          \
    do {                                                                               \
        hipError_t e_ = (x);                                                           \
        if (e_ != hipSuccess) {                                                        \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));  \
            exit(1);                                                                   \
        }                                                                              \
    } while (0)

// every block spins `ticks` of the 100 MHz wall clock (bounded: leaves after `ticks` whatever happens), then block 0
// bumps a counter the NEXT launch reads -- a real dependency through memory, like the loop's state
__global__ __launch_bounds__(1024) void spin_k(const int *__restrict__ in, int *__restrict__ out, long long ticks) {
    const long long t0 = wall_clock64();
    int v = *in;
    while (wall_clock64() - t0 < ticks) v += 1;
    if (blockIdx.x == 0 && threadIdx.x == 0) *out = (v & 1) + *in + 1;
}

static double now_us() {
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main(int argc, char **argv) {
    const int G = argc > 1 ? atoi(argv[1]) : 300, N = argc > 3 ? atoi(argv[3]) : 220;
    const double us = argc > 2 ? atof(argv[2]) : 12.0;
    const long long ticks = (long long)(us * 100.0);
    int *buf = nullptr;
    CK(hipMalloc(&buf, 2 * sizeof(int)));
    CK(hipMemset(buf, 0, 2 * sizeof(int)));
    hipStream_t st[2];
    CK(hipStreamCreateWithFlags(&st[0], hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&st[1], hipStreamNonBlocking));
    std::vector<hipEvent_t> ev(2 * N);
    for (auto &e : ev) CK(hipEventCreate(&e));
    auto launch = [&](hipStream_t s, int i) { hipLaunchKernelGGL(spin_k, dim3(G), dim3(1024), 0, s, buf + (i & 1), buf + 1 - (i & 1), ticks); };

    for (int warm = 0; warm < 2; ++warm) {
        const bool report = warm == 1;
        // plain
        CK(hipDeviceSynchronize());
        double t0 = now_us();
        for (int i = 0; i < N; ++i) launch(st[0], i);
        CK(hipStreamSynchronize(st[0]));
        if (report) printf("plain   %7.2f us per launch (kernel body %.1f us, %d blocks)\n", (now_us() - t0) / N, us, G);
        // events: completion signal on every dispatch
        CK(hipDeviceSynchronize());
        t0 = now_us();
        for (int i = 0; i < N; ++i)
            hipExtLaunchKernelGGL(spin_k, dim3(G), dim3(1024), 0, st[0], ev[2 * i], ev[2 * i + 1], 0, buf + (i & 1), buf + 1 - (i & 1), ticks);
        CK(hipStreamSynchronize(st[0]));
        if (report) {
            const double wall = (now_us() - t0) / N;
            double sum = 0.0;
            for (int i = 0; i < N; ++i) {
                float ms = 0.0f;
                CK(hipEventElapsedTime(&ms, ev[2 * i], ev[2 * i + 1]));
                sum += 1e3 * ms;
            }
            printf("events  %7.2f us per launch (sum of the kernels' own durations %.2f us per launch)\n", wall, sum / N);
        }
        // record: an event behind every launch
        CK(hipDeviceSynchronize());
        t0 = now_us();
        for (int i = 0; i < N; ++i) {
            launch(st[0], i);
            CK(hipEventRecord(ev[i], st[0]));
        }
        CK(hipStreamSynchronize(st[0]));
        if (report) printf("record  %7.2f us per launch\n", (now_us() - t0) / N);
        // graph
        {
            hipGraph_t g = nullptr;
            hipGraphExec_t ge = nullptr;
            CK(hipStreamBeginCapture(st[0], hipStreamCaptureModeThreadLocal));
            for (int i = 0; i < N; ++i) launch(st[0], i);
            CK(hipStreamEndCapture(st[0], &g));
            CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
            CK(hipGraphLaunch(ge, st[0]));  // first replay uploads
            CK(hipStreamSynchronize(st[0]));
            t0 = now_us();
            CK(hipGraphLaunch(ge, st[0]));
            CK(hipStreamSynchronize(st[0]));
            if (report) printf("graph   %7.2f us per launch\n", (now_us() - t0) / N);
            CK(hipGraphExecDestroy(ge));
            CK(hipGraphDestroy(g));
        }
        // two streams, dependency as an explicit event wait
        CK(hipDeviceSynchronize());
        t0 = now_us();
        for (int i = 0; i < N; ++i) {
            hipStream_t s = st[i & 1];
            if (i > 0) CK(hipStreamWaitEvent(s, ev[i - 1], 0));
            launch(s, i);
            CK(hipEventRecord(ev[i], s));
        }
        CK(hipStreamSynchronize(st[0]));
        CK(hipStreamSynchronize(st[1]));
        if (report) printf("two     %7.2f us per launch\n", (now_us() - t0) / N);
    }
    int h[2];
    CK(hipMemcpy(h, buf, sizeof(h), hipMemcpyDeviceToHost));
    printf("chain counter %d %d (every launch ran after the one before it)\n", h[0], h[1]);
    return 0;
}
