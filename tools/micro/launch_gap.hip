// launch_gap.hip -- what makes a dependent launch cost more than the ~1.5 us launch_chain.hip measured?
//
// The 200-frame PointFusion run spends ~0.39 ms per frame in kernels (rocprofv3) but 0.61 ms of wall clock in a clean run,
// with the host ~4x ahead of the GPU (0.14 ms per frame to enqueue): ~7 us per kernel boundary that is neither kernel
// nor host.  launch_chain.hip's chain of identical tiny-footprint kernels shows 1.5 us.  This program varies what real
// kernels have and that chain had not: DIFFERENT kernels alternating, a large static LDS allocation, megabytes WRITTEN
// per kernel (dirty L2 lines to write back at the kernel boundary), block size, a large by-value argument.
// Every kernel spins for `us` microseconds of wall clock; reported: (chain time / launches) - us.
//
// Build: hipcc --offload-arch=gfx950 -O3 launch_gap.hip -o launch_gap ; run: ./launch_gap [us=10] [N=300]
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>

#define CK(x)                                                                          \
    do {                                                                               \
        hipError_t e_ = (x);                                                           \
        if (e_ != hipSuccess) {                                                        \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));  \
            exit(1);                                                                   \
        }                                                                              \
    } while (0)

struct Big {
    float v[96];
};

template <int LDS_WORDS, int VARIANT>
__global__ void work_k(const int *__restrict__ in, int *__restrict__ out, long long ticks, float *__restrict__ sink, long long wwords,
                       Big big) {
    __shared__ int lds[LDS_WORDS > 0 ? LDS_WORDS : 1];
    const long long t0 = wall_clock64();
    int v = *in + VARIANT;
    if (LDS_WORDS > 0) lds[threadIdx.x % LDS_WORDS] = v;
    // write wwords floats (spread over the grid): dirty lines the kernel boundary has to make visible
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x, gsz = (long long)gridDim.x * blockDim.x;
    for (long long i = gid; i < wwords; i += gsz) sink[i] = big.v[i % 96] + (float)v;
    while (wall_clock64() - t0 < ticks) v += 1;
    if (LDS_WORDS > 0) v += lds[(threadIdx.x + 1) % LDS_WORDS];
    if (blockIdx.x == 0 && threadIdx.x == 0) *out = (v & 1) + *in + 1;
}

static double now_us() {
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main(int argc, char **argv) {
    const double us = argc > 1 ? atof(argv[1]) : 10.0;
    const int N = argc > 2 ? atoi(argv[2]) : 300;
    const long long ticks = (long long)(us * 100.0);
    int *buf = nullptr;
    float *sink = nullptr;
    const long long max_words = 64ll << 20;  // 256 MB
    CK(hipMalloc(&buf, 2 * sizeof(int)));
    CK(hipMemset(buf, 0, 2 * sizeof(int)));
    CK(hipMalloc(&sink, max_words * 4));
    hipStream_t st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    Big big{};
    auto chain = [&](const char *label, auto launch) {
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipDeviceSynchronize());
            const double t0 = now_us();
            for (int i = 0; i < N; ++i) launch(i);
            CK(hipStreamSynchronize(st));
            if (rep == 1) printf("%-64s %6.2f us per launch beyond the %.0f us body\n", label, (now_us() - t0) / N - us, us);
        }
    };
#define L(KERNEL, G, T, W) hipLaunchKernelGGL(KERNEL, dim3(G), dim3(T), 0, st, buf + (i & 1), buf + 1 - (i & 1), ticks, sink, (long long)(W), big)
    chain("one kernel, 300 x 1024, no LDS, nothing written", [&](int i) { L((work_k<0, 0>), 300, 1024, 0); });
    chain("one kernel, 300 x 1024, 72 KB LDS, nothing written", [&](int i) { L((work_k<18432, 0>), 300, 1024, 0); });
    chain("two kernels alternating, 300 x 1024, 72 KB LDS, nothing written", [&](int i) { if (i & 1) L((work_k<18432, 1>), 300, 1024, 0); else L((work_k<18432, 0>), 300, 1024, 0); });
    chain("one kernel, 300 x 1024, no LDS, 1 MB written", [&](int i) { L((work_k<0, 0>), 300, 1024, 1 << 18); });
    chain("one kernel, 300 x 1024, no LDS, 16 MB written", [&](int i) { L((work_k<0, 0>), 300, 1024, 4 << 20); });
    chain("one kernel, 300 x 1024, no LDS, 128 MB written", [&](int i) { L((work_k<0, 0>), 300, 1024, 32 << 20); });
    chain("one kernel, 2048 x 256, no LDS, 16 MB written", [&](int i) { L((work_k<0, 2>), 2048, 256, 4 << 20); });
    chain("mix like a frame: 256-thread streaming (16 MB) then 11 x (1024, LDS, 1 MB)", [&](int i) {
        if (i % 12 == 0) L((work_k<0, 2>), 2048, 256, 4 << 20); else L((work_k<18432, 0>), 300, 1024, 1 << 18);
    });
    chain("one kernel, 1 x 64 (a tiny step kernel)", [&](int i) { L((work_k<0, 3>), 1, 64, 0); });
    int h[2];
    CK(hipMemcpy(h, buf, sizeof(h), hipMemcpyDeviceToHost));
    printf("chain counter %d %d\n", h[0], h[1]);
    return 0;
}
