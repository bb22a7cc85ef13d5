// Micro-benchmark: do two 1024-thread workgroups share a CU at a given LDS size / register count?
// 283 workgroups (more than the 256 CUs) each busy-wait ~20 us; if the 27 extra ones co-reside the launch takes ~20 us,
// if they have to wait for a CU it takes ~40 us.   hipcc --offload-arch=gfx950 -O3 coresidency.hip -o coresidency
#include <hip/hip_runtime.h>
#include <stdio.h>
extern __shared__ char dyn[];
template <int NV>
__global__ __launch_bounds__(1024) void spin_k(unsigned long long *out, int ticks, float seed) {
    float v[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = seed + i;
    const unsigned long long t0 = wall_clock64();
    dyn[threadIdx.x] = (char)threadIdx.x;
    while (wall_clock64() - t0 < (unsigned long long)ticks) {
#pragma unroll
        for (int i = 0; i < NV; ++i) v[i] = v[i] * 1.0001f + 0.5f;
        __builtin_amdgcn_s_sleep(8);
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < NV; ++i) s += v[i];
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = t0; out[2 * blockIdx.x + 1] = (unsigned long long)s + dyn[5]; }
}

// many live scalar registers: 46 pointer arguments read only after the spin
__global__ __launch_bounds__(1024) void spin_sgpr_k(unsigned long long *out, int ticks, const float *p0, const float *p1, const float *p2, const float *p3, const float *p4, const float *p5, const float *p6, const float *p7, const float *p8, const float *p9, const float *p10, const float *p11, const float *p12, const float *p13, const float *p14, const float *p15, const float *p16, const float *p17, const float *p18, const float *p19, const float *p20, const float *p21, const float *p22, const float *p23, const float *p24, const float *p25, const float *p26, const float *p27, const float *p28, const float *p29, const float *p30, const float *p31, const float *p32, const float *p33, const float *p34, const float *p35, const float *p36, const float *p37, const float *p38, const float *p39, const float *p40, const float *p41, const float *p42, const float *p43, const float *p44, const float *p45) {
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < (unsigned long long)ticks) __builtin_amdgcn_s_sleep(8);
    const float s = p0[blockIdx.x & 1] + p1[blockIdx.x & 1] + p2[blockIdx.x & 1] + p3[blockIdx.x & 1] + p4[blockIdx.x & 1] + p5[blockIdx.x & 1] + p6[blockIdx.x & 1] + p7[blockIdx.x & 1] + p8[blockIdx.x & 1] + p9[blockIdx.x & 1] + p10[blockIdx.x & 1] + p11[blockIdx.x & 1] + p12[blockIdx.x & 1] + p13[blockIdx.x & 1] + p14[blockIdx.x & 1] + p15[blockIdx.x & 1] + p16[blockIdx.x & 1] + p17[blockIdx.x & 1] + p18[blockIdx.x & 1] + p19[blockIdx.x & 1] + p20[blockIdx.x & 1] + p21[blockIdx.x & 1] + p22[blockIdx.x & 1] + p23[blockIdx.x & 1] + p24[blockIdx.x & 1] + p25[blockIdx.x & 1] + p26[blockIdx.x & 1] + p27[blockIdx.x & 1] + p28[blockIdx.x & 1] + p29[blockIdx.x & 1] + p30[blockIdx.x & 1] + p31[blockIdx.x & 1] + p32[blockIdx.x & 1] + p33[blockIdx.x & 1] + p34[blockIdx.x & 1] + p35[blockIdx.x & 1] + p36[blockIdx.x & 1] + p37[blockIdx.x & 1] + p38[blockIdx.x & 1] + p39[blockIdx.x & 1] + p40[blockIdx.x & 1] + p41[blockIdx.x & 1] + p42[blockIdx.x & 1] + p43[blockIdx.x & 1] + p44[blockIdx.x & 1] + p45[blockIdx.x & 1];
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = t0; out[2 * blockIdx.x + 1] = (unsigned long long)s; }
}
template <int NV>
void run(int lds, unsigned long long *d, unsigned long long *h, int nblk) {
    hipFuncSetAttribute((const void *)spin_k<NV>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(a);
        hipLaunchKernelGGL(spin_k<NV>, dim3(nblk), dim3(1024), lds, 0, d, 2000 /* 20 us at 100 MHz */, 1.0f);
        hipEventRecord(b);
        hipEventSynchronize(b);
    }
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    hipMemcpy(h, d, nblk * 16, hipMemcpyDeviceToHost);
    unsigned long long t0 = ~0ull, late = 0;
    for (int i = 0; i < nblk; ++i) t0 = h[2 * i] < t0 ? h[2 * i] : t0;
    for (int i = 0; i < nblk; ++i) late += (h[2 * i] - t0) > 500;  // started more than 5 us after the first
    printf("accumulators %3d  LDS %6d B : launch %.1f us, %llu of %d workgroups started > 5 us late\n", NV, lds, ms * 1e3, late, nblk);
}
int main() {
    const int nblk = 283;
    unsigned long long *d, *h = (unsigned long long *)malloc(nblk * 16);
    hipMalloc(&d, nblk * 16);
    for (int lds : {8192, 26624, 32768, 40960, 49152, 57344, 65536, 69632, 73728, 81920}) run<8>(lds, d, h, nblk);
    for (int lds : {8192, 57344}) run<40>(lds, d, h, nblk);
    // register count: the spin loop keeps NV accumulators live (the build prints the VGPR count of each instance)
    run<30>(8192, d, h, nblk); run<36>(8192, d, h, nblk); run<42>(8192, d, h, nblk); run<46>(8192, d, h, nblk);
    run<50>(8192, d, h, nblk); run<54>(8192, d, h, nblk); run<58>(8192, d, h, nblk);

    {
        hipEvent_t a, b;
        hipEventCreate(&a); hipEventCreate(&b);
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(a);
            hipLaunchKernelGGL(spin_sgpr_k, dim3(nblk), dim3(1024), 0, 0, d, 2000, (const float *)d, (const float *)d, (const float *)d, (const float *)d, (const float *)d, (const float *)d, (const float *)d, (const float *)d, (const float *)d, (const float *)d, (const float *)d, (const float *)d, (const float *)d, (const float *)d, (const float *)d, (const float *)d, (const float *)d, (const float *)d, (const float *)d, (const float *)d, (const float *)d, (const float *)d, (const float *)d, (const float *)d, (const float *)d, (const float *)d, (const float *)d, (const float *)d, (const float *)d, (const float *)d, (const float *)d, (const float *)d, (const float *)d, (const float *)d, (const float *)d, (const float *)d, (const float *)d, (const float *)d, (const float *)d, (const float *)d, (const float *)d, (const float *)d, (const float *)d, (const float *)d, (const float *)d, (const float *)d);
            hipEventRecord(b);
            hipEventSynchronize(b);
        }
        float ms = 0;
        hipEventElapsedTime(&ms, a, b);
        hipMemcpy(h, d, nblk * 16, hipMemcpyDeviceToHost);
        unsigned long long t0 = ~0ull, late = 0;
        for (int i = 0; i < nblk; ++i) t0 = h[2 * i] < t0 ? h[2 * i] : t0;
        for (int i = 0; i < nblk; ++i) late += (h[2 * i] - t0) > 500;
        printf("46 live pointer arguments (see the build's SGPR count): launch %.1f us, %llu of %d workgroups started > 5 us late\n", ms * 1e3, late, nblk);
    }
    return 0;
}
