// Probe for the one-wave-per-strip form of the time-skewed refine kernel (profiles/LAB_NOTES.md 4, round 5):
//  (a) fp64 fma issue rate of a SIMD as a function of resident waves x independent dependent-chains per wave,
//  (b) the lane shifts v_mov_b32_dpp wave_shr:1 / wave_shl:1 on gfx950 (which way they move, what the edge lanes get),
//  (c) how many 128-thread workgroups of 39 296 B of LDS (two strips + the exp table) a CU holds at once.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

template <int C>
__global__ __launch_bounds__(64) void k_chain(double *o, int iters, double m) {
    double x[C];
#pragma unroll
    for (int c = 0; c < C; c++) x[c] = threadIdx.x * 1e-3 + c;
    for (int i = 0; i < iters; i += 16) { // (16 rounds per trip: the loop's scalar instructions and branch stay out of the measurement)
#pragma unroll
        for (int u = 0; u < 16; u++)
#pragma unroll
            for (int c = 0; c < C; c++) x[c] = __builtin_fma(x[c], m, 1e-9);
    }
    double s = 0;
#pragma unroll
    for (int c = 0; c < C; c++) s += x[c];
    if (s == 12345.678) o[threadIdx.x] = s;
}

template <int C>
void run_chain(int waves_per_simd, double *d) {
    const int blocks = 256 * 4 * waves_per_simd; // one wave per workgroup
    const int iters = 4096 * 4 / C;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_chain<C>, dim3(blocks), dim3(64), 0, 0, d, iters, 1.0000001);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        best = std::min(best, ms);
    }
    const double fma = (double)blocks * 64.0 * iters * C;
    printf("waves/SIMD %d chains/wave %d: %.3f ms, %.1f Gfma/s = %.2f of 256 CU x 4 SIMD x 16 lanes x 2.4 GHz\n", waves_per_simd, C, best, fma / best / 1e6, fma / (best * 1e-3) / (256.0 * 4 * 16 * 2.4e9));
}

__global__ void k_dpp(const double *a, double *r, double *l) {
    const double v = a[threadIdx.x];
    int lo = __double2loint(v), hi = __double2hiint(v);
    const int rl = __builtin_amdgcn_update_dpp(-1, lo, 0x138, 0xf, 0xf, false), rh = __builtin_amdgcn_update_dpp(-1, hi, 0x138, 0xf, 0xf, false);
    const int ll = __builtin_amdgcn_update_dpp(-1, lo, 0x130, 0xf, 0xf, false), lh = __builtin_amdgcn_update_dpp(-1, hi, 0x130, 0xf, 0xf, false);
    r[threadIdx.x] = __hiloint2double(rh, rl);
    l[threadIdx.x] = __hiloint2double(lh, ll);
}

template <int LDS>
__global__ __launch_bounds__(128) void k_res(unsigned long long *o, int iters) {
    __shared__ double pad[LDS / 8];
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    double x = threadIdx.x * 1e-3 + 1.0;
    pad[threadIdx.x] = x;
    __syncthreads();
    for (int i = 0; i < iters; i++) x = __builtin_fma(x, 1.0000001, pad[(i + threadIdx.x) % (LDS / 8)]);
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) {
        o[2 * blockIdx.x] = r0;
        o[2 * blockIdx.x + 1] = r1 + (x == 1.5);
    }
}
template <int LDS>
void run_res(int blocks, unsigned long long *d) {
    std::vector<unsigned long long> h(2 * blocks);
    hipLaunchKernelGGL(k_res<LDS>, dim3(blocks), dim3(128), 0, 0, d, 20000);
    hipDeviceSynchronize();
    hipMemcpy(h.data(), d, 16 * blocks, hipMemcpyDeviceToHost);
    unsigned long long s0 = ~0ull;
    int late = 0;
    for (int b = 0; b < blocks; b++) s0 = std::min(s0, h[2 * b]);
    for (int b = 0; b < blocks; b++) late += (h[2 * b] - s0) > 1000;
    printf("LDS %6d B x %d workgroups of 128: %d started more than 10 us after the first\n", LDS, blocks, late);
}

int main() {
    double *d;
    hipMalloc(&d, 1 << 20);
    for (int w : {1, 2, 3, 4, 5, 8}) run_chain<1>(w, d);
    for (int w : {1, 2, 3}) run_chain<2>(w, d);
    for (int w : {1, 2, 3}) run_chain<4>(w, d);
    for (int w : {1, 2}) run_chain<8>(w, d);
    double h[64], r[64], l[64];
    for (int i = 0; i < 64; i++) h[i] = 100.0 + i;
    hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_dpp, dim3(1), dim3(64), 0, 0, d, d + 64, d + 128);
    hipMemcpy(r, d + 64, sizeof r, hipMemcpyDeviceToHost);
    hipMemcpy(l, d + 128, sizeof l, hipMemcpyDeviceToHost);
    printf("wave_shr:1 lanes 0 1 2 15 16 17 31 32 33 63: %g %g %g %g %g %g %g %g %g %g\n", r[0], r[1], r[2], r[15], r[16], r[17], r[31], r[32], r[33], r[63]);
    printf("wave_shl:1 lanes 0 1 2 15 16 17 31 32 33 62 63: %g %g %g %g %g %g %g %g %g %g %g\n", l[0], l[1], l[2], l[15], l[16], l[17], l[31], l[32], l[33], l[62], l[63]);
    run_res<38912>(1024, (unsigned long long *)d);
    run_res<39296>(1024, (unsigned long long *)d);
    run_res<39680>(1024, (unsigned long long *)d);
    run_res<40960>(1024, (unsigned long long *)d);
    return 0;
}
