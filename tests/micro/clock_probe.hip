// Probe: what do s_memtime / s_memrealtime tick at on this device, and how many workgroups of a launch really run at once?
// Every workgroup spins through a fixed fp64 chain and records its start / end (s_memrealtime, 100 MHz, chip-wide).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
template <int LDS>
__global__ __launch_bounds__(256) void k(unsigned long long *o, int iters) {
    __shared__ double pad[LDS / 8 + 1];
    const unsigned long long m0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    double x = threadIdx.x * 1e-3 + 1.0;
    pad[threadIdx.x % (LDS / 8 + 1)] = x;
    __syncthreads();
    for (int i = 0; i < iters; i++) x = __builtin_fma(x, 1.0000001, pad[(i + threadIdx.x) % (LDS / 8 + 1)]);
    const unsigned long long m1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { o[4 * blockIdx.x] = r0; o[4 * blockIdx.x + 1] = r1; o[4 * blockIdx.x + 2] = m1 - m0; o[4 * blockIdx.x + 3] = (unsigned long long)x; }
}
template <int LDS>
void run(int blocks, unsigned long long *d) {
    std::vector<unsigned long long> h(4 * blocks);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<LDS>, dim3(blocks), dim3(256), 0, 0, d, 20000);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(h.data(), d, 32 * blocks, hipMemcpyDeviceToHost);
        unsigned long long s0 = ~0ull, s1 = 0, e_1 = 0; double life = 0; int late = 0;
        for (int b = 0; b < blocks; b++) { s0 = std::min(s0, h[4 * b]); s1 = std::max(s1, h[4 * b]); e_1 = std::max(e_1, h[4 * b + 1]); life += (h[4 * b + 1] - h[4 * b]) / 100.0; }
        for (int b = 0; b < blocks; b++) late += (h[4 * b] - s0) > 1000; // started more than 10 us after the first
        printf("LDS %6d B blocks %4d: event %.3f ms | first start -> last end %.3f ms, mean lifetime %.3f ms, start spread %.3f ms, %d blocks started late | memtime %.0f MHz\n", LDS, blocks, ms,
               (e_1 - s0) / 1e5, life / blocks / 1e3, (s1 - s0) / 1e5, late, h[2] / ((h[1] - h[0]) / 100.0));
    }
}
int main() {
    unsigned long long *d;
    hipMalloc(&d, 32 * 4096);
    run<64>(1280, d);
    run<30008>(1280, d); run<30712>(1280, d); run<31224>(1280, d); run<31736>(1280, d); run<31864>(1280, d); run<31992>(1280, d); run<32504>(1280, d); run<32760>(1280, d);
    return 0;
}
