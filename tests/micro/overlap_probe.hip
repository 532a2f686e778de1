// Probe (round 5, profiles/LAB_NOTES.md 4): do a wave's outstanding global loads make progress while the SIMD's vector unit issues fp64
// fma back to back?  Two waves per SIMD (256 VGPRs each would allow no more), every wave streams down its own 512-byte column
// of a row-major array, PF steps of prefetch into registers, and runs `fmas` dependent-chain fp64 fma (4 chains) per step.
//   loads only, fma only, both: if both ~ max(loads, fma) the two overlap; if ~ sum, they do not.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

template <int FMAS, int LOADS, int NARR>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) void k(const double *__restrict__ a, size_t plane, int W, int rows, double *o, double m) {
    const int lane = threadIdx.x;
    const int strip = blockIdx.x, chunk = blockIdx.y;
    const double *p = a + (size_t)chunk * rows * W + (size_t)strip * 56 + lane;
    double x0 = lane, x1 = 1, x2 = 2, x3 = 3, acc = 0;
    double g[2][NARR];
#pragma unroll
    for (int j = 0; j < NARR; j++) g[0][j] = g[1][j] = 0;
    for (int s = 0; s < rows; s += 2) {
#pragma unroll
        for (int ph = 0; ph < 2; ph++) {
            if (LOADS) {
#pragma unroll
                for (int j = 0; j < NARR; j++) g[ph][j] = p[(size_t)j * plane + (size_t)(s + ph) * W]; // consumed one step later
            }
#pragma unroll
            for (int u = 0; u < FMAS / 4; u++) {
                x0 = __builtin_fma(x0, m, 1e-9);
                x1 = __builtin_fma(x1, m, 1e-9);
                x2 = __builtin_fma(x2, m, 1e-9);
                x3 = __builtin_fma(x3, m, 1e-9);
            }
            if (LOADS) {
#pragma unroll
                for (int j = 0; j < NARR; j++) acc += g[ph ^ 1][j];
            }
        }
    }
    const double r = x0 + x1 + x2 + x3 + acc;
    if (r == 12345.678) o[lane] = r;
}

template <int FMAS, int LOADS, int NARR>
float run(const double *a, size_t plane, int W, int H, double *o) {
    const int strips = W / 56 - 1, chunks = 2048 / strips, rows = (H / chunks) & ~1;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<FMAS, LOADS, NARR>), dim3(strips, chunks), dim3(64), 0, 0, a, plane, W, rows, o, 1.0000001);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        best = std::min(best, ms);
    }
    const double bytes = LOADS ? (double)strips * chunks * rows * 512.0 * NARR : 0.0, fma = (double)strips * chunks * rows * 64.0 * FMAS;
    printf("fma/step %4d loads %d x %d arrays: %.3f ms   %.2f TB/s   %.1f Tfma/s (%d waves, %d rows each)\n", FMAS, LOADS, NARR, best, bytes / best / 1e9, fma / best / 1e9, strips * chunks, rows);
    return best;
}

int main() {
    const int W = 4096, H = 3072, NARR = 5;
    const size_t plane = (size_t)W * H;
    double *a, *o;
    hipMalloc(&a, plane * NARR * sizeof(double));
    hipMalloc(&o, 4096);
    hipMemset(a, 0, plane * NARR * sizeof(double));
    run<0, 1, NARR>(a, plane, W, H, o);
    run<400, 0, NARR>(a, plane, W, H, o);
    run<400, 1, NARR>(a, plane, W, H, o);
    run<800, 0, NARR>(a, plane, W, H, o);
    run<800, 1, NARR>(a, plane, W, H, o);
    run<1600, 0, NARR>(a, plane, W, H, o);
    run<1600, 1, NARR>(a, plane, W, H, o);
    return 0;
}
