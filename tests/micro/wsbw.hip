// Development microbenchmark: streaming bandwidth as a function of the working-set size (does a working set that
// fits the 256 MB Infinity Cache stream faster than one that does not?).  read = 16 B/lane loads reduced to one
// store per block; copy = float4 copy (half the working set read, half written).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ __launch_bounds__(256) void k_read(const float4 *__restrict__ in, float *__restrict__ out, size_t n) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t s = (size_t)gridDim.x * 256;
    float acc = 0.f;
    for (; i + 3 * s < n; i += 4 * s) {
        const float4 a = in[i], b = in[i + s], c = in[i + 2 * s], d = in[i + 3 * s];
        acc += a.x + b.y + c.z + d.w;
    }
    for (; i < n; i += s) acc += in[i].x;
    if (acc == 12345.678f) out[blockIdx.x] = acc;
}
__global__ __launch_bounds__(256) void k_copy(const float4 *__restrict__ in, float4 *__restrict__ out, size_t n) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t s = (size_t)gridDim.x * 256;
    for (; i < n; i += s) out[i] = in[i];
}

int main() {
    const size_t maxb = (size_t)2048 << 20;
    float4 *buf; float *o;
    CK(hipMalloc(&buf, maxb)); CK(hipMalloc(&o, 1 << 20));
    CK(hipMemset(buf, 0, maxb));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int grids[2] = {2048, 8192};
    for (int g = 0; g < 2; g++)
    for (size_t mb = 4; mb <= 2048; mb *= 2) {
        const size_t bytes = mb << 20, n = bytes / 16;
        const int reps = (int)((size_t)16384 / mb) + 2;
        float ms;
        hipLaunchKernelGGL(k_read, dim3(grids[g]), dim3(256), 0, 0, buf, o, n);
        CK(hipEventRecord(e0));
        for (int it = 0; it < reps; it++) hipLaunchKernelGGL(k_read, dim3(grids[g]), dim3(256), 0, 0, buf, o, n);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        const double rd = (double)bytes * reps / (ms * 1e-3) / 1e12;
        hipLaunchKernelGGL(k_copy, dim3(grids[g]), dim3(256), 0, 0, buf, buf + n / 2, n / 2);
        CK(hipEventRecord(e0));
        for (int it = 0; it < reps; it++) hipLaunchKernelGGL(k_copy, dim3(grids[g]), dim3(256), 0, 0, buf, buf + n / 2, n / 2);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        const double cp = (double)bytes * reps / (ms * 1e-3) / 1e12;
        printf("grid %5d  working set %5zu MB: read %.2f TB/s (%.1f us/launch)   copy %.2f TB/s\n", grids[g], mb, rd, bytes / rd / 1e6, cp);
    }
    return 0;
}
