// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the access widths the refine kernels use
// (MI355X_MICROARCH.md, HBM: "FETCH_SIZE reports exactly 1/2 of the bytes of a wide coalesced streaming read (16 B/lane) ...
// other access widths and WRITE_SIZE are uncalibrated: calibrate on a known byte count in your own access pattern").
// Every kernel streams a KNOWN number of bytes once through arrays far larger than the 256 MB Infinity Cache:
//   k_read<T>   T = 16, 8, 4, 2 bytes per lane, consecutive lanes consecutive elements (a wave reads 64 * T contiguous bytes)
//   k_write<T>  the same as stores
//   k_rows      the time-skewed refine kernel's staging pattern: per 64-pixel row segment one 8-byte load from each of five arrays
//               and one 2-byte load from each of two (44 B per pixel), one 8-byte store
// tests/tools/gpu_pmc_calib.sh runs it under `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` and writes
// profiles/pmc_calibration.json = known bytes / reported bytes per kernel.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <typename T> struct Acc { static __device__ unsigned f(T v) { return (unsigned)v; } };
template <> struct Acc<float4> { static __device__ unsigned f(float4 v) { return __float_as_uint(v.x) ^ __float_as_uint(v.y) ^ __float_as_uint(v.z) ^ __float_as_uint(v.w); } };
template <> struct Acc<double> { static __device__ unsigned f(double v) { return (unsigned)__double2loint(v) ^ (unsigned)__double2hiint(v); } };

template <typename T>
__global__ __launch_bounds__(256) void k_read(const T *__restrict__ in, size_t n, unsigned *__restrict__ sink) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t s = (size_t)gridDim.x * 256;
    unsigned a = 0;
    for (; i < n; i += s) a ^= Acc<T>::f(in[i]);
    if ((a & 0xffffu) == 0x5678u) sink[0] = a; // never true for the 0x11 test pattern (and not decidable at compile time): no write traffic
}
template <typename T>
__global__ __launch_bounds__(256) void k_write(T *__restrict__ out, size_t n, T v) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t s = (size_t)gridDim.x * 256;
    for (; i < n; i += s) out[i] = v;
}
__global__ __launch_bounds__(256) void k_rows(const double *__restrict__ a0, const double *__restrict__ a1, const double *__restrict__ a2,
                                              const double *__restrict__ a3, const double *__restrict__ a4, const uint16_t *__restrict__ k0,
                                              const uint16_t *__restrict__ k1, double *__restrict__ out, int W, int H) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int rows = (H + gridDim.y * 4 - 1) / (gridDim.y * 4);
    const int y0 = (blockIdx.y * 4 + (threadIdx.x >> 6)) * rows;
    if (x >= W) return;
    for (int y = y0; y < y0 + rows && y < H; y++) {
        const size_t p = (size_t)y * W + x;
        out[p] = a0[p] + a1[p] + a2[p] + a3[p] + a4[p] + (double)k0[p] + (double)k1[p];
    }
}

int main() {
    const size_t BYTES = (size_t)1 << 30; // 1 GiB per streamed array
    char *buf;
    unsigned *sink;
    CK(hipMalloc(&buf, BYTES));
    CK(hipMalloc(&sink, 64));
    CK(hipMemset(buf, 0x11, BYTES));
    CK(hipDeviceSynchronize());
    const dim3 g(16384), b(256);
    for (int rep = 0; rep < 3; rep++) {
        hipLaunchKernelGGL(k_read<float4>, g, b, 0, 0, (const float4 *)buf, BYTES / 16, sink);
        hipLaunchKernelGGL(k_read<double>, g, b, 0, 0, (const double *)buf, BYTES / 8, sink);
        hipLaunchKernelGGL(k_read<uint32_t>, g, b, 0, 0, (const uint32_t *)buf, BYTES / 4, sink);
        hipLaunchKernelGGL(k_read<uint16_t>, g, b, 0, 0, (const uint16_t *)buf, BYTES / 2, sink);
        hipLaunchKernelGGL(k_write<float4>, g, b, 0, 0, (float4 *)buf, BYTES / 16, make_float4(1, 2, 3, 4));
        hipLaunchKernelGGL(k_write<double>, g, b, 0, 0, (double *)buf, BYTES / 8, 1.5);
        hipLaunchKernelGGL(k_write<uint32_t>, g, b, 0, 0, (uint32_t *)buf, BYTES / 4, 7u);
        hipLaunchKernelGGL(k_write<uint16_t>, g, b, 0, 0, (uint16_t *)buf, BYTES / 2, (uint16_t)7);
        CK(hipDeviceSynchronize());
    }
    // the staging pattern: 4096 x 6144 pixels (both directions of a C2 top level), seven input arrays, one output
    const int W = 4096, H = 6144;
    const size_t px = (size_t)W * H;
    double *a[5], *out;
    uint16_t *k[2];
    for (int i = 0; i < 5; i++) { CK(hipMalloc(&a[i], px * 8)); CK(hipMemset(a[i], 0, px * 8)); }
    for (int i = 0; i < 2; i++) { CK(hipMalloc(&k[i], px * 2)); CK(hipMemset(k[i], 0, px * 2)); }
    CK(hipMalloc(&out, px * 8));
    CK(hipDeviceSynchronize());
    for (int rep = 0; rep < 3; rep++) {
        hipLaunchKernelGGL(k_rows, dim3(W / 64, 20), dim3(256), 0, 0, a[0], a[1], a[2], a[3], a[4], k[0], k[1], out, W, H);
        CK(hipDeviceSynchronize());
    }
    printf("known bytes: k_read* %zu read, k_write* %zu written, k_rows %zu read %zu written\n", BYTES, BYTES, px * 44, px * 8);
    return 0;
}
