// Development microbenchmark: memory-side cost of the refine light sweep's access pattern (no math).
//   v0: 1 pixel per lane and row, 8-byte loads, entry array chosen per pixel (as the shipped kernel)
//   v1: 2 adjacent pixels per lane, 16-byte loads, BOTH entry arrays always read, choice in registers
//   v2: plain float4 copy of the same number of bytes (reference point)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ __launch_bounds__(256) void v0(const double *__restrict__ in, double *__restrict__ out, const int16_t *__restrict__ key,
                                          const double *__restrict__ pwp, const double *__restrict__ dl, size_t stride, int W, int H) {
    const int x = 1 + blockIdx.x * 256 + threadIdx.x;
    const int y0 = 1 + blockIdx.y * 4;
    if (x >= W - 1) return;
    double col[6], dE[4], dW[4];
    for (int i = 0; i < 6; i++) col[i] = in[(size_t)min(y0 - 1 + i, H - 1) * W + x];
    for (int i = 0; i < 4; i++) { const size_t p = (size_t)min(y0 + i, H - 2) * W + x; dE[i] = in[p + 1]; dW[i] = in[p - 1]; }
    for (int i = 0; i < 4; i++) {
        if (y0 + i > H - 2) break;
        const size_t p = (size_t)(y0 + i) * W + x;
        const int k = (int)(col[i + 1] - 1.5);
        const size_t c = p + (size_t)(k & 1) * stride;
        const double r = col[i] + col[i + 2] + dE[i] + dW[i] + pwp[c] + dl[c] + (double)key[c];
        out[p] = r;
    }
}

__global__ __launch_bounds__(256) void v1(const double *__restrict__ in, double *__restrict__ out, const int16_t *__restrict__ key,
                                          const double *__restrict__ pwp, const double *__restrict__ dl, size_t stride, int W, int H) {
    const int x = 2 * (blockIdx.x * 256 + threadIdx.x); // even column, pixels x and x+1
    const int y0 = 1 + blockIdx.y * 2;
    if (x >= W) return;
    const int lane = threadIdx.x & 63;
    double2 col[4];
    for (int i = 0; i < 4; i++) col[i] = *(const double2 *)(in + (size_t)min(y0 - 1 + i, H - 1) * W + x);
    for (int i = 0; i < 2; i++) {
        if (y0 + i > H - 2) break;
        const size_t p = (size_t)(y0 + i) * W + x;
        const double2 c = col[i + 1];
        double wL = __shfl_up(c.y, 1), eR = __shfl_down(c.x, 1);
        if (lane == 0 && x > 0) wL = in[p - 1];
        if (lane == 63 && x + 2 < W) eR = in[p + 2];
        const double2 p0 = *(const double2 *)(pwp + p), p1 = *(const double2 *)(pwp + p + stride);
        const double2 d0 = *(const double2 *)(dl + p), d1 = *(const double2 *)(dl + p + stride);
        const short2 k0 = *(const short2 *)(key + p), k1 = *(const short2 *)(key + p + stride);
        const int ka = (int)(c.x - 1.5) & 1, kb = (int)(c.y - 1.5) & 1;
        double2 r;
        r.x = col[i].x + col[i + 2].x + c.y + wL + (ka ? p1.x + d1.x + k1.x : p0.x + d0.x + k0.x);
        r.y = col[i].y + col[i + 2].y + c.x + eR + (kb ? p1.y + d1.y + k1.y : p0.y + d0.y + k0.y);
        *(double2 *)(out + p) = r;
    }
}

__global__ __launch_bounds__(256) void v2(const float4 *__restrict__ in, float4 *__restrict__ out, size_t n) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t s = (size_t)gridDim.x * 256;
    for (; i < n; i += s) out[i] = in[i];
}

int main() {
    const int W = 4096, H = 3072 * 2; // both directions as one image
    const size_t px = (size_t)W * H;
    double *in, *out, *pwp, *dl; int16_t *key;
    CK(hipMalloc(&in, px * 8)); CK(hipMalloc(&out, px * 8)); CK(hipMalloc(&pwp, px * 16)); CK(hipMalloc(&dl, px * 16)); CK(hipMalloc(&key, px * 4));
    std::vector<double> h(px);
    uint32_t s = 12345;
    for (size_t i = 0; i < px; i++) { s = s * 1664525u + 1013904223u; h[i] = 20.0 + (double)((s >> 16) & 1) + 0.25; } // random parity of int(d - 1.5)
    CK(hipMemcpy(in, h.data(), px * 8, hipMemcpyHostToDevice));
    CK(hipMemset(pwp, 0, px * 16)); CK(hipMemset(dl, 0, px * 16)); CK(hipMemset(key, 0, px * 4)); CK(hipMemset(out, 0, px * 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float ms;
    for (int rep = 0; rep < 2; rep++) {
        CK(hipEventRecord(e0));
        for (int it = 0; it < 20; it++) hipLaunchKernelGGL(v0, dim3((W + 255) / 256, (H + 3) / 4), dim3(256), 0, 0, in, out, key, pwp, dl, px, W, H);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        printf("v0  8B loads, 1 px/lane : %.1f us/launch  (%.2f TB/s on 52 B/px)\n", ms * 50, px * 52.0 / (ms / 20 * 1e-3) / 1e12);
        CK(hipEventRecord(e0));
        for (int it = 0; it < 20; it++) hipLaunchKernelGGL(v1, dim3((W / 2 + 255) / 256, (H + 1) / 2), dim3(256), 0, 0, in, out, key, pwp, dl, px, W, H);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        printf("v1 16B loads, 2 px/lane : %.1f us/launch  (%.2f TB/s on 52 B/px)\n", ms * 50, px * 52.0 / (ms / 20 * 1e-3) / 1e12);
        const size_t n4 = px * 26 / 16; // 26 B read + 26 B written per pixel = 52 B/px
        CK(hipEventRecord(e0));
        for (int it = 0; it < 20; it++) hipLaunchKernelGGL(v2, dim3(8192), dim3(256), 0, 0, (const float4 *)pwp, (float4 *)dl, n4 < px ? n4 : px);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        printf("v2 float4 copy same bytes: %.1f us/launch (%.2f TB/s)\n", ms * 50, (n4 < px ? n4 : px) * 32.0 / (ms / 20 * 1e-3) / 1e12);
    }
    return 0;
}
