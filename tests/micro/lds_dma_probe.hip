// Probe: LDS-DMA (global_load_lds_dwordx4 / _dword) of a 512-byte row segment of doubles and a 128-byte segment of uint16 by the
// lower 32 lanes, for every element offset 0..3 of the source (is a 16-byte load from an 8-byte-aligned address, or a 4-byte
// load from a 2-byte-aligned one, delivered intact?).  k_refine_skew stages its rows this way.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glob_void;
__global__ void k(const double *in, const unsigned short *k16, double *out, int off) {
    __shared__ __attribute__((aligned(16))) double s[68];
    __shared__ __attribute__((aligned(16))) unsigned short sk[64];
    const int lane = threadIdx.x & 63;
    if (lane < 32) {
        __builtin_amdgcn_global_load_lds((glob_void *)(in + off + 2 * lane), (lds_void *)&s[2], 16, 0, 0);
        __builtin_amdgcn_global_load_lds((glob_void *)(k16 + off + 2 * lane), (lds_void *)&sk[0], 4, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    out[lane] = s[lane + 2] + (double)sk[lane];
}
int main() {
    std::vector<double> h(256); std::vector<unsigned short> hk(256);
    for (int i = 0; i < 256; i++) { h[i] = 1000.0 * i + 0.5; hk[i] = (unsigned short)(i * 3 + 1); }
    double *d, *o; unsigned short *dk;
    hipMalloc(&d, 256 * 8); hipMalloc(&o, 64 * 8); hipMalloc(&dk, 256 * 2);
    hipMemcpy(d, h.data(), 256 * 8, hipMemcpyHostToDevice); hipMemcpy(dk, hk.data(), 256 * 2, hipMemcpyHostToDevice);
    for (int off = 0; off < 4; off++) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, dk, o, off);
        double r[64]; hipMemcpy(r, o, sizeof r, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int i = 0; i < 64; i++) bad += r[i] != h[off + i] + (double)hk[off + i];
        printf("element offset %d: %s (%d of 64 wrong) err %s\n", off, bad ? "WRONG" : "ok", bad, hipGetErrorString(hipGetLastError()));
    }
    return 0;
}
