// Development probe: operand layout of v_mfma_i32_16x16x64_i8 on gfx950 (assumed: lane l holds A[l & 15][16 * (l >> 4) .. + 15],
// B[16 * (l >> 4) .. + 15][l & 15], D[(l >> 4) * 4 + r][l & 15]).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
typedef int v4i __attribute__((ext_vector_type(4)));
__global__ void k(const int8_t *A /*16x64*/, const int8_t *B /*64x16 stored as Bt[n][k]*/, int *D /*16x16*/) {
    const int l = threadIdx.x;
    v4i a, b, c = {0, 0, 0, 0};
    const int8_t *pa = A + (l & 15) * 64 + 16 * (l >> 4);
    const int8_t *pb = B + (l & 15) * 64 + 16 * (l >> 4);
    a = *(const v4i *)pa;
    b = *(const v4i *)pb;
    c = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; r++) D[((l >> 4) * 4 + r) * 16 + (l & 15)] = c[r];
}
int main() {
    int8_t hA[16 * 64], hB[16 * 64];
    srand(7);
    for (int i = 0; i < 16 * 64; i++) { hA[i] = (int8_t)(rand() % 256 - 128); hB[i] = (int8_t)(rand() % 256 - 128); }
    int8_t *dA, *dB; int *dD; int hD[256];
    hipMalloc(&dA, 1024); hipMalloc(&dB, 1024); hipMalloc(&dD, 1024);
    hipMemcpy(dA, hA, 1024, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 1024, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    hipMemcpy(hD, dD, 1024, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int m = 0; m < 16; m++) for (int n = 0; n < 16; n++) {
        int s = 0;
        for (int kk = 0; kk < 64; kk++) s += (int)hA[m * 64 + kk] * (int)hB[n * 64 + kk];
        bad += s != hD[m * 16 + n];
    }
    printf("mfma_i32_16x16x64_i8 layout probe: %d mismatches of 256\n", bad);
    return 0;
}
