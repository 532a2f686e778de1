// k_pyramid.hip -- pyramid construction, margins and NCC window-sum tables.
//   pyr_down      replaces cv::pyrDown at reconstruction/CStereoMatching.cpp:1049-1050
//   find_margin   replaces CStereoMatching::FindMargin, .cpp:1011-1038
//   box_sums      hoists the per-candidate mean/norm of CManageData::WindowToVec
//                 (CManageData.cpp:81-90) into exact int32 window sums, computed once per level
#include "rsm_dev.h"

// ---------------------------------------------------------------- fills
template <typename T>
__global__ void k_fill(T *p, size_t n, T v) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) p[i] = v;
}
template <typename T>
static void fill_t(T *p, size_t n, T v, hipStream_t st) {
    if (n == 0) return;
    size_t blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(k_fill<T>, dim3((unsigned)blocks), dim3(256), 0, st, p, n, v);
}
void launch_fill_i16(int16_t *p, size_t n, int16_t v, hipStream_t st) { fill_t(p, n, v, st); }
void launch_fill_f64(double *p, size_t n, double v, hipStream_t st) { fill_t(p, n, v, st); }
void launch_fill_i32(int32_t *p, size_t n, int32_t v, hipStream_t st) { fill_t(p, n, v, st); }

// ---------------------------------------------------------------- pyrDown (8U, C = 1 or 3)
// 5x5 [1 4 6 4 1]^2, BORDER_REFLECT_101, (v + 128) >> 8; dst = ((W+1)/2, (H+1)/2).
__device__ __forceinline__ int reflect101(int p, int n) {
    if (n == 1) return 0;
    while (p < 0 || p >= n) p = (p < 0) ? -p : 2 * n - 2 - p;
    return p;
}

template <int C>
__global__ void k_pyr_down(const uint8_t *__restrict__ src, int W, int H, uint8_t *__restrict__ dst, int Wd, int Hd) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    if (x >= Wd || y >= Hd) return;
    int sx[5], sy[5];
#pragma unroll
    for (int k = 0; k < 5; k++) {
        sx[k] = reflect101(2 * x - 2 + k, W) * C;
        sy[k] = reflect101(2 * y - 2 + k, H);
    }
    int acc[C];
#pragma unroll
    for (int c = 0; c < C; c++) acc[c] = 0;
#pragma unroll
    for (int j = 0; j < 5; j++) {
        const uint8_t *row = src + (size_t)sy[j] * W * C;
        const int wj = (j == 0 || j == 4) ? 1 : (j == 2 ? 6 : 4);
#pragma unroll
        for (int c = 0; c < C; c++) {
            const int h = row[sx[0] + c] + row[sx[4] + c] + 4 * (row[sx[1] + c] + row[sx[3] + c]) + 6 * row[sx[2] + c];
            acc[c] += wj * h;
        }
    }
#pragma unroll
    for (int c = 0; c < C; c++) dst[((size_t)y * Wd + x) * C + c] = (uint8_t)((acc[c] + 128) >> 8);
}

void launch_pyr_down(const uint8_t *src, int W, int H, int C, uint8_t *dst, hipStream_t st) {
    const int Wd = (W + 1) / 2, Hd = (H + 1) / 2;
    dim3 grid((Wd + 255) / 256, Hd);
    if (C == 3) hipLaunchKernelGGL(k_pyr_down<3>, grid, dim3(256), 0, st, src, W, H, dst, Wd, Hd);
    else hipLaunchKernelGGL(k_pyr_down<1>, grid, dim3(256), 0, st, src, W, H, dst, Wd, Hd);
}

// ---------------------------------------------------------------- FindMargin
__global__ void k_margin_init(int *out4, int W, int H, int r) {
    out4[0] = W - 1 - r; // XL
    out4[1] = r;         // XR
    out4[2] = H - 1 - r; // YL
    out4[3] = r;         // YR
}

// FM_ROWS rows y in [r, H-r) per block (one wave per row at a time); columns [r, W-r).  Few blocks, so that the
// four global atomics per block do not pile up on one address (~88 same-address atomics per microsecond).
#define FM_ROWS 16
__device__ __forceinline__ void find_margin_body(const uint8_t *__restrict__ mask, int W, int H, int r, int *out4, int bx) {
    const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
    int lo = 0x7fffffff, hi = -1, ylo = 0x7fffffff, yhi = -1;
    for (int k = wid; k < FM_ROWS; k += 4) {
        const int y = r + bx * FM_ROWS + k;
        if (y >= H - r) break;
        const uint8_t *p = mask + (size_t)y * W;
        int rlo = 0x7fffffff, rhi = -1;
        // 16 bytes per lane and load (row bases are only byte aligned: unaligned dwordx4 loads)
        for (int x0 = r + 16 * lane; x0 < W - r; x0 += 16 * 64) {
            if (x0 + 16 <= W - r) {
                uint32_t v[4];
                __builtin_memcpy(v, p + x0, 16);
#pragma unroll
                for (int q = 0; q < 16; q++)
                    if (((v[q >> 2] >> (8 * (q & 3))) & 255u) == 255u) {
                        rlo = min(rlo, x0 + q);
                        rhi = max(rhi, x0 + q);
                    }
            } else {
                for (int x = x0; x < W - r; x++)
                    if (p[x] == 255) {
                        rlo = min(rlo, x);
                        rhi = max(rhi, x);
                    }
            }
        }
        if (rhi >= 0) {
            lo = min(lo, rlo);
            hi = max(hi, rhi);
            ylo = min(ylo, y);
            yhi = max(yhi, y);
        }
    }
    for (int o = 32; o > 0; o >>= 1) {
        lo = min(lo, __shfl_xor(lo, o));
        hi = max(hi, __shfl_xor(hi, o));
        ylo = min(ylo, __shfl_xor(ylo, o));
        yhi = max(yhi, __shfl_xor(yhi, o));
    }
    __shared__ int sv[4][4];
    if (lane == 0) {
        sv[wid][0] = lo;
        sv[wid][1] = hi;
        sv[wid][2] = ylo;
        sv[wid][3] = yhi;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 1; i < 4; i++) {
            lo = min(lo, sv[i][0]);
            hi = max(hi, sv[i][1]);
            ylo = min(ylo, sv[i][2]);
            yhi = max(yhi, sv[i][3]);
        }
        if (hi >= 0) { // a stale read only costs a redundant atomic
            volatile int *o = out4;
            if (lo < o[0]) atomicMin(&out4[0], lo);
            if (hi > o[1]) atomicMax(&out4[1], hi);
            if (ylo < o[2]) atomicMin(&out4[2], ylo);
            if (yhi > o[3]) atomicMax(&out4[3], yhi);
        }
    }
}

__global__ __launch_bounds__(256) void k_find_margin(const uint8_t *__restrict__ mask, int W, int H, int r, int *out4) {
    find_margin_body(mask, W, H, r, out4, blockIdx.x);
}
// all levels and views of a pair in one launch: blockIdx.y selects the image
#define FM_BATCH 24 // images per launch (2 views x RSM_MAX_LEVELS)
struct MarginBatch {
    const uint8_t *mask[FM_BATCH];
    int W[FM_BATCH], H[FM_BATCH];
};
__global__ __launch_bounds__(256) void k_find_margin_batch(MarginBatch b, int r, int *out4) {
    const int i = blockIdx.y;
    if ((int)blockIdx.x * FM_ROWS >= b.H[i] - 2 * r) return; // uniform: smaller levels need fewer blocks
    find_margin_body(b.mask[i], b.W[i], b.H[i], r, out4 + 4 * i, blockIdx.x);
}

void launch_find_margin(const uint8_t *mask, int W, int H, int r, int *out4, hipStream_t st) {
    hipLaunchKernelGGL(k_margin_init, dim3(1), dim3(1), 0, st, out4, W, H, r);
    const int rows = H - 2 * r;
    if (rows <= 0 || W - 2 * r <= 0) return;
    hipLaunchKernelGGL(k_find_margin, dim3((rows + FM_ROWS - 1) / FM_ROWS), dim3(256), 0, st, mask, W, H, r, out4);
}

// n images (masks[i], Ws[i] x Hs[i]); out4[4 * i ..] receives {XL, XR, YL, YR}; h_init: 4 * n ints of host scratch that
// must stay valid until the stream has consumed the copy
void launch_find_margin_batch(int n, const uint8_t *const *masks, const int *Ws, const int *Hs, int r, int *out4, int *h_init,
                              hipStream_t st) {
    MarginBatch b{};
    int maxrows = 0;
    for (int i = 0; i < n; i++) {
        b.mask[i] = masks[i];
        b.W[i] = Ws[i];
        b.H[i] = Hs[i];
        h_init[4 * i + 0] = Ws[i] - 1 - r; // inverted defaults (.cpp:1014-1017)
        h_init[4 * i + 1] = r;
        h_init[4 * i + 2] = Hs[i] - 1 - r;
        h_init[4 * i + 3] = r;
        if (Ws[i] - 2 * r > 0) maxrows = max(maxrows, Hs[i] - 2 * r);
    }
    (void)hipMemcpyAsync(out4, h_init, sizeof(int) * 4 * n, hipMemcpyHostToDevice, st);
    if (maxrows <= 0) return;
    hipLaunchKernelGGL(k_find_margin_batch, dim3((maxrows + FM_ROWS - 1) / FM_ROWS, n), dim3(256), 0, st, b, r, out4);
}

// masked pixels inside a margin (V_top of the metric): rows strided over a fixed number of blocks, one atomic each
__global__ __launch_bounds__(256) void k_count_masked(const uint8_t *__restrict__ mask, int W, Mg m, unsigned long long *count) {
    int c = 0;
    for (int y = m.YL + blockIdx.x; y <= m.YR; y += gridDim.x)
        for (int x = m.XL + threadIdx.x; x <= m.XR; x += 256) c += mask[(size_t)y * W + x] == 255;
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
    __shared__ int sc[4];
    if ((threadIdx.x & 63) == 0) sc[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) {
        c = sc[0] + sc[1] + sc[2] + sc[3];
        if (c) atomicAdd(count, (unsigned long long)c);
    }
}
void launch_count_masked(const uint8_t *mask, int W, int H, Mg m, unsigned long long *d_count, hipStream_t st) {
    (void)H;
    (void)hipMemsetAsync(d_count, 0, sizeof(unsigned long long), st);
    if (m.YR < m.YL || m.XR < m.XL) return;
    hipLaunchKernelGGL(k_count_masked, dim3(min(m.YR - m.YL + 1, 512)), dim3(256), 0, st, mask, W, m, d_count);
}

// ---------------------------------------------------------------- window-sum tables
// S1(x,y) = sum of the (2r+1)x(2r+1)x3 bytes centred on (x,y); S2 = sum of their squares.
// Defined where the window fits; 0 elsewhere. Separable: horizontal then vertical.
// Horizontal pass on the BGRX copy: one aligned dword per pixel, v_sad_u8 sums its three bytes (X = 0) and
// v_dot4_u32_u8 with itself their squares.  A thread produces BH_P consecutive pixels with a sliding window: every
// dword is loaded and reduced once per thread (BH_P + 2R of them instead of BH_P * (2R + 1)).
#define BH_P 4
template <int R>
__global__ void k_box_h(const uint32_t *__restrict__ img4, int W, int H, int32_t *__restrict__ h1,
                        int32_t *__restrict__ h2) {
    const int x0 = (blockIdx.x * blockDim.x + threadIdx.x) * BH_P;
    const int y = blockIdx.y;
    if (x0 >= W) return;
    const uint32_t *row = img4 + (size_t)y * W;
    constexpr int NW = BH_P + 2 * R; // dwords x0 - R .. x0 + BH_P - 1 + R
    int e1[NW], e2[NW];
#pragma unroll
    for (int i = 0; i < NW; i++) {
        const int col = x0 - R + i;
        const uint32_t v = (col >= 0 && col < W) ? row[col] : 0u;
        e1[i] = (int)__builtin_amdgcn_sad_u8(v, 0u, 0u);
        e2[i] = (int)__builtin_amdgcn_udot4(v, v, 0u, false);
    }
    int s1 = 0, s2 = 0;
#pragma unroll
    for (int i = 0; i <= 2 * R; i++) {
        s1 += e1[i];
        s2 += e2[i];
    }
#pragma unroll
    for (int p = 0; p < BH_P; p++) {
        const int x = x0 + p;
        if (x < W) {
            const bool ok = x - R >= 0 && x + R < W;
            h1[(size_t)y * W + x] = ok ? s1 : 0;
            h2[(size_t)y * W + x] = ok ? s2 : 0;
        }
        if (p + 1 < BH_P) { // slide: dword p leaves, dword p + 2R + 1 enters
            s1 += e1[p + 2 * R + 1] - e1[p];
            s2 += e2[p + 2 * R + 1] - e2[p];
        }
    }
}

// any radius (validate() allows up to 15): one pixel per thread
__global__ void k_box_h_any(const uint32_t *__restrict__ img4, int W, int H, int r, int32_t *__restrict__ h1,
                            int32_t *__restrict__ h2) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    if (x >= W) return;
    uint32_t s1 = 0, s2 = 0;
    if (x - r >= 0 && x + r < W) {
        const uint32_t *p = img4 + (size_t)y * W + (x - r);
        for (int i = 0; i <= 2 * r; i++) {
            const uint32_t v = p[i];
            s1 = __builtin_amdgcn_sad_u8(v, 0u, s1);
            s2 = __builtin_amdgcn_udot4(v, v, s2, false);
        }
    }
    h1[(size_t)y * W + x] = (int32_t)s1;
    h2[(size_t)y * W + x] = (int32_t)s2;
}

// Vertical pass: a thread walks BV_ROWS consecutive rows of its column with running sums (one row enters, one
// leaves) instead of re-adding 2r+1 rows per output.
#define BV_ROWS 16
__global__ void k_box_v(const int32_t *__restrict__ h1, const int32_t *__restrict__ h2, int W, int H, int r,
                        int32_t *__restrict__ S1, int32_t *__restrict__ S2) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y0 = blockIdx.y * BV_ROWS;
    if (x >= W) return;
    int s1 = 0, s2 = 0;
    bool have = false; // s1 / s2 hold the window sums of the previous row
    for (int y = y0; y < min(y0 + BV_ROWS, H); y++) {
        int o1 = 0, o2 = 0;
        if (y - r >= 0 && y + r < H) {
            if (have) {
                s1 += h1[(size_t)(y + r) * W + x] - h1[(size_t)(y - r - 1) * W + x];
                s2 += h2[(size_t)(y + r) * W + x] - h2[(size_t)(y - r - 1) * W + x];
            } else {
                s1 = s2 = 0;
                for (int j = -r; j <= r; j++) {
                    s1 += h1[(size_t)(y + j) * W + x];
                    s2 += h2[(size_t)(y + j) * W + x];
                }
                have = true;
            }
            o1 = s1;
            o2 = s2;
        }
        S1[(size_t)y * W + x] = o1;
        S2[(size_t)y * W + x] = o2;
    }
}

void launch_box_sums(const uint32_t *img4, int W, int H, int r, int32_t *tmp1, int32_t *tmp2, int32_t *S1,
                     int32_t *S2, hipStream_t st) {
    const dim3 gh((W + 256 * BH_P - 1) / (256 * BH_P), H);
    switch (r) {
    case 1: hipLaunchKernelGGL(k_box_h<1>, gh, dim3(256), 0, st, img4, W, H, tmp1, tmp2); break;
    case 2: hipLaunchKernelGGL(k_box_h<2>, gh, dim3(256), 0, st, img4, W, H, tmp1, tmp2); break;
    case 3: hipLaunchKernelGGL(k_box_h<3>, gh, dim3(256), 0, st, img4, W, H, tmp1, tmp2); break;
    case 4: hipLaunchKernelGGL(k_box_h<4>, gh, dim3(256), 0, st, img4, W, H, tmp1, tmp2); break;
    case 5: hipLaunchKernelGGL(k_box_h<5>, gh, dim3(256), 0, st, img4, W, H, tmp1, tmp2); break;
    case 6: hipLaunchKernelGGL(k_box_h<6>, gh, dim3(256), 0, st, img4, W, H, tmp1, tmp2); break;
    case 7: hipLaunchKernelGGL(k_box_h<7>, gh, dim3(256), 0, st, img4, W, H, tmp1, tmp2); break;
    default: hipLaunchKernelGGL(k_box_h_any, dim3((W + 255) / 256, H), dim3(256), 0, st, img4, W, H, r, tmp1, tmp2); break;
    }
    hipLaunchKernelGGL(k_box_v, dim3((W + 255) / 256, (H + BV_ROWS - 1) / BV_ROWS), dim3(256), 0, st, tmp1, tmp2, W, H, r, S1, S2);
}

// ---------------------------------------------------------------- BGR -> BGRX (one dword per pixel)
// The NCC kernels read both views as dwords: a (2R+1)-pixel window row is 2R+1 aligned dwords and the
// zero X byte adds nothing to the v_dot4_u32_u8 sums.
__global__ void k_bgr_to_bgrx(const uint8_t *__restrict__ img, size_t npix, uint32_t *__restrict__ out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < npix; i += stride) {
        const uint8_t *p = img + 3 * i;
        out[i] = (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16);
    }
}
void launch_bgr_to_bgrx(const uint8_t *img, int W, int H, uint32_t *out, hipStream_t st) {
    const size_t n = (size_t)W * H;
    size_t blocks = (n + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(k_bgr_to_bgrx, dim3((unsigned)blocks), dim3(256), 0, st, img, n, out);
}
