// k_rectify.hip -- device part of CStereoMatching::Rectify (reconstruction/CStereoMatching.cpp:144-158):
//   rect_map     cv::initUndistortRectifyMap(..., CV_16SC2, ...)  :144  (zero distortion)
//   remap        cv::remap(..., CV_INTER_LINEAR), BORDER_CONSTANT 0 :154,156
//   erode_gray   cv::erode(mask, ellipse 3*2^(N-1))                :157-158 (grey-level min, border ignored)
// OpenCV 2.4's arithmetic is restated from its published algorithm (fixed-point maps with INTER_BITS = 5,
// 15-bit bilinear weights, (sum + 16384) >> 15); parity unpinned, see DESIGN.md.
#include "rsm_dev.h"

struct RectMapArgs {
    double ir[9];
    double fx, fy, u0, v0;
};

// One thread per destination row: OpenCV advances _x, _y, _w by running sums along the row, and those
// roundings decide the 1/32-pixel ties, so the chain is kept sequential.
__global__ void k_rect_map(RectMapArgs m, int W, int H, int16_t *__restrict__ map1, uint16_t *__restrict__ map2) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= H) return;
    double _x = i * m.ir[1] + m.ir[2], _y = i * m.ir[4] + m.ir[5], _w = i * m.ir[7] + m.ir[8];
    int16_t *m1 = map1 + (size_t)i * W * 2;
    uint16_t *m2 = map2 + (size_t)i * W;
    for (int j = 0; j < W; j++, _x += m.ir[0], _y += m.ir[3], _w += m.ir[6]) {
        const double w = 1. / _w, x = _x * w, y = _y * w;
        const double u = m.fx * x + m.u0, v = m.fy * y + m.v0;
        const int iu = (int)rint(u * 32), iv = (int)rint(v * 32); // saturate_cast<int> = round half to even
        m1[2 * j] = (int16_t)(iu >> 5);
        m1[2 * j + 1] = (int16_t)(iv >> 5);
        m2[j] = (uint16_t)((iv & 31) * 32 + (iu & 31));
    }
}

void launch_rect_map(const double *ir, double fx, double fy, double u0, double v0, int W, int H, int16_t *map1,
                     uint16_t *map2, hipStream_t st) {
    RectMapArgs m;
    for (int k = 0; k < 9; k++) m.ir[k] = ir[k];
    m.fx = fx;
    m.fy = fy;
    m.u0 = u0;
    m.v0 = v0;
    hipLaunchKernelGGL(k_rect_map, dim3((H + 63) / 64), dim3(64), 0, st, m, W, H, map1, map2);
}

template <int C>
__global__ void k_remap(const uint8_t *__restrict__ src, int Ws, int Hs, const int16_t *__restrict__ map1,
                        const uint16_t *__restrict__ map2, int W, int H, uint8_t *__restrict__ dst) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y;
    if (j >= W || i >= H) return;
    const size_t o = (size_t)i * W + j;
    const int sx = map1[2 * o], sy = map1[2 * o + 1];
    const int f = map2[o] & 1023, fx = f & 31, fy = f >> 5;
    const int w00 = (32 - fy) * (32 - fx) * 32, w01 = (32 - fy) * fx * 32, w10 = fy * (32 - fx) * 32, w11 = fy * fx * 32;
    const bool x0 = sx >= 0 && sx < Ws, x1 = sx + 1 >= 0 && sx + 1 < Ws;
    const bool y0 = sy >= 0 && sy < Hs, y1 = sy + 1 >= 0 && sy + 1 < Hs;
#pragma unroll
    for (int c = 0; c < C; c++) {
        const int v00 = (x0 && y0) ? src[((size_t)sy * Ws + sx) * C + c] : 0;
        const int v01 = (x1 && y0) ? src[((size_t)sy * Ws + sx + 1) * C + c] : 0;
        const int v10 = (x0 && y1) ? src[((size_t)(sy + 1) * Ws + sx) * C + c] : 0;
        const int v11 = (x1 && y1) ? src[((size_t)(sy + 1) * Ws + sx + 1) * C + c] : 0;
        dst[o * C + c] = (uint8_t)((v00 * w00 + v01 * w01 + v10 * w10 + v11 * w11 + (1 << 14)) >> 15);
    }
}

void launch_remap(const uint8_t *src, int Ws, int Hs, int C, const int16_t *map1, const uint16_t *map2, int W, int H,
                  uint8_t *dst, hipStream_t st) {
    dim3 grid((W + 255) / 256, H);
    if (C == 3) hipLaunchKernelGGL(k_remap<3>, grid, dim3(256), 0, st, src, Ws, Hs, map1, map2, W, H, dst);
    else hipLaunchKernelGGL(k_remap<1>, grid, dim3(256), 0, st, src, Ws, Hs, map1, map2, W, H, dst);
}

// Grey-level erosion by an ellipse given as per-row spans [j1, j2): horizontal sparse tables
// ST_l(x) = min(src[x .. x + 2^l - 1]) (beyond the row: 255 = neutral, cv::erode ignores the border), then per
// pixel one two-probe range-min per structuring-element row.
__global__ void k_st_level(const uint8_t *__restrict__ prev, int W, int H, int half, uint8_t *__restrict__ next) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= W || y >= H) return;
    const size_t o = (size_t)y * W + x;
    const int a = prev[o], b = (x + half < W) ? prev[o + half] : 255;
    next[o] = (uint8_t)min(a, b);
}

__global__ void k_erode_gray(const uint8_t *__restrict__ st, size_t level_stride, int W, int H, int ksize,
                             const int *__restrict__ j1, const int *__restrict__ j2, uint8_t *__restrict__ dst) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= W || y >= H) return;
    const int ax = ksize / 2, ay = ksize / 2;
    int v = 255;
    for (int i = 0; i < ksize; i++) {
        const int yy = y + i - ay;
        if (yy < 0 || yy >= H) continue;
        const int a = j1[i], b = j2[i];
        if (b <= a) continue;
        const int lo = max(x + a - ax, 0), hi = min(x + b - 1 - ax, W - 1);
        if (lo > hi) continue;
        const int l = 31 - __clz(hi - lo + 1);
        const uint8_t *t = st + (size_t)l * level_stride + (size_t)yy * W;
        v = min(v, min((int)t[lo], (int)t[hi - (1 << l) + 1]));
    }
    dst[(size_t)y * W + x] = (uint8_t)v;
}

// st: (levels) x W*H bytes scratch, level 0 is filled from src here. levels = floor(log2(ksize)) + 1.
void launch_erode_gray(const uint8_t *src, int W, int H, int ksize, const int *d_j1, const int *d_j2, uint8_t *st,
                       uint8_t *dst, hipStream_t st_) {
    const size_t px = (size_t)W * H;
    (void)hipMemcpyAsync(st, src, px, hipMemcpyDeviceToDevice, st_);
    int levels = 1;
    while ((1 << levels) <= ksize) levels++;
    dim3 grid((W + 255) / 256, H);
    for (int l = 1; l < levels; l++)
        hipLaunchKernelGGL(k_st_level, grid, dim3(256), 0, st_, st + (size_t)(l - 1) * px, W, H, 1 << (l - 1), st + (size_t)l * px);
    hipLaunchKernelGGL(k_erode_gray, grid, dim3(256), 0, st_, st, px, W, H, ksize, d_j1, d_j2, dst);
}
