// k_cloud.hip -- CStereoMatching::DisparityToCloud<double>, reconstruction/CStereoMatching.cpp:682-761.
//   mask erosion by an ellipse of size ceil(0.02*rows) (.cpp:703-705): only "eroded == 255" is ever
//   tested (.cpp:741), i.e. a binary question -> per-row prefix counts of non-255 pixels + one span
//   test per structuring-element row (pixels outside the image are ignored, as cv::erode's default
//   border does).
//   Q-reprojection + R_final*X + T_final (.cpp:745-749), emitted in row-major pixel order: per-row
//   counts -> exclusive scan -> ordered compaction (wave ballots), so the point list has exactly the
//   InsertPoint order of the reference.
#include "rsm_dev.h"

#include <algorithm>

// prefix[y*(W+1) + x] = number of pixels != 255 in row y, columns [0, x)
__global__ __launch_bounds__(256) void k_bad_prefix(const uint8_t *__restrict__ mask, int W, int H,
                                                    int32_t *__restrict__ prefix) {
    const int y = blockIdx.x;
    if (y >= H) return;
    __shared__ int s_w[4], s_base;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const uint8_t *m = mask + (size_t)y * W;
    int32_t *o = prefix + (size_t)y * (W + 1);
    if (tid == 0) {
        s_base = 0;
        o[0] = 0;
    }
    __syncthreads();
    for (int x0 = 0; x0 < W; x0 += 256) {
        const int x = x0 + tid;
        const bool bad = (x < W) && (m[x] != 255);
        const unsigned long long b = __ballot(bad);
        const int incl = __popcll(b & ((lane == 63) ? ~0ull : ((2ull << lane) - 1ull)));
        if (lane == 0) s_w[wid] = __popcll(b);
        __syncthreads();
        int base = s_base;
        for (int w = 0; w < wid; w++) base += s_w[w];
        if (x < W) o[x + 1] = base + incl;
        __syncthreads();
        if (tid == 0) s_base += s_w[0] + s_w[1] + s_w[2] + s_w[3];
        __syncthreads();
    }
}

void launch_bad_prefix(const uint8_t *mask, int W, int H, int32_t *prefix, hipStream_t st) {
    hipLaunchKernelGGL(k_bad_prefix, dim3(H), dim3(256), 0, st, mask, W, H, prefix);
}

// Coarse map for a quick accept: blk[by * nbx + bx] = 1 if the EB x EB block holds any pixel != 255.  A pixel whose
// structuring-element bounding box only meets clean blocks is certainly kept; only the band of ~ksize pixels around
// the mask's edges and holes runs the per-row span test.
#define EB 32
__global__ void k_bad_blocks(const int32_t *__restrict__ prefix, int W, int H, int nbx, int nby, uint8_t *__restrict__ blk) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nbx * nby) return;
    const int bx = b % nbx, by = b / nbx;
    const int x0 = bx * EB, x1 = min(x0 + EB, W), y1 = min(by * EB + EB, H);
    int bad = 0;
    for (int y = by * EB; y < y1; y++) {
        const int32_t *p = prefix + (size_t)y * (W + 1);
        bad |= p[x1] - p[x0];
    }
    blk[b] = bad != 0;
}

__device__ __forceinline__ bool eroded_is_255(const int32_t *__restrict__ prefix, const uint8_t *__restrict__ blk, int W,
                                              int H, int ksize, const int *__restrict__ j1, const int *__restrict__ j2,
                                              int x, int y) {
    const int ax = ksize / 2, ay = ksize / 2;
    if (blk) {
        const int nbx = (W + EB - 1) / EB;
        const int bx0 = max(x - ax, 0) / EB, bx1 = min(x + ksize - 1 - ax, W - 1) / EB;
        const int by0 = max(y - ay, 0) / EB, by1 = min(y + ksize - 1 - ay, H - 1) / EB;
        int bad = 0;
        for (int by = by0; by <= by1; by++)
            for (int bx = bx0; bx <= bx1; bx++) bad |= blk[by * nbx + bx];
        if (!bad) return true;
    }
    // no early exit inside a batch: with one the span loads of consecutive rows could not be in flight together,
    // and the loop would cost ksize dependent memory round trips
    for (int i0 = 0; i0 < ksize; i0 += 8) {
        int bad = 0;
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int i = min(i0 + u, ksize - 1);
            const int yy = y + i - ay;
            const int a = j1[i], b = j2[i];
            const int lo = max(x + a - ax, 0), hi = min(x + b - 1 - ax, W - 1);
            const bool use = yy >= 0 && yy < H && b > a && lo <= hi;
            const int32_t *p = prefix + (size_t)(use ? yy : y) * (W + 1);
            const int dcount = p[(use ? hi : x) + 1] - p[use ? lo : x];
            bad |= use ? dcount : 0;
        }
        if (bad) return false;
    }
    return true;
}

__global__ void k_erode_binary(const int32_t *__restrict__ prefix, int W, int H, int ksize,
                               const int *__restrict__ j1, const int *__restrict__ j2,
                               uint8_t *__restrict__ dst) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= W || y >= H) return;
    dst[(size_t)y * W + x] = eroded_is_255(prefix, nullptr, W, H, ksize, j1, j2, x, y) ? 255 : 0;
}

void launch_erode_binary(const int32_t *prefix, int W, int H, int ksize, const int *d_j1, const int *d_j2,
                         uint8_t *dst255, hipStream_t st) {
    hipLaunchKernelGGL(k_erode_binary, dim3((W + 255) / 256, H), dim3(256), 0, st, prefix, W, H, ksize, d_j1,
                       d_j2, dst255);
}

struct CloudArgs {
    const double *disp;
    const int32_t *prefix;
    const uint8_t *img;
    int W, H, ksize;
    const int *j1, *j2;
    const double *q; // 16 doubles, column 3 already scaled (.cpp:698)
    const double *R, *T;
    Mg own;
    uint8_t *flags; // W*H scratch: pass 0 stores cloud_flag, pass 1 reads it back; then the coarse bad-block map
    const uint8_t *blk;
    int32_t *row_count;
    int64_t *row_offset;
    int64_t *npoints;
    double *xyz;
    uint8_t *bgr;
    int64_t max_points;
};

__device__ __forceinline__ bool cloud_flag(const CloudArgs &c, int x, int y) {
    if (x > c.own.XR) return false;
    if (c.disp[(size_t)y * c.W + x] == (double)NOMATCH) return false; // .cpp:743
    return eroded_is_255(c.prefix, c.blk, c.W, c.H, c.ksize, c.j1, c.j2, x, y); // .cpp:741
}

// pass 0: count per row; pass 1: ordered write. One 256-thread workgroup per margin row.
template <int PASS>
__global__ __launch_bounds__(256) void k_cloud(CloudArgs c) {
    const int y = c.own.YL + blockIdx.x;
    if (y > c.own.YR) return;
    __shared__ int s_w[4];
    __shared__ long long s_base;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    if (PASS == 0) { // flags + the row's count: no ordering needed yet, so no barrier per 256-pixel chunk
        int cnt = 0;
        for (int x = c.own.XL + tid; x <= c.own.XR; x += 256) {
            const bool f = cloud_flag(c, x, y);
            c.flags[(size_t)y * c.W + x] = f;
            cnt += f;
        }
        for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o);
        if (lane == 0) s_w[wid] = cnt;
        __syncthreads();
        if (tid == 0) c.row_count[blockIdx.x] = s_w[0] + s_w[1] + s_w[2] + s_w[3];
        return;
    }
    if (tid == 0) s_base = (PASS == 1) ? (long long)c.row_offset[blockIdx.x] : 0ll;
    __syncthreads();
    double q03 = 0, q13 = 0, qz = 0, qw = 0, q32 = 0;
    if (PASS == 1) {
        q03 = c.q[3];
        q13 = c.q[7];
        qz = c.q[11];
        qw = c.q[15];
        q32 = c.q[14];
    }
    for (int x0 = c.own.XL; x0 <= c.own.XR; x0 += 256) {
        const int x = x0 + tid;
        bool f;
        if (PASS == 0) {
            f = cloud_flag(c, x, y);
            if (x <= c.own.XR) c.flags[(size_t)y * c.W + x] = f;
        } else {
            f = (x <= c.own.XR) && c.flags[(size_t)y * c.W + x];
        }
        const unsigned long long b = __ballot(f);
        if (lane == 0) s_w[wid] = __popcll(b);
        __syncthreads();
        if (PASS == 1 && f) {
            long long pos = s_base + __popcll(b & ((1ull << lane) - 1ull));
            for (int w = 0; w < wid; w++) pos += s_w[w];
            if (pos < c.max_points) {
                const double dv = c.disp[(size_t)y * c.W + x];
                const double qy = y + q13;           // .cpp:736
                const double iW = 1. / (qw + q32 * dv); // .cpp:745
                const double F0 = (q03 + (double)x) * iW;
                const double F1 = qy * iW;
                const double F2 = qz * iW;
                if (c.xyz) {
                    for (int i = 0; i < 3; i++) // R_final*Fout + T_final, .cpp:749
                        c.xyz[3 * pos + i] = (c.R[3 * i] * F0 + c.R[3 * i + 1] * F1 + c.R[3 * i + 2] * F2) + c.T[i];
                }
                if (c.bgr) {
                    const uint8_t *s = c.img + ((size_t)y * c.W + x) * 3; // .cpp:735,740,756
                    c.bgr[3 * pos] = s[0];
                    c.bgr[3 * pos + 1] = s[1];
                    c.bgr[3 * pos + 2] = s[2];
                }
            }
        }
        __syncthreads();
        if (tid == 0) s_base += s_w[0] + s_w[1] + s_w[2] + s_w[3];
        __syncthreads();
    }
    if (PASS == 0 && tid == 0) c.row_count[blockIdx.x] = (int)s_base;
}

// exclusive scan of row counts (single workgroup; rows <= a few thousand)
__global__ __launch_bounds__(256) void k_row_scan(const int32_t *__restrict__ cnt, int rows, int64_t *__restrict__ off,
                                                  int64_t *__restrict__ total) {
    __shared__ long long s_w[4];
    __shared__ long long s_base;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    if (tid == 0) s_base = 0;
    __syncthreads();
    for (int r0 = 0; r0 < rows; r0 += 256) {
        const int r = r0 + tid;
        const long long v = (r < rows) ? cnt[r] : 0;
        long long incl = v;
        for (int o = 1; o < 64; o <<= 1) {
            const long long t = __shfl_up(incl, o);
            if (lane >= o) incl += t;
        }
        if (lane == 63) s_w[wid] = incl;
        __syncthreads();
        long long base = s_base;
        for (int w = 0; w < wid; w++) base += s_w[w];
        if (r < rows) off[r] = base + incl - v;
        __syncthreads();
        if (tid == 0) s_base += s_w[0] + s_w[1] + s_w[2] + s_w[3];
        __syncthreads();
    }
    if (tid == 0) *total = s_base;
}

// blk[CLOUD_BLOCKS(W, H)]: the coarse bad-block map of `prefix` (quick accept of the erosion test)
void launch_bad_blocks(const int32_t *prefix, int W, int H, uint8_t *blk, hipStream_t st) {
    const int nbx = (W + EB - 1) / EB, nby = (H + EB - 1) / EB;
    hipLaunchKernelGGL(k_bad_blocks, dim3((nbx * nby + 255) / 256), dim3(256), 0, st, prefix, W, H, nbx, nby, blk);
}

void launch_cloud(const double *disp, const int32_t *bad_prefix, const uint8_t *img, int W, int H, int ksize,
                  const int *d_j1, const int *d_j2, const double *q16_scaled, const double *R, const double *T,
                  Mg own, uint8_t *flags, const uint8_t *blk, int32_t *row_count, int64_t *row_offset, int64_t *d_npoints,
                  double *xyz, uint8_t *bgr, int64_t max_points, hipStream_t st) {
    const int rows = own.YR - own.YL + 1;
    if (rows <= 0 || own.XR < own.XL) {
        (void)hipMemsetAsync(d_npoints, 0, sizeof(int64_t), st);
        return;
    }
    CloudArgs c{disp, bad_prefix, img, W, H, ksize, d_j1, d_j2, q16_scaled, R, T, own, flags, blk,
                row_count, row_offset, d_npoints, xyz, bgr, max_points};
    hipLaunchKernelGGL(k_cloud<0>, dim3(rows), dim3(256), 0, st, c);
    hipLaunchKernelGGL(k_row_scan, dim3(1), dim3(256), 0, st, row_count, rows, row_offset, d_npoints);
    hipLaunchKernelGGL(k_cloud<1>, dim3(rows), dim3(256), 0, st, c);
}

// ---------------------------------------------------------------- 16-byte point records (the RCCL payload)
// fp64 xyz -> float as CCloudOptimization::InsertPoint does (CloudOptimization/CCloudOptimization.cpp:61), + BGR.
__global__ void k_pack_cloud16(const double *__restrict__ xyz, const uint8_t *__restrict__ bgr, int64_t n, uint4 *__restrict__ dst) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        uint4 v;
        v.x = __float_as_uint((float)xyz[3 * i]);
        v.y = __float_as_uint((float)xyz[3 * i + 1]);
        v.z = __float_as_uint((float)xyz[3 * i + 2]);
        v.w = (uint32_t)bgr[3 * i] | ((uint32_t)bgr[3 * i + 1] << 8) | ((uint32_t)bgr[3 * i + 2] << 16);
        dst[i] = v;
    }
}
void launch_pack_cloud16(const double *xyz, const uint8_t *bgr, int64_t n, void *dst, hipStream_t st) {
    if (n <= 0) return;
    const int64_t blocks = std::min<int64_t>((n + 255) / 256, 8192);
    hipLaunchKernelGGL(k_pack_cloud16, dim3((unsigned)blocks), dim3(256), 0, st, xyz, bgr, n, (uint4 *)dst);
}
