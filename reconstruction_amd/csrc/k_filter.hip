// k_filter.hip -- the per-pair cloud filter of CCloudOptimization::filter (CloudOptimization/CCloudOptimization.cpp:82-121),
// SURVEY 8(f3), on the GPU that produced the cloud (it runs before the RCCL gather and shrinks its payload):
//   pcl::StatisticalOutlierRemoval, meanK = 100, stddevMulThresh = 1 (:85-89; parameters CReconstruction.cpp:18)
//   pcl::NormalEstimationOMP, radius search m_mls_radius = 2.5 (:103-109)
//   normals turned toward CamCenter (:114-121)
// PCL is a third-party dependency that is not in the reference tree; the algorithms are restated from PCL 1.7.2's
// published sources exactly as oracle/cloud_oracle.c states them (parity unpinned; same documented choices: float32
// neighbour distances, dist^2 < r^2, double accumulators).
//
// Neighbour search: the points are sorted by the key of a uniform grid (cell edge h); the 3 x-adjacent cells of one
// (y, z) row of cells have consecutive keys, so the 27 cells around a point are 9 contiguous ranges of the sorted
// array, found by binary search on the keys (no hash table).  Threads walk the sorted order, so the lanes of a wave
// search the same few ranges and their loads coalesce.
//   k nearest: every point within distance h of p lies in those 27 cells.  The (k+1)-th smallest squared distance tau
//     (the point itself included, as in PCL's nearestKSearch(k + 1)) is found by bisection on its float bit pattern over
//     [0, h^2], re-scanning the candidates per step (~900 candidates from L1/L2: cheaper than keeping 100 running minima
//     per thread); then sum sqrt(d2) over d2 < tau + (k + 1 - #less) sqrt(tau).  The sum of <= 100 float32 values in a
//     double is exact (no rounding as long as they span < 2^29), so its order does not matter.
//     Points with fewer than k + 1 candidates within h (isolated points -- what this filter removes -- and patch
//     corners) go to a list and are redone against ALL points by a workgroup each (three-pass radix select).
//   radius search: cell edge = radius, the 27 cells cover the ball.
#include "../../include/rsm.h"
#include "rsm_dev.h"

#include <cstring>
#include <string.h>

#include <rocprim/rocprim.hpp>

#include <algorithm>
#include <cmath>
#include <limits>
#include <vector>

#include <string.h>

struct FGrid {
    float ox, oy, oz, inv_h;
    int nx, ny, nz;
};

__device__ __forceinline__ float fdist2(float ax, float ay, float az, float bx, float by, float bz) {
    const float dx = ax - bx, dy = ay - by, dz = az - bz;
    return (dx * dx + dy * dy) + dz * dz;
}
__device__ __forceinline__ int cell_of(float v, float o, float inv_h, int n) {
    const int c = (int)floorf((v - o) * inv_h);
    return min(max(c, 0), n - 1);
}

__global__ void k_cell_keys(const float *__restrict__ xyz, int64_t n, FGrid g, unsigned long long *__restrict__ keys,
                            unsigned int *__restrict__ vals) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    vals[i] = (unsigned int)i;
    const float x = xyz[3 * i], y = xyz[3 * i + 1], z = xyz[3 * i + 2];
    if (!(isfinite(x) && isfinite(y) && isfinite(z))) { // PCL's searches skip such points: they sort behind every cell
        keys[i] = ~0ull;
        return;
    }
    const int ix = cell_of(x, g.ox, g.inv_h, g.nx), iy = cell_of(y, g.oy, g.inv_h, g.ny), iz = cell_of(z, g.oz, g.inv_h, g.nz);
    keys[i] = ((unsigned long long)iz * g.ny + iy) * g.nx + ix;
}

__global__ void k_gather_sorted(const float *__restrict__ xyz, const unsigned int *__restrict__ vals, int64_t n,
                                float4 *__restrict__ sxyz) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const unsigned int i = vals[j];
    sxyz[j] = make_float4(xyz[3 * (size_t)i], xyz[3 * (size_t)i + 1], xyz[3 * (size_t)i + 2], __uint_as_float(i));
}

__device__ __forceinline__ int lower_bound_key(const unsigned long long *__restrict__ keys, int n, unsigned long long k) {
    int lo = 0, hi = n;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (keys[mid] < k) lo = mid + 1;
        else hi = mid;
    }
    return lo;
}

// the 9 contiguous ranges of the sorted array that hold the 27 cells around p
__device__ __forceinline__ void ranges9(const unsigned long long *__restrict__ keys, int n, const FGrid &g, float px, float py,
                                        float pz, int (&rs)[9], int (&re)[9]) {
    const int ix = cell_of(px, g.ox, g.inv_h, g.nx), iy = cell_of(py, g.oy, g.inv_h, g.ny), iz = cell_of(pz, g.oz, g.inv_h, g.nz);
    const int x0 = max(ix - 1, 0), x1 = min(ix + 1, g.nx - 1);
#pragma unroll
    for (int t = 0; t < 9; t++) {
        const int yy = iy + t % 3 - 1, zz = iz + t / 3 - 1;
        if (yy < 0 || yy >= g.ny || zz < 0 || zz >= g.nz) {
            rs[t] = re[t] = 0;
            continue;
        }
        const unsigned long long base = ((unsigned long long)zz * g.ny + yy) * g.nx;
        rs[t] = lower_bound_key(keys, n, base + x0);
        re[t] = lower_bound_key(keys, n, base + x1 + 1);
    }
}

// mean distance to the k nearest neighbours (statistical_outlier_removal.hpp).  One wave per query point: the squared
// distances to the candidates of the 27 cells are computed once into registers (lane l holds candidates l, l + 64, ...),
// the (k+1)-th smallest is found by bisection on its bit pattern with ballots, the sum by a wave reduction.
// A query with more than 64 * KNN_C candidates re-computes them per bisection step instead.
#define KNN_C 32
__global__ __launch_bounds__(256) void k_sor_knn(const float *__restrict__ xyz, const float4 *__restrict__ sxyz,
                                                  const unsigned long long *__restrict__ keys, int n, FGrid g, float h2, int mean_k,
                                                  const unsigned int *__restrict__ queries, int nq, float *__restrict__ dist_orig,
                                                  unsigned int *__restrict__ redo, int *__restrict__ redo_cnt) {
    const int lane = threadIdx.x & 63;
    const int qi = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (qi >= nq) return; // wave-uniform
    const unsigned int orig = queries[qi];
    const float px = xyz[3 * (size_t)orig], py = xyz[3 * (size_t)orig + 1], pz = xyz[3 * (size_t)orig + 2];
    // lanes 0..8 find one range each
    int my_s = 0, my_e = 0;
    if (lane < 9) {
        const int ix = cell_of(px, g.ox, g.inv_h, g.nx), iy = cell_of(py, g.oy, g.inv_h, g.ny), iz = cell_of(pz, g.oz, g.inv_h, g.nz);
        const int yy = iy + lane % 3 - 1, zz = iz + lane / 3 - 1;
        if (yy >= 0 && yy < g.ny && zz >= 0 && zz < g.nz) {
            const unsigned long long base = ((unsigned long long)zz * g.ny + yy) * g.nx;
            my_s = lower_bound_key(keys, n, base + max(ix - 1, 0));
            my_e = lower_bound_key(keys, n, base + min(ix + 1, g.nx - 1) + 1);
        }
    }
    int rs[9], pre[10];
    pre[0] = 0;
#pragma unroll
    for (int t = 0; t < 9; t++) {
        rs[t] = __shfl(my_s, t);
        pre[t + 1] = pre[t] + (__shfl(my_e, t) - rs[t]);
    }
    const int M = pre[9], want = mean_k + 1; // the point itself is the first of the k + 1 results
    auto cand = [&](int c) -> float { // squared distance to candidate c < M of the concatenated ranges
        int r = 0;
#pragma unroll
        for (int t = 1; t < 9; t++) r += c >= pre[t];
        int off = c, base = 0;
#pragma unroll
        for (int t = 0; t < 9; t++)
            if (t == r) {
                off = c - pre[t];
                base = rs[t];
            }
        const float4 o = sxyz[base + off];
        return fdist2(px, py, pz, o.x, o.y, o.z);
    };
    const float inf = __uint_as_float(0x7f800000u);
    float tau;
    double sum = 0.0;
    int less = 0;
    if (M <= 64 * KNN_C) {
        float d2[KNN_C];
        const int nreg = (M + 63) >> 6; // wave-uniform
#pragma unroll
        for (int i = 0; i < KNN_C; i++) {
            d2[i] = inf;
            if (i < nreg) {
                const int c = lane + 64 * i;
                if (c < M) d2[i] = cand(c);
            }
        }
        auto count_le = [&](float t) {
            int cnt = 0;
#pragma unroll
            for (int i = 0; i < KNN_C; i++)
                if (i < nreg) cnt += __popcll(__ballot(d2[i] <= t));
            return cnt;
        };
        if (count_le(h2) < want) { // not decidable inside the 27 cells
            if (lane == 0) redo[atomicAdd(redo_cnt, 1)] = orig;
            return;
        }
        unsigned int lo = 0u, hi = __float_as_uint(h2); // smallest bit pattern b with count_le(b) >= want
        while (lo < hi) {
            const unsigned int mid = lo + ((hi - lo) >> 1);
            if (count_le(__uint_as_float(mid)) >= want) hi = mid;
            else lo = mid + 1;
        }
        tau = __uint_as_float(lo);
#pragma unroll
        for (int i = 0; i < KNN_C; i++)
            if (i < nreg && d2[i] < tau) {
                sum += (double)sqrtf(d2[i]);
                less++;
            }
    } else {
        auto count_le = [&](float t) {
            int cnt = 0;
            for (int c0 = 0; c0 < M; c0 += 64) { // uniform trip count
                const int c = c0 + lane;
                cnt += __popcll(__ballot(c < M && cand(min(c, M - 1)) <= t));
            }
            return cnt;
        };
        if (count_le(h2) < want) {
            if (lane == 0) redo[atomicAdd(redo_cnt, 1)] = orig;
            return;
        }
        unsigned int lo = 0u, hi = __float_as_uint(h2);
        while (lo < hi) {
            const unsigned int mid = lo + ((hi - lo) >> 1);
            if (count_le(__uint_as_float(mid)) >= want) hi = mid;
            else lo = mid + 1;
        }
        tau = __uint_as_float(lo);
        for (int c = lane; c < M; c += 64) {
            const float v = cand(c);
            if (v < tau) {
                sum += (double)sqrtf(v);
                less++;
            }
        }
    }
    for (int o = 32; o > 0; o >>= 1) {
        sum += __shfl_xor(sum, o);
        less += __shfl_xor(less, o);
    }
    if (lane == 0) dist_orig[orig] = (float)((sum + (double)(want - less) * (double)sqrtf(tau)) / mean_k);
}

// the same for one listed point per workgroup against ALL points: radix select of the (k+1)-th smallest d2
// (bits 30..20, 19..9, 8..0 of its pattern), then the sum
__global__ __launch_bounds__(256) void k_sor_knn_all(const float *__restrict__ xyz, const float4 *__restrict__ sxyz, int n, int mean_k,
                                                      float *__restrict__ dist_orig, const unsigned int *__restrict__ redo,
                                                      const int *__restrict__ redo_cnt) {
    __shared__ int hist[2048];
    __shared__ unsigned int s_prefix;
    __shared__ int s_rank;
    __shared__ double s_sum[256];
    __shared__ int s_less[256];
    const int count = *redo_cnt;
    for (int item = blockIdx.x; item < count; item += gridDim.x) {
        const unsigned int orig = redo[item];
        const float3 p = make_float3(xyz[3 * (size_t)orig], xyz[3 * (size_t)orig + 1], xyz[3 * (size_t)orig + 2]);
        const int want = min(mean_k + 1, n);
        if (threadIdx.x == 0) {
            s_prefix = 0u;
            s_rank = want; // rank (1-based) of the wanted element among those matching the prefix so far
        }
        __syncthreads();
        const int shifts[3] = {20, 9, 0}, widths[3] = {11, 11, 9};
        unsigned int mask = 0u;
        for (int pass = 0; pass < 3; pass++) {
            for (int b = threadIdx.x; b < 2048; b += 256) hist[b] = 0;
            __syncthreads();
            const unsigned int prefix = s_prefix;
            for (int q = threadIdx.x; q < n; q += 256) {
                const float4 o = sxyz[q];
                const unsigned int u = __float_as_uint(fdist2(p.x, p.y, p.z, o.x, o.y, o.z));
                if ((u & mask) == prefix) atomicAdd(&hist[(u >> shifts[pass]) & ((1u << widths[pass]) - 1u)], 1);
            }
            __syncthreads();
            if (threadIdx.x == 0) {
                int rank = s_rank, b = 0;
                for (; b < (1 << widths[pass]) - 1; b++) {
                    if (hist[b] >= rank) break;
                    rank -= hist[b];
                }
                s_rank = rank;
                s_prefix = prefix | ((unsigned int)b << shifts[pass]);
            }
            mask |= ((1u << widths[pass]) - 1u) << shifts[pass];
            __syncthreads();
        }
        const float tau = __uint_as_float(s_prefix);
        double sum = 0.0;
        int less = 0;
        for (int q = threadIdx.x; q < n; q += 256) {
            const float4 o = sxyz[q];
            const float d2 = fdist2(p.x, p.y, p.z, o.x, o.y, o.z);
            if (d2 < tau) {
                sum += (double)sqrtf(d2);
                less++;
            }
        }
        s_sum[threadIdx.x] = sum;
        s_less[threadIdx.x] = less;
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) {
            if ((int)threadIdx.x < o) {
                s_sum[threadIdx.x] += s_sum[threadIdx.x + o];
                s_less[threadIdx.x] += s_less[threadIdx.x + o];
            }
            __syncthreads();
        }
        if (threadIdx.x == 0)
            dist_orig[orig] = (float)((s_sum[0] + (double)(want - s_less[0]) * (double)sqrtf(tau)) / mean_k);
        __syncthreads();
    }
}

__global__ void k_keep_flags(const float *__restrict__ dist, int64_t n, double thr, unsigned int *__restrict__ flag) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) flag[i] = !((double)dist[i] > thr); // removed: distances[i] > distance_threshold
}
__global__ void k_compact_kept(const float *__restrict__ xyz, const unsigned int *__restrict__ flag, const unsigned int *__restrict__ pos,
                               int64_t n, float *__restrict__ fxyz, int32_t *__restrict__ kept_index) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || !flag[i]) return;
    const unsigned int o = pos[i];
    fxyz[3 * (size_t)o] = xyz[3 * i];
    fxyz[3 * (size_t)o + 1] = xyz[3 * i + 1];
    fxyz[3 * (size_t)o + 2] = xyz[3 * i + 2];
    kept_index[o] = (int32_t)i;
}

// ---- eigen33 / computeRoots of PCL's common/impl/eigen.hpp (smallest eigenvalue and its vector), in double
__device__ __forceinline__ void pcl_roots2(double b, double c, double *r) {
    r[0] = 0.0;
    double d = b * b - 4.0 * c;
    if (d < 0.0) d = 0.0;
    const double sd = sqrt(d);
    r[2] = 0.5 * (b + sd);
    r[1] = 0.5 * (b - sd);
}
__device__ void pcl_plane_from_cov(const double *cov, double *nrm, double *curvature) {
    double scale = 0.0;
    for (int i = 0; i < 9; i++) scale = fmax(scale, fabs(cov[i]));
    if (scale <= 2.2250738585072014e-308) scale = 1.0;
    double m[9];
    for (int i = 0; i < 9; i++) m[i] = cov[i] / scale;
    double r[3];
    const double c0 = m[0] * m[4] * m[8] + 2.0 * m[1] * m[2] * m[5] - m[0] * m[5] * m[5] - m[4] * m[2] * m[2] - m[8] * m[1] * m[1];
    const double c1 = m[0] * m[4] - m[1] * m[1] + m[0] * m[8] - m[2] * m[2] + m[4] * m[8] - m[5] * m[5];
    const double c2 = m[0] + m[4] + m[8];
    if (fabs(c0) < 2.220446049250313e-16) {
        pcl_roots2(c2, c1, r);
    } else {
        const double s_inv3 = 1.0 / 3.0, s_sqrt3 = sqrt(3.0);
        const double c2_over_3 = c2 * s_inv3;
        double a_over_3 = (c1 - c2 * c2_over_3) * s_inv3;
        if (a_over_3 > 0.0) a_over_3 = 0.0;
        const double half_b = 0.5 * (c0 + c2_over_3 * (2.0 * c2_over_3 * c2_over_3 - c1));
        double q = half_b * half_b + a_over_3 * a_over_3 * a_over_3;
        if (q > 0.0) q = 0.0;
        const double rho = sqrt(-a_over_3);
        const double theta = atan2(sqrt(-q), half_b) * s_inv3;
        const double ct = cos(theta), st = sin(theta);
        r[0] = c2_over_3 + 2.0 * rho * ct;
        r[1] = c2_over_3 - rho * (ct + s_sqrt3 * st);
        r[2] = c2_over_3 - rho * (ct - s_sqrt3 * st);
        if (r[0] >= r[1]) { const double t = r[0]; r[0] = r[1]; r[1] = t; }
        if (r[1] >= r[2]) {
            const double t = r[1]; r[1] = r[2]; r[2] = t;
            if (r[0] >= r[1]) { const double u = r[0]; r[0] = r[1]; r[1] = u; }
        }
        if (r[0] <= 0.0) pcl_roots2(c2, c1, r);
    }
    const double ev = r[0] * scale;
    m[0] -= r[0];
    m[4] -= r[0];
    m[8] -= r[0];
    const double v1[3] = {m[1] * m[5] - m[2] * m[4], m[2] * m[3] - m[0] * m[5], m[0] * m[4] - m[1] * m[3]};
    const double v2[3] = {m[1] * m[8] - m[2] * m[7], m[2] * m[6] - m[0] * m[8], m[0] * m[7] - m[1] * m[6]};
    const double v3[3] = {m[4] * m[8] - m[5] * m[7], m[5] * m[6] - m[3] * m[8], m[3] * m[7] - m[4] * m[6]};
    const double l1 = v1[0] * v1[0] + v1[1] * v1[1] + v1[2] * v1[2];
    const double l2 = v2[0] * v2[0] + v2[1] * v2[1] + v2[2] * v2[2];
    const double l3 = v3[0] * v3[0] + v3[1] * v3[1] + v3[2] * v3[2];
    const double *v = v3;
    double l = l3;
    if (l1 >= l2 && l1 >= l3) { v = v1; l = l1; }
    else if (l2 >= l1 && l2 >= l3) { v = v2; l = l2; }
    const double s = sqrt(l);
    for (int i = 0; i < 3; i++) nrm[i] = v[i] / s;
    const double tr = cov[0] + cov[4] + cov[8];
    *curvature = (tr != 0.0) ? fabs(ev / tr) : 0.0;
}

// normal_3d.h computePointNormal over the radius neighbourhood + flipNormalTowardsViewpoint(origin) + the turn toward
// CamCenter (CCloudOptimization.cpp:114-121); thread = sorted point, result stored at the point's original position
__global__ __launch_bounds__(256) void k_cloud_normals(const float4 *__restrict__ sxyz, const unsigned long long *__restrict__ keys, int n,
                                                        FGrid g, float r2, float cx, float cy, float cz, float4 *__restrict__ normals) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const float4 p = sxyz[j];
    int rs[9], re[9];
    ranges9(keys, n, g, p.x, p.y, p.z, rs, re);
    double a0 = 0, a1 = 0, a2 = 0, a3 = 0, a4 = 0, a5 = 0, a6 = 0, a7 = 0, a8 = 0;
    int cnt = 0;
#pragma unroll
    for (int r = 0; r < 9; r++)
        for (int q = rs[r]; q < re[r]; q++) {
            const float4 o = sxyz[q];
            if (!(fdist2(p.x, p.y, p.z, o.x, o.y, o.z) < r2)) continue;
            const double x = o.x, y = o.y, z = o.z;
            a0 += x * x; a1 += x * y; a2 += x * z; a3 += y * y; a4 += y * z; a5 += z * z;
            a6 += x; a7 += y; a8 += z;
            cnt++;
        }
    float4 out;
    if (cnt < 3) {
        out.x = out.y = out.z = out.w = __uint_as_float(0x7fc00000u);
    } else {
        const double inv = (double)cnt;
        a0 /= inv; a1 /= inv; a2 /= inv; a3 /= inv; a4 /= inv; a5 /= inv; a6 /= inv; a7 /= inv; a8 /= inv;
        double cov[9];
        cov[0] = a0 - a6 * a6;
        cov[1] = cov[3] = a1 - a6 * a7;
        cov[2] = cov[6] = a2 - a6 * a8;
        cov[4] = a3 - a7 * a7;
        cov[5] = cov[7] = a4 - a7 * a8;
        cov[8] = a5 - a8 * a8;
        double nv[3], curv;
        pcl_plane_from_cov(cov, nv, &curv);
        if ((0.0 - (double)p.x) * nv[0] + (0.0 - (double)p.y) * nv[1] + (0.0 - (double)p.z) * nv[2] < 0.0) {
            nv[0] = -nv[0]; nv[1] = -nv[1]; nv[2] = -nv[2];
        }
        float nx = (float)nv[0], ny = (float)nv[1], nz = (float)nv[2];
        const float vx = cx - p.x, vy = cy - p.y, vz = cz - p.z;
        if (nx * vx + ny * vy + nz * vz < 0.0f) { nx = -nx; ny = -ny; nz = -nz; } // :116-120
        out = make_float4(nx, ny, nz, (float)curv);
    }
    normals[__float_as_uint(p.w)] = out;
}

__global__ void k_pack_filtered16(const double *__restrict__ xyz, const uint8_t *__restrict__ bgr, const int32_t *__restrict__ kept, int64_t m,
                                  uint4 *__restrict__ dst) {
    const int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= m) return;
    const size_t i = (size_t)kept[o];
    uint4 v;
    v.x = __float_as_uint((float)xyz[3 * i]);
    v.y = __float_as_uint((float)xyz[3 * i + 1]);
    v.z = __float_as_uint((float)xyz[3 * i + 2]);
    v.w = (uint32_t)bgr[3 * i] | ((uint32_t)bgr[3 * i + 1] << 8) | ((uint32_t)bgr[3 * i + 2] << 16);
    dst[o] = v;
}
__global__ void k_f64_to_f32x3(const double *__restrict__ src, int64_t n3, float *__restrict__ dst) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n3) dst[i] = (float)src[i]; // InsertPoint: pcl::PointXYZ(double, double, double) -> float (:61)
}

// order-preserving float <-> uint map for atomicMin / atomicMax
__host__ __device__ inline unsigned int float_to_ord(float f) {
    unsigned int u;
    memcpy(&u, &f, 4);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__host__ __device__ inline float ord_to_float(unsigned int o) {
    const unsigned int u = (o & 0x80000000u) ? (o & 0x7fffffffu) : ~o;
    float f;
    memcpy(&f, &u, 4);
    return f;
}
__global__ void k_bbox(const float *__restrict__ xyz, int64_t n, unsigned int *__restrict__ bb) {
    unsigned int lo[3] = {0xffffffffu, 0xffffffffu, 0xffffffffu}, hi[3] = {0u, 0u, 0u};
    unsigned int nfin = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        if (!(isfinite(xyz[3 * i]) && isfinite(xyz[3 * i + 1]) && isfinite(xyz[3 * i + 2]))) continue;
        nfin++;
        for (int a = 0; a < 3; a++) {
            const unsigned int o = float_to_ord(xyz[3 * i + a]);
            lo[a] = min(lo[a], o);
            hi[a] = max(hi[a], o);
        }
    }
    for (int o = 32; o > 0; o >>= 1) nfin += (unsigned int)__shfl_xor((int)nfin, o);
    if ((threadIdx.x & 63) == 0 && nfin) atomicAdd(&bb[6], nfin);
    for (int a = 0; a < 3; a++) {
        for (int o = 32; o > 0; o >>= 1) {
            lo[a] = min(lo[a], (unsigned int)__shfl_xor((int)lo[a], o));
            hi[a] = max(hi[a], (unsigned int)__shfl_xor((int)hi[a], o));
        }
        if ((threadIdx.x & 63) == 0) {
            atomicMin(&bb[a], lo[a]);
            atomicMax(&bb[3 + a], hi[a]);
        }
    }
}

// ------------------------------------------------------------------------------------------------ host side
namespace {
struct DevBuf {
    void *p = nullptr;
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(DevBuf &&o) noexcept {
        if (this != &o) {
            if (p) (void)hipFree(p);
            p = o.p;
            o.p = nullptr;
        }
        return *this;
    }
    ~DevBuf() {
        if (p) (void)hipFree(p);
    }
    template <typename T>
    T *get(size_t n) {
        if (hipMalloc(&p, n * sizeof(T) + 64) != hipSuccess) p = nullptr;
        return (T *)p;
    }
};

// robust grid: extents from the 1 % .. 99 % quantiles of a sample (far outliers must not set the cell size)
bool sample_extent(const float *d_xyz, int64_t n, hipStream_t st, float lo[3], float hi[3], float full_lo[3], float full_hi[3]) {
    const int S = (int)std::min<int64_t>(n, 8192);
    std::vector<float> h((size_t)3 * S);
    const int64_t step = n / S;
    for (int s = 0; s < S; s++)
        if (hipMemcpyAsync(&h[3 * (size_t)s], d_xyz + 3 * (size_t)(s * step), 3 * sizeof(float), hipMemcpyDeviceToHost, st) != hipSuccess) return false;
    if (hipStreamSynchronize(st) != hipSuccess) return false;
    for (int a = 0; a < 3; a++) {
        std::vector<float> v;
        for (int s = 0; s < S; s++)
            if (std::isfinite(h[3 * (size_t)s]) && std::isfinite(h[3 * (size_t)s + 1]) && std::isfinite(h[3 * (size_t)s + 2])) v.push_back(h[3 * (size_t)s + a]);
        if (v.empty()) v.push_back(0.0f);
        std::sort(v.begin(), v.end());
        lo[a] = v[(size_t)(0.01 * (v.size() - 1))];
        hi[a] = v[(size_t)(0.99 * (v.size() - 1))];
        full_lo[a] = v.front();
        full_hi[a] = v.back();
    }
    return true;
}
} // namespace

struct FilterGridDev {
    DevBuf b_keys, b_keys2, b_vals, b_vals2, b_sxyz, b_tmp;
    unsigned long long *keys = nullptr;
    unsigned int *vals = nullptr; // original index of every sorted point
    float4 *sxyz = nullptr;
    FGrid g{};
    FilterGridDev() = default;
    FilterGridDev &operator=(FilterGridDev &&o) noexcept = default;
};

// sorts the n points of d_xyz by the key of a grid with cell edge h covering their bounding box
static int build_grid(const float *d_xyz, int64_t n, float h, const float bb_lo[3], const float bb_hi[3], hipStream_t st, FilterGridDev &G) {
    G.g.ox = bb_lo[0];
    G.g.oy = bb_lo[1];
    G.g.oz = bb_lo[2];
    G.g.inv_h = 1.0f / h;
    auto dim = [&](int a) { return (int)std::min<double>(1 << 20, std::max<double>(1.0, floor((double)(bb_hi[a] - bb_lo[a]) / h) + 1.0)); };
    G.g.nx = dim(0);
    G.g.ny = dim(1);
    G.g.nz = dim(2);
    unsigned long long *k1 = G.b_keys.get<unsigned long long>((size_t)n), *k2 = G.b_keys2.get<unsigned long long>((size_t)n);
    unsigned int *v1 = G.b_vals.get<unsigned int>((size_t)n), *v2 = G.b_vals2.get<unsigned int>((size_t)n);
    G.sxyz = G.b_sxyz.get<float4>((size_t)n);
    if (!k1 || !k2 || !v1 || !v2 || !G.sxyz) return RSM_E_NOMEM;
    const unsigned blocks = (unsigned)((n + 255) / 256);
    hipLaunchKernelGGL(k_cell_keys, dim3(blocks), dim3(256), 0, st, d_xyz, n, G.g, k1, v1);
    size_t tmp_bytes = 0;
    if (rocprim::radix_sort_pairs(nullptr, tmp_bytes, k1, k2, v1, v2, (size_t)n, 0, 64, st) != hipSuccess) return RSM_E_HIP;
    void *tmp = G.b_tmp.get<uint8_t>(tmp_bytes);
    if (!tmp) return RSM_E_NOMEM;
    if (rocprim::radix_sort_pairs(tmp, tmp_bytes, k1, k2, v1, v2, (size_t)n, 0, 64, st) != hipSuccess) return RSM_E_HIP;
    hipLaunchKernelGGL(k_gather_sorted, dim3(blocks), dim3(256), 0, st, d_xyz, v2, n, G.sxyz);
    G.keys = k2;
    G.vals = v2;
    return RSM_OK;
}

// d_xyz: n x 3 float (device).  Outputs (device): kept_index [n] (first *n_kept valid), fxyz [3n], normals [n] float4.
int filter_cloud_device(const float *d_xyz, int64_t n, int mean_k, double std_mul, double normal_radius, const float cam_center[3],
                        int32_t *d_kept_index, float *d_fxyz, float4 *d_normals, int64_t *n_kept, double stats[4], hipStream_t st) {
    *n_kept = 0;
    if (n <= 0) return RSM_OK;
    if (n >= (1ll << 31) || mean_k < 1) return RSM_E_INVALID;
    float lo[3], hi[3], flo[3], fhi[3];
    if (!sample_extent(d_xyz, n, st, lo, hi, flo, fhi)) return RSM_E_HIP;
    // exact bounding box (the sample's extremes are not the cloud's)
    DevBuf b_dist, b_redo, b_cnt, b_flag, b_pos, b_tmp, b_bb;
    unsigned int *d_bb = b_bb.get<unsigned int>(8);
    if (!d_bb) return RSM_E_NOMEM;
    int64_t nv = 0; // finite points: only they take part in the searches (PCL: the k-d tree skips the others)
    {
        const unsigned int init[7] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0u, 0u, 0u, 0u};
        unsigned int bb[7];
        if (hipMemcpyAsync(d_bb, init, sizeof init, hipMemcpyHostToDevice, st) != hipSuccess) return RSM_E_HIP;
        hipLaunchKernelGGL(k_bbox, dim3((unsigned)std::min<int64_t>((n + 255) / 256, 2048)), dim3(256), 0, st, d_xyz, n, d_bb);
        if (hipMemcpyAsync(bb, d_bb, sizeof bb, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return RSM_E_HIP;
        nv = bb[6];
        for (int a = 0; a < 3; a++) {
            flo[a] = nv ? ord_to_float(bb[a]) : 0.0f;
            fhi[a] = nv ? ord_to_float(bb[3 + a]) : 0.0f;
        }
    }
    // first cell edge: ~sqrt(k + 1) point spacings of a surface patch whose area is the product of the two largest
    // robust extents; queries that cannot be decided inside their 27 cells (fewer than k + 1 points within h: thick or
    // sparse parts of the cloud, patch corners, isolated points) are retried on a grid with twice the edge, the few
    // that remain after KNN_LEVELS grids are searched exhaustively
    double e[3] = {(double)hi[0] - lo[0], (double)hi[1] - lo[1], (double)hi[2] - lo[2]};
    std::sort(e, e + 3);
    const double area = std::max(e[2] * e[1], 1e-12), spacing = sqrt(area / (0.98 * 0.98 * 0.98 * (double)std::max<int64_t>(nv, 1)));
    float h = (float)(spacing * sqrt((double)(mean_k + 1)));
    if (!(h > 0.0f) || !std::isfinite(h)) h = 1.0f;
    float *d_dist = b_dist.get<float>((size_t)n);
    DevBuf b_redo2;
    unsigned int *d_redo = b_redo.get<unsigned int>((size_t)n), *d_redo2 = b_redo2.get<unsigned int>((size_t)n);
    int *d_cnt = b_cnt.get<int>(4);
    if (!d_dist || !d_redo || !d_redo2 || !d_cnt) return RSM_E_NOMEM;
    const unsigned blocks = (unsigned)((n + 255) / 256);
    const int KNN_LEVELS = 5;
    if (hipMemsetAsync(d_dist, 0, sizeof(float) * (size_t)n, st) != hipSuccess) return RSM_E_HIP; // non-finite points: distance 0, as PCL
    int nq = (int)nv, redo_n = 0;
    const unsigned int *queries = nullptr;
    int s = RSM_OK;
    {
        FilterGridDev G; // the last grid also serves the exhaustive search (any ordering of the points does)
        for (int level = 0; level < KNN_LEVELS && nq > 0; level++, h *= 2.0f) {
            G = FilterGridDev();
            s = build_grid(d_xyz, n, h, flo, fhi, st, G);
            if (s != RSM_OK) return s;
            if (level == 0) queries = G.vals; // every point, in grid order (coherent waves)
            if (hipMemsetAsync(d_cnt, 0, sizeof(int) * 4, st) != hipSuccess) return RSM_E_HIP;
            unsigned int *out_list = (level & 1) ? d_redo2 : d_redo;
            hipLaunchKernelGGL(k_sor_knn, dim3((unsigned)((nq + 3) / 4)), dim3(256), 0, st, d_xyz, G.sxyz, G.keys, (int)nv, G.g, h * h, mean_k,
                               queries, nq, d_dist, out_list, d_cnt);
            if (hipMemcpyAsync(&redo_n, d_cnt, sizeof(int), hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess)
                return RSM_E_HIP;
            queries = out_list;
            nq = redo_n;
            if (nq <= 64) break; // cheaper to finish exhaustively than to sort again
        }
        if (nq > 0) hipLaunchKernelGGL(k_sor_knn_all, dim3((unsigned)std::min(nq, 4096)), dim3(256), 0, st, d_xyz, G.sxyz, (int)nv, mean_k, d_dist, queries, d_cnt);
        if (hipStreamSynchronize(st) != hipSuccess) return RSM_E_HIP;
    }
    redo_n = nq;
    // mean / stddev exactly as PCL: a sequential host loop over the per-point distances in point order
    std::vector<float> hd((size_t)n);
    if (hipMemcpyAsync(hd.data(), d_dist, sizeof(float) * (size_t)n, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess)
        return RSM_E_HIP;
    double sum = 0.0, sq_sum = 0.0;
    for (int64_t i = 0; i < n; i++) {
        sum += hd[(size_t)i];
        sq_sum += hd[(size_t)i] * hd[(size_t)i]; // float * float, as in PCL
    }
    const double mean = sum / (double)nv; // valid_distances
    const double variance = (sq_sum - sum * sum / (double)nv) / ((double)nv - 1);
    const double stddev = sqrt(variance), thr = mean + std_mul * stddev;
    if (stats) {
        stats[0] = mean;
        stats[1] = stddev;
        stats[2] = thr;
        stats[3] = (double)redo_n;
    }
    unsigned int *d_flag = b_flag.get<unsigned int>((size_t)n), *d_pos = b_pos.get<unsigned int>((size_t)n);
    if (!d_flag || !d_pos) return RSM_E_NOMEM;
    hipLaunchKernelGGL(k_keep_flags, dim3(blocks), dim3(256), 0, st, d_dist, n, thr, d_flag);
    size_t tb = 0;
    if (rocprim::exclusive_scan(nullptr, tb, d_flag, d_pos, 0u, (size_t)n, rocprim::plus<unsigned int>(), st) != hipSuccess) return RSM_E_HIP;
    void *tp = b_tmp.get<uint8_t>(tb);
    if (!tp) return RSM_E_NOMEM;
    if (rocprim::exclusive_scan(tp, tb, d_flag, d_pos, 0u, (size_t)n, rocprim::plus<unsigned int>(), st) != hipSuccess) return RSM_E_HIP;
    hipLaunchKernelGGL(k_compact_kept, dim3(blocks), dim3(256), 0, st, d_xyz, d_flag, d_pos, n, d_fxyz, d_kept_index);
    unsigned int last_pos = 0, last_flag = 0;
    if (hipMemcpyAsync(&last_pos, d_pos + (n - 1), 4, hipMemcpyDeviceToHost, st) != hipSuccess ||
        hipMemcpyAsync(&last_flag, d_flag + (n - 1), 4, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess)
        return RSM_E_HIP;
    const int64_t m = (int64_t)last_pos + last_flag;
    *n_kept = m;
    if (m == 0 || !d_normals) return RSM_OK;
    // normals of the filtered cloud: grid with cell edge = search radius; the non-finite points (all kept: distance 0)
    // sort behind the finite ones and keep the NaN normal the buffer is filled with
    FilterGridDev G2;
    s = build_grid(d_fxyz, m, (float)normal_radius, flo, fhi, st, G2);
    if (s != RSM_OK) return s;
    const float r2 = (float)(normal_radius * normal_radius);
    const int64_t mv = m - (n - nv);
    if (hipMemsetD32Async((hipDeviceptr_t)d_normals, 0x7fc00000, (size_t)4 * m, st) != hipSuccess) return RSM_E_HIP;
    if (mv > 0)
        hipLaunchKernelGGL(k_cloud_normals, dim3((unsigned)((mv + 255) / 256)), dim3(256), 0, st, G2.sxyz, G2.keys, (int)mv, G2.g, r2,
                           cam_center[0], cam_center[1], cam_center[2], d_normals);
    if (hipStreamSynchronize(st) != hipSuccess || hipGetLastError() != hipSuccess) return RSM_E_HIP;
    return RSM_OK;
}

void launch_f64_to_f32x3(const double *src, int64_t n, float *dst, hipStream_t st) {
    if (n > 0) hipLaunchKernelGGL(k_f64_to_f32x3, dim3((unsigned)((3 * n + 255) / 256)), dim3(256), 0, st, src, 3 * n, dst);
}
void launch_pack_filtered16(const double *xyz, const uint8_t *bgr, const int32_t *kept, int64_t m, void *dst, hipStream_t st) {
    if (m > 0) hipLaunchKernelGGL(k_pack_filtered16, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, st, xyz, bgr, kept, m, (uint4 *)dst);
}
