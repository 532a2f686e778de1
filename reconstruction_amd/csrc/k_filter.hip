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

#define FILTER_MAX_CELLS (1 << 25) // most cells a per-cell table is made for (256 MB); beyond that: per-row table / binary search on the keys
// ... for an n-point cloud: a table of far more cells than points is mostly empty, and its bytes are pinned in the context's
// grow-only arena (a 1-point cloud must not cost 256 MB)
static inline size_t filter_max_cells(int64_t n) {
    return (size_t)std::min<int64_t>(FILTER_MAX_CELLS, std::max<int64_t>(1 << 16, 8 * n));
}
// Grid in KEY order: axis "x" is the fastest digit of the cell key, and it is the WORLD axis with the most cells (p0) --
// a depth map is a sheet in a deep box, so the rows of cells along its depth hold a handful of points each and the
// per-row table + short search inside the row (table kind 2) stays cheap when the cells are too many for a table.
struct FGrid {
    float ox, oy, oz, inv_h;
    int nx, ny, nz;
    int p0, p1, p2; // world axis (0 = x, 1 = y, 2 = z) of key axis x, y, z
};
__device__ __forceinline__ float pick_axis(int a, float x, float y, float z) { return a == 0 ? x : (a == 1 ? y : z); }

__device__ __forceinline__ float fdist2(float ax, float ay, float az, float bx, float by, float bz) {
    const float dx = ax - bx, dy = ay - by, dz = az - bz;
    return (dx * dx + dy * dy) + dz * dz;
}
// sqrtf, correctly rounded, for the k-nearest sums (PCL adds sqrt of the float32 squared distances): the compiler's own sequence --
// v_sqrt_f32 (1 ulp), the two neighbouring floats, an fma residual each, two selects -- WITHOUT its input scaling for
// denormals and its zero / infinity test (8 of 17 instructions): the same bits for every x in [2^-96, FLT_MAX] and for 0
// (rsm_stage_sqrt_check holds it against sqrtf on ALL of them); anything else takes sqrtf.  The pixel-window search spends its
// second pass in this function with a tenth of its lanes active.
__device__ __forceinline__ float sqrtf_rn_core(float x) { // x in [2^-96, FLT_MAX] or 0
    float s = __builtin_amdgcn_sqrtf(x);
    const float s_dn = __uint_as_float(__float_as_uint(s) - 1u), s_up = __uint_as_float(__float_as_uint(s) + 1u);
    const float r_dn = __builtin_fmaf(-s_dn, s, x), r_up = __builtin_fmaf(-s_up, s, x);
    s = (r_dn <= 0.0f) ? s_dn : s;
    s = (r_up > 0.0f) ? s_up : s;
    return s;
}
__device__ __forceinline__ float sqrtf_rn(float x) {
    if (__builtin_expect(!(x >= 0x1p-96f || x == 0.0f), 0)) return sqrtf(x); // denormal-range inputs (and NaN / negative): the general sequence
    return sqrtf_rn_core(x);
}
// test entry (rsm_stage_sqrt_check): sqrtf_rn against sqrtf on the floats with bit patterns first .. first + n - 1
__global__ void k_sqrt_check(unsigned int first, long long n, unsigned long long *mismatches) {
    const long long i0 = (long long)blockIdx.x * blockDim.x + threadIdx.x, stride = (long long)gridDim.x * blockDim.x;
    unsigned int bad = 0;
    for (long long i = i0; i < n; i += stride) {
        const float x = __uint_as_float(first + (unsigned int)i);
        bad += __float_as_uint(sqrtf_rn(x)) != __float_as_uint(sqrtf(x));
    }
    if (bad) atomicAdd(mismatches, (unsigned long long)bad);
}
void launch_sqrt_check(unsigned int first, long long n, unsigned long long *mismatches, hipStream_t st) {
    if (n > 0) hipLaunchKernelGGL(k_sqrt_check, dim3(8192), dim3(256), 0, st, first, n, mismatches);
}

__device__ __forceinline__ int cell_of(float v, float o, float inv_h, int n) {
    const int c = (int)floorf((v - o) * inv_h);
    return min(max(c, 0), n - 1);
}
// cell of a world point, in key order
__device__ __forceinline__ void grid_cell(const FGrid &g, float x, float y, float z, int &ix, int &iy, int &iz) {
    ix = cell_of(pick_axis(g.p0, x, y, z), g.ox, g.inv_h, g.nx);
    iy = cell_of(pick_axis(g.p1, x, y, z), g.oy, g.inv_h, g.ny);
    iz = cell_of(pick_axis(g.p2, x, y, z), g.oz, g.inv_h, g.nz);
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
    int ix, iy, iz;
    grid_cell(g, x, y, z, ix, iy, iz);
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

// cell table: (first, one-past-last) sorted index of every cell's points; (0, 0) for an empty cell
// div = 1: per cell; div = nx: per (y, z) row of cells (a deep or thick cloud has too many cells for a table of them)
__global__ void k_cell_table(const unsigned long long *__restrict__ keys, int nv, unsigned long long div, int2 *__restrict__ table) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nv) return;
    const unsigned long long k = keys[i] / div;
    if (i == 0 || keys[i - 1] / div != k) table[k].x = i;
    if (i == nv - 1 || keys[i + 1] / div != k) table[k].y = i + 1;
}
__device__ __forceinline__ int lower_bound_key_in(const unsigned long long *__restrict__ keys, int lo, int hi, unsigned long long k) {
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (keys[mid] < k) lo = mid + 1;
        else hi = mid;
    }
    return lo;
}

// the 9 contiguous ranges of the sorted array that hold the 27 cells around p (binary search on the keys; used by the
// normals and by the k-nearest search when the grid has too many cells for a table)
__device__ __forceinline__ void ranges9(const unsigned long long *__restrict__ keys, int n, const FGrid &g, float px, float py,
                                        float pz, int (&rs)[9], int (&re)[9]) {
    int ix, iy, iz;
    grid_cell(g, px, py, pz, ix, iy, iz);
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

#define KNN_C 32    // registers per lane for the candidates within h
#define KNN_CAP (64 * KNN_C)
#define KNN_B 16    // candidate loads in flight per lane
// The selection a wave runs for one query over M candidates `cand(c)` (squared float32 distances, NaN = no candidate): the (k+1)-th
// smallest among those <= h2 and the exact double sum of the square roots below it.  Returns false -- nothing written but
// `undecided` -- when fewer than `want` candidates lie within h2.  Shared by the grid search (k_sor_knn: the 27 cells of a query)
// and the wave form of the pixel-window search (k_sor_window_wave: a window of the lattice copy).
template <int C_ = KNN_C, class Cand>
__device__ __forceinline__ bool wave_knn_core(const Cand &cand, int M, float h2, int want, float *mine, int lane, unsigned int *undecided_slot,
                                              float &tau_out, double &sum_out, int &less_out, int K_pre = -1) {
    // K_pre >= 0: `mine` already holds the candidates within h2 (the first CAP_ of the K_pre there are, in a fixed order): the
    // workgroup form's four waves have made the first pass together
    constexpr int CAP_ = 64 * C_; // the list's capacity: C_ registers per lane in the selection
    const float inf = __uint_as_float(0x7f800000u);
    // Only candidates within h of the query matter (a query is decided here iff k + 1 of them exist): the squared
    // distances are computed in chunks of KNN_B loads per lane in flight, those <= h^2 are compacted into the wave's
    // LDS list (about a third of the 27 cells' points for a surface, a sixth for a volume) and read back into registers,
    // lane l holding entries l, l + 64, ...; the asm fence keeps the compiler from re-deriving them from memory in
    // every bisection step (the first version did: 120 ns per query).
    if (M < want) { // fewer points in the 27 cells than neighbours wanted: undecidable without looking at them
        if (lane == 0) *undecided_slot = 1u;
        return false;
    }
    const unsigned long long lt = (1ull << lane) - 1ull;
    int K = K_pre >= 0 ? K_pre : 0; // candidates within h so far (uniform)
    for (int c0 = 0; c0 < M && K_pre < 0; c0 += 64 * KNN_B) { // uniform
        float v[KNN_B];
#pragma unroll
        for (int i = 0; i < KNN_B; i++) {
            const int c = c0 + lane + 64 * i;
            v[i] = (c < M) ? cand(c) : inf;
        }
#pragma unroll
        for (int i = 0; i < KNN_B; i++) {
            const bool keep = v[i] <= h2;
            const unsigned long long mm = __ballot(keep);
            const int pos = K + __popcll(mm & lt);
            if (keep && pos < CAP_) mine[pos] = v[i];
            K += __popcll(mm);
        }
    }
    // (a flag per query, compacted by a scan afterwards: appending to one list through a single counter retires ~88
    // appends per microsecond -- 62 ms for a level that decides nothing)
    if (lane == 0) *undecided_slot = K < want;
    if (K < want) return false; // not decidable among these candidates
    if (K > CAP_ && h2 > 0.0f) {
        // More candidates within h than the list holds (a wide window over a dense part, a too-coarse grid level): two more passes
        // narrow them down instead of a 30-step bisection that re-reads them all in every step -- a histogram of the squared
        // distances over CAP_ equal bins of [0, h2] (the list's LDS), the bin b* holding rank `want`, then the list again with
        // the candidates of bins <= b* only (bin() is monotone: they are exactly the values up to some threshold, the rank's among
        // them).  Falls through to the bisection only if even that is too many (thousands of equal distances).
        int *hist = (int *)mine;
        const float inv_w = (float)CAP_ / h2;
        auto bin_of = [&](float v) { return min((int)(v * inv_w), CAP_ - 1); };
        if (isfinite(inv_w) && inv_w > 0.0f) {
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int i = 0; i < C_; i++) hist[lane + 64 * i] = 0;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            for (int c0 = 0; c0 < M; c0 += 64 * KNN_B) { // uniform
                float v[KNN_B];
#pragma unroll
                for (int i = 0; i < KNN_B; i++) {
                    const int c = c0 + lane + 64 * i;
                    v[i] = (c < M) ? cand(c) : inf;
                }
#pragma unroll
                for (int i = 0; i < KNN_B; i++)
                    if (v[i] <= h2) atomicAdd(&hist[bin_of(v[i])], 1);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            // lane l owns bins [32 l, 32 l + 32): its total, the wave's exclusive prefix, then the bin inside the owning lane
            int mine_sum = 0;
#pragma unroll
            for (int i = 0; i < C_; i++) mine_sum += hist[lane * C_ + i];
            int incl = mine_sum;
            for (int o = 1; o < 64; o <<= 1) {
                const int t = __shfl_up(incl, o);
                if (lane >= o) incl += t;
            }
            const unsigned long long has = __ballot(incl >= want);
            const int owner = __builtin_ctzll(has); // (K >= want: some lane reaches it)
            int run = __shfl(incl - mine_sum, owner), bstar = owner * C_, upto = 0;
            for (int i = 0; i < C_; i++) { // uniform: every lane reads the owner's bins
                const int hcount = hist[owner * C_ + i];
                run += hcount;
                if (run >= want) {
                    bstar = owner * C_ + i;
                    upto = run;
                    break;
                }
            }
            if (upto >= want && upto <= CAP_) {
                __builtin_amdgcn_wave_barrier();
                K = 0;
                for (int c0 = 0; c0 < M; c0 += 64 * KNN_B) { // uniform
                    float v[KNN_B];
#pragma unroll
                    for (int i = 0; i < KNN_B; i++) {
                        const int c = c0 + lane + 64 * i;
                        v[i] = (c < M) ? cand(c) : inf;
                    }
#pragma unroll
                    for (int i = 0; i < KNN_B; i++) {
                        const bool keep = v[i] <= h2 && bin_of(v[i]) <= bstar;
                        const unsigned long long mm = __ballot(keep);
                        const int pos = K + __popcll(mm & lt);
                        if (keep && pos < CAP_) mine[pos] = v[i];
                        K += __popcll(mm);
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            } else {
                K = CAP_ + 1; // (the list's LDS holds the histogram now: the bisection below reads the candidates themselves)
            }
        }
    }
    float tau;
    double sum = 0.0;
    int less = 0;
    if (K <= CAP_) {
        __builtin_amdgcn_wave_barrier();
        float d2[C_];
        const int nreg = (K + 63) >> 6; // wave-uniform
#pragma unroll
        for (int i = 0; i < C_; i++) {
            d2[i] = inf;
            if (i < nreg && lane + 64 * i < K) d2[i] = mine[lane + 64 * i];
        }
#pragma unroll
        for (int i = 0; i < C_; i++) asm volatile("" : "+v"(d2[i])); // materialise: the values live in registers from here on (the fences AFTER
                                                                        // the last read: one right behind each read makes it a round trip of its own)
        auto count4 = [&](float t, int i0) {
            int cnt = 0;
#pragma unroll
            for (int i = 0; i < 4; i++) cnt += __popcll(__ballot(d2[i0 + i] <= t));
            return cnt;
        };
        auto count_le = [&](float t) { // registers past nreg hold +inf; whole groups of 4 are skipped (uniform)
            int cnt = count4(t, 0);
#pragma unroll
            for (int g4 = 1; g4 < C_ / 4; g4++)
                if (nreg > 4 * g4) cnt += count4(t, 4 * g4);
            return cnt;
        };
        // smallest bit pattern b with count_le(b) >= want = the want-th smallest value.  A step that counts EXACTLY want
        // has it bracketed -- it is the largest value <= mid -- which happens after ~log2(K) of the 30 steps unless equal
        // distances straddle the rank (then the bisection runs to the end as before).  (Snapping the bracket to the data
        // in every step -- a wave min / max reduction per step, ~9 steps -- measured slower: 46 instead of 39.5 ms.)
        unsigned int lo = 0u, hi = __float_as_uint(h2);
        while (lo < hi) {
            const unsigned int mid = lo + ((hi - lo) >> 1);
            const int cnt = count_le(__uint_as_float(mid));
            if (cnt == want) {
                const float fm = __uint_as_float(mid);
                float m = 0.0f;
#pragma unroll
                for (int i = 0; i < C_; i++)
                    if (i < nreg && d2[i] <= fm) m = fmaxf(m, d2[i]);
                for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
                lo = hi = __float_as_uint(m);
                break;
            }
            if (cnt >= want) hi = mid;
            else lo = mid + 1;
        }
        tau = __uint_as_float(lo);
#pragma unroll
        for (int i = 0; i < C_; i++)
            if (i < nreg && d2[i] < tau) {
                sum += (double)sqrtf(d2[i]);
                less++;
            }
    } else { // more than CAP_ points within h (a far too coarse first level): bisection straight on the candidates
        auto count_le = [&](float t) {
            int cnt = 0;
            for (int c0 = 0; c0 < M; c0 += 64) { // uniform trip count
                const int c = c0 + lane;
                cnt += __popcll(__ballot(c < M && cand(min(c, M - 1)) <= t));
            }
            return cnt;
        };
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
    tau_out = tau;
    sum_out = sum;
    less_out = less;
    return true;
}

// mean distance to the k nearest neighbours (statistical_outlier_removal.hpp).  One wave per query point: the 27 cells
// around it are 9 contiguous ranges of the sorted array -- their ends come from the cell table (lanes 0..26 load one
// cell each: one round trip) or, without a table, from binary searches; the squared distances to the candidates are
// computed once into registers (lane l holds candidates l, l + 64, ...; the asm fence keeps the compiler from
// re-loading them in every bisection step -- the first version did, 36 VGPRs and 120 ns per query), the (k+1)-th
// smallest is found by bisection on its bit pattern with ballots, the sum by a wave reduction.
// A query with more than 64 * KNN_C candidates re-computes them per bisection step instead.
template <int TABLE> // 1: per-cell table, 2: per-row table + a short binary search inside the row, 0: binary search on all keys
__global__ __launch_bounds__(256) void k_sor_knn(const float *__restrict__ xyz, const float4 *__restrict__ sxyz,
                                                  const unsigned long long *__restrict__ keys, const int2 *__restrict__ table, int n,
                                                  FGrid g, float h2, int mean_k, const unsigned int *__restrict__ queries, int nq,
                                                  float *__restrict__ dist_orig, unsigned int *__restrict__ undecided /* [nq]: 1 = retry on a coarser grid */) {
    __shared__ float s_d2[4][KNN_CAP];
    const int lane = threadIdx.x & 63;
    const int qi = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (qi >= nq) return; // wave-uniform; no workgroup barrier below
    const unsigned int orig = queries[qi];
    const float px = xyz[3 * (size_t)orig], py = xyz[3 * (size_t)orig + 1], pz = xyz[3 * (size_t)orig + 2];
    int ix, iy, iz;
    grid_cell(g, px, py, pz, ix, iy, iz);
    int my_s = 0, my_e = 0; // range t in lane t (t < 9)
    if (TABLE == 2) {
        if (lane < 9) {
            const int yy = iy + lane % 3 - 1, zz = iz + lane / 3 - 1;
            if (yy >= 0 && yy < g.ny && zz >= 0 && zz < g.nz) {
                const unsigned long long row = (unsigned long long)zz * g.ny + yy, base = row * g.nx;
                const int2 se = table[row];
                my_s = lower_bound_key_in(keys, se.x, se.y, base + max(ix - 1, 0));
                my_e = lower_bound_key_in(keys, my_s, se.y, base + min(ix + 1, g.nx - 1) + 1);
            }
        }
    } else if (TABLE == 1) {
        // lane 3 t + j: cell (ix - 1 + j, iy + t % 3 - 1, iz + t / 3 - 1); its points are [x, y) of the sorted array
        int cs = 0, ce = 0;
        if (lane < 27) {
            const int t = lane / 3, xx = ix - 1 + lane % 3, yy = iy + t % 3 - 1, zz = iz + t / 3 - 1;
            if (xx >= 0 && xx < g.nx && yy >= 0 && yy < g.ny && zz >= 0 && zz < g.nz) {
                const int2 se = table[((size_t)zz * g.ny + yy) * g.nx + xx];
                cs = se.x;
                ce = se.y;
            }
        }
        // the three x-adjacent cells hold consecutive keys: first non-empty cell's start .. last non-empty cell's end
        const int s0 = __shfl(cs, 3 * lane), s1 = __shfl(cs, 3 * lane + 1), s2 = __shfl(cs, 3 * lane + 2);
        const int e0 = __shfl(ce, 3 * lane), e1 = __shfl(ce, 3 * lane + 1), e2 = __shfl(ce, 3 * lane + 2);
        if (lane < 9) {
            my_s = e0 > s0 ? s0 : (e1 > s1 ? s1 : s2);
            my_e = e2 > s2 ? e2 : (e1 > s1 ? e1 : (e0 > s0 ? e0 : my_s));
            if (!(e0 > s0 || e1 > s1 || e2 > s2)) my_s = my_e = 0;
        }
    } else if (lane < 9) {
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
        int base = rs[0] - pre[0];
#pragma unroll
        for (int t = 1; t < 9; t++) base = c >= pre[t] ? rs[t] - pre[t] : base;
        const float4 o = sxyz[base + c];
        return fdist2(px, py, pz, o.x, o.y, o.z);
    };
    float tau;
    double sum;
    int less;
    if (!wave_knn_core(cand, M, h2, want, s_d2[threadIdx.x >> 6], lane, &undecided[qi], tau, sum, less)) return;
    if (lane == 0) dist_orig[orig] = (float)((sum + (double)(want - less) * (double)sqrtf(tau)) / mean_k);
}

// =================================================================================================================
// Pixel-window k nearest (rsm_filter_last_cloud, round 5): the cloud of a matched pair is a DEPTH MAP -- point i sits on
// the ray of its pixel, P = iW (x + q03, y + q13, qz) before the rigid R_final / T_final (DisparityToCloud,
// CStereoMatching.cpp:732-749) -- so the pixel lattice itself is the search structure: the k nearest of a point are among
// the points of the (2 WR + 1)^2 pixels around its own, PROVIDED the (k+1)-th smallest distance found there is below the
// distance from P to every ray OUTSIDE that window.  For a ray through pixel offset (a, b), rho = |(a, b)| >= WR + 1:
//     dist(P, ray) = |iW| |D x D'| / |D'| >= |iW| |qz| rho / (|D| + rho),   D = (x + q03, y + q13, qz), D' = D + (a, b, 0)
// (|D x D'|^2 = qz^2 rho^2 + (Dx b - Dy a)^2, |D'| <= |D| + rho), increasing in rho, so with |iW| |qz| = |F2| (the point's
// depth in the camera frame) and |iW| |D| = |P - T|:
//     LB = |F2| (WR + 1) / (|P - T| |qz| / |F2| + WR + 1).
// A query whose (k+1)-th smallest float32 distance stays below LB minus a rounding margin is decided HERE with exactly the
// values the generic search would find (same float32 squared distances, the k smallest by value, the same exact double
// sum); every other query -- near a depth edge, a hole, the mask's border, an outlier: the thick or thin parts of the
// sheet -- is flagged and goes to the grid ladder below.  The selection is per LANE (a query per thread, a 32 x 8 tile of
// pixels per workgroup, the tile and its halo in LDS as float4): pass 1 bins the window's squared distances into 32
// per-thread LDS counters over [0, 4 tau_est] (tau_est = the (k+1)/pi lateral spacings^2 of a fronto-parallel sheet), the
// scan finds the bin holding rank k + 1, pass 2 sums sqrt(d2) of the bins below it in double (exact, order-free) and lists
// the few values of that bin, whose rank-th smallest is tau.  ~120 wave-instructions per query against ~2000 of the
// wave-per-query grid search.
#define WIN_PAD 40  // invalid-pixel border of the lattice copy (>= the largest window radius): no bounds checks in the loops
#define WIN_TX 32
#define WIN_TY 8
#define WIN_NB 32   // distance bins per query over [0, bound^2): the rank's bin holds ~1.5 (k + 1) / (its index) values, a handful
#define WIN_LCAP 24 // values of the rank's bin a query can list (more: undecided -> the ladder); the list reuses the counters' LDS
#ifndef WIN_DRAIN_W
#define WIN_DRAIN_W 4 // waiting values whose square roots are taken together (reads in flight together, sequences interleaved)
#endif
#ifndef WIN_TILE_CH
#define WIN_TILE_CH 9 // candidates per chunk of the tile form (two register buffers of that many 16-byte entries)
#endif
struct WinGeom {
    int gw, gh;         // lattice copy incl. the border: (XR - XL + 1 + 2 PAD) x (YR - YL + 1 + 2 PAD)
    double qz;          // Q[2][3] scaled (.cpp:698)
    double rz[3], T[3]; // third column of R_final (F2 = rz . (P - T)), T_final
};

// lattice copy of the cloud: float4 (x, y, z as InsertPoint keeps them, bits of the point index) per flagged pixel of the
// margin's box, NaN elsewhere (the buffer is pre-filled); block = row, the compaction order of k_cloud<1>
__global__ __launch_bounds__(256) void k_cloud_lattice(const uint8_t *__restrict__ flags, const int64_t *__restrict__ row_offset, int W, int XL, int XR, int YL,
                                                        const double *__restrict__ xyz, int64_t n, int gw, float4 *__restrict__ lat,
                                                        unsigned int *__restrict__ cell_of) {
    __shared__ int s_w[4];
    __shared__ long long s_base;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int y = YL + blockIdx.x;
    if (tid == 0) s_base = (long long)row_offset[blockIdx.x];
    __syncthreads();
    for (int x0 = XL; x0 <= XR; x0 += 256) {
        const int x = x0 + tid;
        const bool f = (x <= XR) && flags[(size_t)y * W + x];
        const unsigned long long b = __ballot(f);
        if (lane == 0) s_w[wid] = __popcll(b);
        __syncthreads();
        if (f) {
            long long pos = s_base + __popcll(b & ((1ull << lane) - 1ull));
            for (int w = 0; w < wid; w++) pos += s_w[w];
            if (pos < n) {
                const float px = (float)xyz[3 * pos], py = (float)xyz[3 * pos + 1], pz = (float)xyz[3 * pos + 2]; // InsertPoint's cast
                const size_t cell = (size_t)(blockIdx.x + WIN_PAD) * gw + (x - XL + WIN_PAD);
                cell_of[pos] = (unsigned int)cell;
                if (isfinite(px) && isfinite(py) && isfinite(pz)) // (non-finite points take no part in the searches)
                    lat[cell] = make_float4(px, py, pz, __uint_as_float((unsigned int)pos));
            }
        }
        __syncthreads();
        if (tid == 0) s_base += s_w[0] + s_w[1] + s_w[2] + s_w[3];
        __syncthreads();
    }
}

// One query of the pixel-window search: P = the query's lattice entry, ld(r, c) = the lattice entry at window row r / column c
// (0 .. 2 WR), col = the thread's private LDS column (stride 256 words: WIN_NB + 1 words of counters, later the list), CH = the
// candidates whose reads are in flight together (a divisor of 2 WR + 1).  Returns whether the query is decided; *out = its mean
// neighbour distance, *lim_out = the bound.
// The pixels a window of "radius" WR has to look at: the bound holds for every ray at a pixel distance rho >= WR + 1, so the DISC
// a^2 + b^2 <= (WR + 1)^2 - 1 suffices -- the corners of the square are farther than the rays the bound already covers.  The rows
// of the window fall into four classes by |b| (<= WR/2, <= 3WR/4, <= 7WR/8, <= WR), each as wide as its widest row needs: 14 %
// fewer candidates than the square at WR = 16 (937 of 1089), the code of a row four times.
constexpr int win_isqrt(int v) {
    int r = 0;
    while ((r + 1) * (r + 1) <= v) r++;
    return r;
}
constexpr int win_halfwidth(int WR, int b) { // widest |a| of the disc's row b, at most WR
    const int a = win_isqrt((WR + 1) * (WR + 1) - 1 - b * b);
    return a < WR ? a : WR;
}
// a row of wd candidates in nch chunks: the first wd % nch one longer
template <int wd, int nch>
struct WinSplit {
    static constexpr int nmax = (wd + nch - 1) / nch;
    static constexpr int size(int j) { return wd / nch + (j < wd % nch ? 1 : 0); }
    static constexpr int off(int j) {
        int o = 0;
        for (int i = 0; i < j; i++) o += size(i);
        return o;
    }
};
template <int WR, int CH, class Loader>
__device__ __forceinline__ bool win_query(const Loader &ld, const float4 P, const WinGeom &g, int mean_k, unsigned int *col, float *out, double *lim_out) {
    constexpr int NC = 2 * WR + 1;
    (void)NC;
    // the bound and its rounding margin, in double (once per query)
    const double dx = (double)P.x - g.T[0], dy = (double)P.y - g.T[1], dz = (double)P.z - g.T[2];
    const double F2 = fabs(g.rz[0] * dx + g.rz[1] * dy + g.rz[2] * dz), nP = sqrt(dx * dx + dy * dy + dz * dz);
    const double iw = F2 / fabs(g.qz);                                   // |iW|: the lateral spacing of adjacent pixels at this depth
    const double LB = F2 * (WR + 1) / (nP / fmax(iw, 1e-300) + (WR + 1));
    const double coord = fmax(fmax(fabs((double)P.x), fabs((double)P.y)), fabs((double)P.z)) + nP;
    const double lim = LB * (1.0 - 1e-5) - 4e-7 * coord; // float32 coordinates and float32 distance arithmetic: relative 1e-7 each
    *lim_out = lim;
    const float range = (lim > 0.0) ? (float)(lim * lim * (1.0 - 1e-6)) : 0.0f; // tau must stay below this; the bins cover [0, range)
    const int want = mean_k + 1;
    const float inv_w = (range > 0.0f) ? (float)WIN_NB / range : 0.0f;
    if (!(range > 0.0f && inv_w > 0.0f && isfinite(inv_w))) return false;
    // bin of a squared distance: 0 .. WIN_NB - 1 below the bound, WIN_NB (the sink row) at or beyond it and for NaN (a pixel without a
    // point: fminf returns the other operand) -- monotone in d2; one multiply, one minimum, one conversion
    auto bin_of = [&](float d2) { return (int)fminf(d2 * inv_w, (float)WIN_NB); };
    // A chunk of a window row at a time: all its reads first (16-byte reads, in flight together), then the arithmetic -- a read,
    // its wait and a data-dependent branch per candidate would expose the read latency (2 x 1089 times per query at WR = 16).
    // the squared distance of fdist2 -- (dx dx + dy dy) + dz dz, the same operations on the same operands -- with x and y as one
    // packed pair (v_pk_add_f32 / v_pk_mul_f32: the lattice entry's x and y arrive in adjacent registers)
    typedef float v2f __attribute__((ext_vector_type(2)));
    const v2f Pxy = {P.x, P.y};
    auto d2_of = [&](const float4 &o) {
        const v2f oxy = {o.x, o.y};
        const v2f dxy = Pxy - oxy;
        const v2f sq = dxy * dxy;
        const float dz = P.z - o.z;
        return (sq.x + sq.y) + dz * dz;
    };
    // The disc, a row class at a time.  A row is cut into an even number of chunks of at most CH candidates and the chunks stream
    // through two register buffers: the 16-byte reads of the NEXT chunk (of this row, or the first of the next row) are issued
    // before the arithmetic of the current one, so a wave computes while its own reads are under way (round 5 read a whole row,
    // waited, computed: with the two waves per SIMD that 75 KB of LDS leave, the vector unit idled a third of the time).
    // pre(n) before a chunk of n candidates, body(d2) per candidate; NaN for a pixel without a point: every test fails.
    auto for_disc = [&](auto &&pre, auto &&body) {
        auto rows = [&](int lo, int hi, auto hw_) { // the rows with |b| in [lo, hi] (lo = 0: one run of rows, else one above and one below), half width hw
            constexpr int hw = decltype(hw_)::value, wd = 2 * hw + 1, nch0 = (wd + CH - 1) / CH, nch = nch0 + (nch0 & 1);
            typedef WinSplit<wd, nch> S;
            float4 buf[2][S::nmax];
            auto issue = [&](auto j_, auto b_, int r) { // the reads of chunk j of window row r, into buffer b
                constexpr int j = decltype(j_)::value, bb = decltype(b_)::value;
#pragma unroll
                for (int i = 0; i < S::size(j); i++) buf[bb][i] = ld(r, WR - hw + S::off(j) + i);
            };
            auto use = [&](auto j_, auto b_) {
                constexpr int j = decltype(j_)::value, bb = decltype(b_)::value;
#pragma unroll
                for (int i = 0; i < S::size(j); i++) asm volatile("" : "+v"(buf[bb][i].w)); // (keeps the reads 16-byte ones: twice the LDS rate of the 12-byte form)
                pre(S::size(j));
#pragma unroll
                for (int i = 0; i < S::size(j); i++) body(d2_of(buf[bb][i]));
            };
            auto steps = [&](auto &&self, auto j_, int r, int rn) -> void {
                constexpr int j = decltype(j_)::value;
                if constexpr (j + 1 < nch) issue(std::integral_constant<int, j + 1>(), std::integral_constant<int, (j + 1) & 1>(), r);
                else issue(std::integral_constant<int, 0>(), std::integral_constant<int, 0>(), rn);
                __builtin_amdgcn_sched_barrier(0); // (the reads stay ahead of the arithmetic)
                use(j_, std::integral_constant<int, j & 1>());
                if constexpr (j + 1 < nch) self(self, std::integral_constant<int, j + 1>(), r, rn);
            };
#pragma unroll 1
            for (int run = 0; run < (lo == 0 ? 1 : 2); run++) {
                const int r0 = lo == 0 ? WR - hi : (run ? WR + lo : WR - hi), r1 = lo == 0 ? WR + hi : (run ? WR + hi : WR - lo);
                issue(std::integral_constant<int, 0>(), std::integral_constant<int, 0>(), r0);
#pragma unroll 1
                for (int r = r0; r <= r1; r++) steps(steps, std::integral_constant<int, 0>(), r, r + (r < r1 ? 1 : 0)); // (after the last row: its first chunk once more, unused)
            }
        };
        constexpr int g1 = WR / 2, g2 = 3 * WR / 4, g3 = 7 * WR / 8;
        rows(0, g1, std::integral_constant<int, win_halfwidth(WR, 0)>());
        if (g2 > g1) rows(g1 + 1, g2, std::integral_constant<int, win_halfwidth(WR, g1 + 1)>());
        if (g3 > g2) rows(g2 + 1, g3, std::integral_constant<int, win_halfwidth(WR, g2 + 1)>());
        if (WR > g3) rows(g3 + 1, WR, std::integral_constant<int, win_halfwidth(WR, g3 + 1)>());
    };
    // pass 1: histogram of the window's squared distances (the point itself included: d2 = 0, as nearestKSearch(k + 1)); branch-free:
    // what lies beyond the bound goes to the sink row
    for_disc([](int) {}, [&](float d2) {
        const int b = bin_of(d2);
#if defined(WIN_EXP) && (WIN_EXP & 1) // timing experiment (results invalid): no histogram updates
        asm volatile("" ::"v"(b));
#else
        atomicAdd(col + 256 * b, 1u); // (no return value: a fire-and-forget ds_add)
#endif
    });
    // the bin holding rank `want` (all counters read first: one LDS round trip, not one per bin)
    int below = 0, bstar = -1, in_bin = 0;
    {
        int h[WIN_NB];
#pragma unroll
        for (int b = 0; b < WIN_NB; b++) h[b] = (int)col[256 * b];
#pragma unroll
        for (int b = 0; b < WIN_NB; b++) {
            const bool hit = bstar < 0 && below + h[b] >= want;
            bstar = hit ? b : bstar;
            in_bin = hit ? h[b] : in_bin;
            below = bstar < 0 ? below + h[b] : below;
        }
    }
    if (!(bstar >= 0 && in_bin <= WIN_LCAP)) return false; // fewer than k + 1 points within the bound in the window, or too many (nearly) equal distances
    // pass 2: the exact sum below the bin, the bin's values listed (over the counters: they are dead now).  The bin tests become
    // two float compares: t_lo / t_hi = the smallest floats whose bin is >= bstar / > bstar (bin_of is monotone).
    // A tenth of a wave's lanes has a value below t_lo per candidate, so a square root taken where the value is met runs its whole
    // sequence for a few lanes nearly every time (36 instructions per candidate, round 5).  Instead every value below t_hi WAITS in
    // the lane's own column -- rows in_bin .. WIN_NB, the rows the bin's list leaves free -- and before a chunk that some lane has
    // no room for, the wave takes the square roots of everything waiting, eight rows at a time (the reads in flight together, the
    // sequences interleaved): most lanes busy, a tenth as often.  The sum is exact in double (order-free), so its bits are those
    // of the sum in window order.
    float *lst = (float *)col;
    double sum = 0.0;
    int nl = 0;
    unsigned int odd_min = 0xffffffffu; // tracks whether an input outside the trimmed square root's range (0 < d2 < 2^-96) was met
    auto first_in = [&](int b) { // smallest float t >= 0 with bin_of(t) >= b, b in 1 .. WIN_NB
        float t = (float)b / inv_w;
        while (t > 0.0f && bin_of(__uint_as_float(__float_as_uint(t) - 1u)) >= b) t = __uint_as_float(__float_as_uint(t) - 1u);
        while (bin_of(t) < b) t = __uint_as_float(__float_as_uint(t) + 1u);
        return t;
    };
    const float t_lo = bstar > 0 ? first_in(bstar) : 0.0f;
    const float t_hi = fminf(first_in(bstar + 1), range); // (a value at the bound's edge can round into the last bin: it is not listed, the count then disagrees and the query is left undecided)
    static_assert(WIN_NB + 1 - WIN_LCAP >= CH, "a lane's waiting rows hold at least one chunk");
    typedef __attribute__((address_space(3))) float lds_float; // (32-bit LDS addresses: a generic pointer costs 64-bit arithmetic per candidate)
    lds_float *const wait0 = (lds_float *)lst + 256 * in_bin, *const wait_end = (lds_float *)lst + 256 * (WIN_NB + 1), *const wait_last = wait_end - 256;
    lds_float *wait = wait0;
    asm volatile("" : "+v"(wait)); // (one register, not base + offset: an addition less per waiting value)
    auto drain = [&]() {
        // (the trimmed square root itself; an input it is not made for -- positive and below 2^-96: float32 coordinates do not
        // produce such differences -- marks the query `odd`, and it is left to the later passes, which take sqrtf)
#pragma unroll 1
        for (const lds_float *p = wait0; p < wait; p += WIN_DRAIN_W * 256) {
            float v[WIN_DRAIN_W];
#pragma unroll
            for (int t = 0; t < WIN_DRAIN_W; t++) v[t] = *(p + 256 * t < wait_last ? p + 256 * t : wait_last); // (rows at or beyond `wait`: read, not used)
#pragma unroll
            for (int t = 0; t < WIN_DRAIN_W; t++) {
                const bool live = p + 256 * t < wait, low = v[t] < t_lo;
                if (live && !low) { // one of the bin's own values: to the list
                    lst[min(nl, in_bin) * 256] = v[t]; // (nl <= in_bin by the histogram; the clamp keeps a disagreement inside the column -- it then
                    nl++;                              // spoils a waiting value of a query that is left undecided anyway)
                }
                const float u = live && low ? v[t] : 0.0f;       // (the square root of 0 adds nothing)
                odd_min = min(odd_min, __float_as_uint(u) - 1u); // (the smallest positive pattern met, minus one; 0 wraps to the top)
                sum += (double)sqrtf_rn_core(u);
            }
        }
        wait = wait0;
        asm volatile("" : "+v"(wait));
    };
#if !(defined(WIN_EXP) && (WIN_EXP & 2)) // (timing experiment, results invalid: no second pass)
    for_disc(
        [&](int n) {
            if (__builtin_expect(__any(wait + 256 * n > wait_end), 0)) drain();
        },
        [&](float d2) {
            if (d2 < t_hi) // below the bin or in it: waits (NaN -- a pixel without a point -- fails the test).  *wait++ = d2 with a row's stride, as
                           // the two instructions it is (the compiler adds into a temporary and copies it back: a third of this region)
            {
                *wait = d2;
                wait += 256;
            }
        });
#endif
    drain();
    // tau = the (want - below)-th smallest of the listed values (by value: ties are equal distances): the list into registers
    // (unused slots +inf), sorted by Batcher's merge exchange on the bit patterns (non-negative floats order as unsigned integers)
    const int rank = want - below; // 1 .. in_bin
    unsigned int u[WIN_LCAP];
#pragma unroll
    for (int j = 0; j < WIN_LCAP; j++) u[j] = __float_as_uint(lst[j * 256]);
#pragma unroll
    for (int j = 0; j < WIN_LCAP; j++) u[j] = j < nl ? u[j] : 0x7f800000u;
#pragma unroll
    for (int pp = 1; pp < WIN_LCAP; pp *= 2) {
#pragma unroll
        for (int k = pp; k >= 1; k /= 2) {
#pragma unroll
            for (int j = k % pp; j + k < WIN_LCAP; j += 2 * k) {
#pragma unroll
                for (int i = 0; i < (k < WIN_LCAP - j - k ? k : WIN_LCAP - j - k); i++) {
                    if ((i + j) / (2 * pp) == (i + j + k) / (2 * pp)) {
                        const unsigned int lo = min(u[i + j], u[i + j + k]), hi = max(u[i + j], u[i + j + k]);
                        u[i + j] = lo;
                        u[i + j + k] = hi;
                    }
                }
            }
        }
    }
    unsigned int tau_b = 0u;
#pragma unroll
    for (int j = 0; j < WIN_LCAP; j++) tau_b = (j == rank - 1) ? u[j] : tau_b;
    const float tau = __uint_as_float(tau_b);
    int less = 0;
#pragma unroll
    for (int j = 0; j < WIN_LCAP; j++) less += u[j] < tau_b;
#pragma unroll
    for (int j = 0; j < WIN_LCAP; j++) { // the values below tau are the first `less` of the sorted list
        if (!__any(j < less)) break;
        const float x = j < less ? __uint_as_float(u[j]) : 0.0f;
        odd_min = min(odd_min, __float_as_uint(x) - 1u);
        sum += (double)sqrtf_rn_core(x);
    }
    if (!(nl == in_bin && tau < range) || odd_min < 0x0f7fffffu) return false;
    // every point outside the window is farther than sqrt(tau): the k + 1 smallest are all here
    *out = (float)((sum + (double)(want - (below + less)) * (double)sqrtf(tau)) / mean_k);
    return true;
}

// Tile form: a workgroup = a 32 x 8 tile of pixels, its window halo staged in LDS, a thread = the query of its pixel.
// PROBE: every `tile_step`-th tile in x and y only, nothing written but three counters (queries seen, queries decided, the sum of
// their bounds): the host picks the smallest window radius that decides most of a sparse sample before it pays for the full pass.
template <int WR, bool PROBE>
__global__ __launch_bounds__(256) void k_sor_window(const float4 *__restrict__ lat, WinGeom g, int mean_k, float *__restrict__ dist,
                                                     unsigned int *__restrict__ undecided, int tile_step, unsigned int *__restrict__ probe_cnt) {
    constexpr int SW = WIN_TX + 2 * WR, SH = WIN_TY + 2 * WR;
    __shared__ float4 s_t[SH][SW];
    __shared__ unsigned int s_u[WIN_NB + 1][256]; // column tid is private to thread tid: first the bin counters (row WIN_NB: the sink of
                                                  // candidates beyond the bound / pixels without a point), then the list of the rank's bin
    static_assert(WIN_LCAP <= WIN_NB, "the list reuses the counters' storage");
    const int tid = threadIdx.x, tx = tid & (WIN_TX - 1), ty = tid / WIN_TX;
    const int gx0 = WIN_PAD + (int)blockIdx.x * tile_step * WIN_TX - WR, gy0 = WIN_PAD + (int)blockIdx.y * tile_step * WIN_TY - WR; // lattice position of s_t[0][0]
    for (int e = tid; e < SW * SH; e += 256) {
        const int r = e / SW, c = e - r * SW;
        const int gy = min(gy0 + r, g.gh - 1), gx = min(gx0 + c, g.gw - 1); // (the last tiles' overhang re-reads the border: NaN)
        s_t[r][c] = lat[(size_t)gy * g.gw + gx];
    }
#pragma unroll
    for (int b = 0; b <= WIN_NB; b++) s_u[b][tid] = 0u;
    __syncthreads();
    const float4 P = s_t[ty + WR][tx + WR];
    const bool valid = P.x == P.x; // NaN: no point at this pixel
    if (!__syncthreads_or(valid)) return;
    auto ld = [&](int r, int c) { return s_t[ty + r][tx + c]; };
    float res = 0.0f;
    double lim = 0.0;
    const bool ok = valid && win_query<WR, WIN_TILE_CH>(ld, P, g, mean_k, &s_u[0][tid], &res, &lim);
    if (PROBE) {
        const unsigned long long mv = __ballot(valid), mo = __ballot(ok);
        float sl = valid ? (float)fmax(lim, 0.0) : 0.0f; // the mean bound: what the queries this radius leaves over have in common (tau >= lim^2)
        for (int o = 32; o > 0; o >>= 1) sl += __shfl_xor(sl, o);
        if ((tid & 63) == 0 && mv) {
            atomicAdd(&probe_cnt[0], (unsigned int)__popcll(mv));
            atomicAdd(&probe_cnt[1], (unsigned int)__popcll(mo));
            atomicAdd((float *)&probe_cnt[2], sl);
        }
    } else if (valid) {
        if (ok) dist[__float_as_uint(P.w)] = res;
        undecided[__float_as_uint(P.w)] = ok ? 0u : 1u;
    }
}

// List form: a thread = one listed query (what the tile pass left over -- a seventh of a thick sheet's points, scattered), its
// window read straight from the lattice copy in global memory (neighbouring queries share its lines in L1 / L2): the wide window
// the scattered rest needs, at full lane occupancy, where the tile form would run a 49 x 49 window for one lane in seven.
template <int WR, int CH>
__global__ __launch_bounds__(256) void k_sor_window_list(const float4 *__restrict__ lat, const unsigned int *__restrict__ cell_of, WinGeom g, int mean_k,
                                                          const unsigned int *__restrict__ list, int nq, float *__restrict__ dist, unsigned int *__restrict__ undecided) {
    __shared__ unsigned int s_u[WIN_NB + 1][256];
    const int tid = threadIdx.x, q = blockIdx.x * 256 + tid;
#pragma unroll
    for (int b = 0; b <= WIN_NB; b++) s_u[b][tid] = 0u;
    if (q >= nq) return; // (no workgroup barrier below: the columns are private)
    const unsigned int pt = list[q];
    const unsigned int cell = cell_of[pt]; // the point's lattice position gy * gw + gx
    const float4 *base = lat + (size_t)cell - (size_t)WR * g.gw - WR;
    const float4 P = lat[cell];
    auto ld = [&](int r, int c) { return base[(size_t)r * g.gw + c]; };
    float res = 0.0f;
    double lim = 0.0;
    const bool ok = win_query<WR, CH>(ld, P, g, mean_k, &s_u[0][tid], &res, &lim);
    if (ok) dist[pt] = res;
    undecided[q] = ok ? 0u : 1u; // (indexed like the list: the caller compacts it into the ladder's first list)
}

// Wave form: a wave = one listed query -- what the list passes leave: the points within a few dozen pixels of the mask's corners,
// around holes, the outliers this filter exists to remove; hundreds, not millions -- its window of ANY radius read from the lattice
// copy, candidates lane-strided (wave_knn_core: the grid search's selection).  The same bound as the other forms decides whether
// the window holds the k + 1 nearest; the window is clipped to the lattice copy (no point lies outside the margin's box).
__global__ __launch_bounds__(256) void k_sor_window_wave(const float4 *__restrict__ lat, const unsigned int *__restrict__ cell_of, WinGeom g, int mean_k, int WR,
                                                          const unsigned int *__restrict__ list, int nq, float *__restrict__ dist, unsigned int *__restrict__ undecided) {
    __shared__ float s_d2[4][KNN_CAP];
    const int lane = threadIdx.x & 63;
    const int qi = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (qi >= nq) return; // wave-uniform; no workgroup barrier below
    const unsigned int pt = list[qi];
    const unsigned int cell = cell_of[pt];
    const int cy = (int)(cell / (unsigned int)g.gw), cx = (int)(cell - (unsigned int)cy * (unsigned int)g.gw);
    const float4 P = lat[cell];
    // the bound and its rounding margin, as win_query forms them
    const double dx = (double)P.x - g.T[0], dy = (double)P.y - g.T[1], dz = (double)P.z - g.T[2];
    const double F2 = fabs(g.rz[0] * dx + g.rz[1] * dy + g.rz[2] * dz), nP = sqrt(dx * dx + dy * dy + dz * dz);
    const double iw = F2 / fabs(g.qz);
    const double LB = F2 * (WR + 1) / (nP / fmax(iw, 1e-300) + (WR + 1));
    const double coord = fmax(fmax(fabs((double)P.x), fabs((double)P.y)), fabs((double)P.z)) + nP;
    const double lim = LB * (1.0 - 1e-5) - 4e-7 * coord;
    const float range = (lim > 0.0) ? (float)(lim * lim * (1.0 - 1e-6)) : 0.0f; // tau must stay below this
    if (!(range > 0.0f && isfinite(range))) {
        if (lane == 0) undecided[qi] = 1u;
        return;
    }
    const float h2 = __uint_as_float(__float_as_uint(range) - 1u); // d2 <= h2 is d2 < range
    const int x0 = max(cx - WR, 0), x1 = min(cx + WR, g.gw - 1), y0 = max(cy - WR, 0), y1 = min(cy + WR, g.gh - 1);
    const int ncx = x1 - x0 + 1, M = ncx * (y1 - y0 + 1);
    const float4 *base = lat + (size_t)y0 * g.gw + x0;
    const int gw = g.gw;
    // (row of candidate c by a multiplication: c / ncx exactly for c ncx < 2^32 -- the integer division was half of this kernel's instructions)
    const unsigned int ncx_inv = 0xffffffffu / (unsigned int)ncx + 1u;
    auto cand = [&](int c) -> float { // NaN for a pixel without a point: every comparison fails
        const int r = ncx > 1 ? (int)__umulhi((unsigned int)c, ncx_inv) : c, col = c - r * ncx; // (ncx = 1: the reciprocal does not fit)
        const float4 o = base[(size_t)r * gw + col];
        return fdist2(P.x, P.y, P.z, o.x, o.y, o.z);
    };
    float tau;
    double sum;
    int less;
    const int want = mean_k + 1;
    if (!wave_knn_core(cand, M, h2, want, s_d2[threadIdx.x >> 6], lane, &undecided[qi], tau, sum, less)) return;
    if (lane == 0) dist[pt] = (float)((sum + (double)(want - less) * (double)sqrtf(tau)) / mean_k);
}

// Workgroup form: the wave form's query with its first pass -- every candidate of the window, those within the bound kept -- made by
// four waves together, for the passes that have only hundreds of queries left (C2: 564 at 80 pixels, 15 at 160): a wave per query
// leaves the chip idle and takes as long as its 400 ... 1 600 chunks of candidates do.  Wave w takes the chunks w, w + 4, ... into
// its own quarter of the list (a fixed order: the sum's bits do not depend on timing), the quarters are put together and wave 0
// selects as the wave form does; a quarter that overflows leaves the whole query to wave 0.
__global__ __launch_bounds__(256) void k_sor_window_wg(const float4 *__restrict__ lat, const unsigned int *__restrict__ cell_of, WinGeom g, int mean_k, int WR,
                                                        const unsigned int *__restrict__ list, int nq, float *__restrict__ dist, unsigned int *__restrict__ undecided) {
    constexpr int WG_C = 64, WG_CAP = 64 * WG_C; // (twice the wave form's list: these few queries' windows hold thousands of candidates within the
                                                 // bound, and one pass over them by four waves beats three by one)
    __shared__ float s_seg[4][WG_CAP / 4];
    __shared__ float s_d2[WG_CAP];
    __shared__ int s_cnt[4];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int qi = blockIdx.x;
    const unsigned int pt = list[qi];
    const unsigned int cell = cell_of[pt];
    const int cy = (int)(cell / (unsigned int)g.gw), cx = (int)(cell - (unsigned int)cy * (unsigned int)g.gw);
    const float4 P = lat[cell];
    const double dx = (double)P.x - g.T[0], dy = (double)P.y - g.T[1], dz = (double)P.z - g.T[2];
    const double F2 = fabs(g.rz[0] * dx + g.rz[1] * dy + g.rz[2] * dz), nP = sqrt(dx * dx + dy * dy + dz * dz);
    const double iw = F2 / fabs(g.qz);
    const double LB = F2 * (WR + 1) / (nP / fmax(iw, 1e-300) + (WR + 1));
    const double coord = fmax(fmax(fabs((double)P.x), fabs((double)P.y)), fabs((double)P.z)) + nP;
    const double lim = LB * (1.0 - 1e-5) - 4e-7 * coord;
    const float range = (lim > 0.0) ? (float)(lim * lim * (1.0 - 1e-6)) : 0.0f; // (k_sor_window_wave's bound, the same expressions)
    if (!(range > 0.0f && isfinite(range))) { // uniform
        if (threadIdx.x == 0) undecided[qi] = 1u;
        return;
    }
    const float h2 = __uint_as_float(__float_as_uint(range) - 1u);
    const int x0 = max(cx - WR, 0), x1 = min(cx + WR, g.gw - 1), y0 = max(cy - WR, 0), y1 = min(cy + WR, g.gh - 1);
    const int ncx = x1 - x0 + 1, M = ncx * (y1 - y0 + 1);
    const float4 *base = lat + (size_t)y0 * g.gw + x0;
    const int gw = g.gw;
    const unsigned int ncx_inv = 0xffffffffu / (unsigned int)ncx + 1u;
    auto cand = [&](int c) -> float {
        const int r = ncx > 1 ? (int)__umulhi((unsigned int)c, ncx_inv) : c, col = c - r * ncx;
        const float4 o = base[(size_t)r * gw + col];
        return fdist2(P.x, P.y, P.z, o.x, o.y, o.z);
    };
    const float inf = __uint_as_float(0x7f800000u);
    const unsigned long long lt = (1ull << lane) - 1ull;
    int Kw = 0;
    for (int c0 = w * 64 * KNN_B; c0 < M; c0 += 4 * 64 * KNN_B) { // wave-uniform
        float v[KNN_B];
#pragma unroll
        for (int i = 0; i < KNN_B; i++) {
            const int c = c0 + lane + 64 * i;
            v[i] = (c < M) ? cand(c) : inf;
        }
#pragma unroll
        for (int i = 0; i < KNN_B; i++) {
            const bool keep = v[i] <= h2;
            const unsigned long long mm = __ballot(keep);
            const int pos = Kw + __popcll(mm & lt);
            if (keep && pos < WG_CAP / 4) s_seg[w][pos] = v[i];
            Kw += __popcll(mm);
        }
    }
    if (lane == 0) s_cnt[w] = Kw;
    __syncthreads();
    const int k0 = s_cnt[0], k1 = s_cnt[1], k2 = s_cnt[2], k3 = s_cnt[3];
    const bool fits = k0 <= WG_CAP / 4 && k1 <= WG_CAP / 4 && k2 <= WG_CAP / 4 && k3 <= WG_CAP / 4;
    if (fits) {
        const int off = w == 0 ? 0 : (w == 1 ? k0 : (w == 2 ? k0 + k1 : k0 + k1 + k2));
        for (int j = lane; j < Kw; j += 64) s_d2[off + j] = s_seg[w][j];
    }
    __syncthreads();
    if (w != 0) return;
    float tau;
    double sum;
    int less;
    const int want = mean_k + 1;
    if (!wave_knn_core<WG_C>(cand, M, h2, want, s_d2, lane, &undecided[qi], tau, sum, less, fits ? k0 + k1 + k2 + k3 : -1)) return;
    if (lane == 0) dist[pt] = (float)((sum + (double)(want - less) * (double)sqrtf(tau)) / mean_k);
}

void launch_cloud_lattice(const uint8_t *flags, const int64_t *row_offset, int W, int XL, int XR, int YL, int YR, const double *xyz, int64_t n, float4 *lat,
                          unsigned int *cell_of, hipStream_t st) {
    const int gw = XR - XL + 1 + 2 * WIN_PAD, gh = YR - YL + 1 + 2 * WIN_PAD;
    (void)hipMemsetD32Async((hipDeviceptr_t)lat, 0x7fc00000, (size_t)4 * gw * gh, st);
    hipLaunchKernelGGL(k_cloud_lattice, dim3((unsigned)(YR - YL + 1)), dim3(256), 0, st, flags, row_offset, W, XL, XR, YL, xyz, n, gw, lat, cell_of);
}
static WinGeom win_geom(int XL, int XR, int YL, int YR, double qz, const double *R, const double *T) {
    WinGeom g;
    g.gw = XR - XL + 1 + 2 * WIN_PAD;
    g.gh = YR - YL + 1 + 2 * WIN_PAD;
    g.qz = qz;
    for (int i = 0; i < 3; i++) {
        g.rz[i] = R[3 * i + 2];
        g.T[i] = T[i];
    }
    return g;
}
// the list form over nq listed queries at radius 24 or 40: dist (per point) / undecided (per list entry)
void launch_sor_window_list(const float4 *lat, const unsigned int *cell_of, int XL, int XR, int YL, int YR, double qz, const double *R, const double *T, int mean_k,
                            int radius, const unsigned int *list, int nq, float *dist, unsigned int *undecided, hipStream_t st) {
    if (nq <= 0) return;
    const WinGeom g = win_geom(XL, XR, YL, YR, qz, R, T);
    const dim3 grid((unsigned)((nq + 255) / 256));
    if (radius <= 24) hipLaunchKernelGGL((k_sor_window_list<24, 7>), grid, dim3(256), 0, st, lat, cell_of, g, mean_k, list, nq, dist, undecided);
    else hipLaunchKernelGGL((k_sor_window_list<40, 9>), grid, dim3(256), 0, st, lat, cell_of, g, mean_k, list, nq, dist, undecided);
}
// the wave form over nq listed queries at any radius: dist (per point) / undecided (per list entry)
void launch_sor_window_wave(const float4 *lat, const unsigned int *cell_of, int XL, int XR, int YL, int YR, double qz, const double *R, const double *T, int mean_k,
                            int radius, const unsigned int *list, int nq, float *dist, unsigned int *undecided, hipStream_t st, int wg_max = 2048) {
    if (nq <= 0) return;
    const WinGeom g = win_geom(XL, XR, YL, YR, qz, R, T);
    if (nq <= wg_max) // (few enough to leave most of the chip idle at a wave each: four waves per query)
        hipLaunchKernelGGL(k_sor_window_wg, dim3((unsigned)nq), dim3(256), 0, st, lat, cell_of, g, mean_k, radius, list, nq, dist, undecided);
    else
        hipLaunchKernelGGL(k_sor_window_wave, dim3((unsigned)((nq + 3) / 4)), dim3(256), 0, st, lat, cell_of, g, mean_k, radius, list, nq, dist, undecided);
}
size_t cloud_lattice_bytes(int XL, int XR, int YL, int YR) {
    return sizeof(float4) * (size_t)(XR - XL + 1 + 2 * WIN_PAD) * (size_t)(YR - YL + 1 + 2 * WIN_PAD);
}
// the window pass over the whole lattice: dist / undecided (one entry per cloud point; undecided pre-set to 1 by the caller);
// tile_step > 1 + probe_cnt: the sparse probe (see k_sor_window)
void launch_sor_window(const float4 *lat, int XL, int XR, int YL, int YR, double qz, const double *R, const double *T, int mean_k, int radius, float *dist,
                       unsigned int *undecided, hipStream_t st, int tile_step = 1, unsigned int *probe_cnt = nullptr) {
    const WinGeom g = win_geom(XL, XR, YL, YR, qz, R, T);
    const int tx = (XR - XL + 1 + WIN_TX - 1) / WIN_TX, ty = (YR - YL + 1 + WIN_TY - 1) / WIN_TY;
    const dim3 grid((unsigned)((tx + tile_step - 1) / tile_step), (unsigned)((ty + tile_step - 1) / tile_step));
#define WIN_LAUNCH(R_)                                                                                                                     \
    do {                                                                                                                                   \
        if (probe_cnt) hipLaunchKernelGGL((k_sor_window<R_, true>), grid, dim3(256), 0, st, lat, g, mean_k, dist, undecided, tile_step, probe_cnt); \
        else hipLaunchKernelGGL((k_sor_window<R_, false>), grid, dim3(256), 0, st, lat, g, mean_k, dist, undecided, tile_step, probe_cnt);          \
    } while (0)
    switch (radius) {
    case 7: WIN_LAUNCH(7); break;
    case 12: WIN_LAUNCH(12); break;
    case 16: WIN_LAUNCH(16); break;
    case 20: WIN_LAUNCH(20); break;
    default: WIN_LAUNCH(24); break;
    }
#undef WIN_LAUNCH
}

// ---- the few queries no grid level could decide (isolated points -- what this filter removes -- and clusters smaller
// than k): against ALL points, the whole chip on every query.  Three-pass radix select of the (k+1)-th smallest squared
// distance (bits 30..20, 19..9, 8..0 of its pattern): per pass every workgroup histograms its slice of the points for
// one query and adds the non-empty bins to the query's global histogram, a one-workgroup kernel picks the bin; then the
// partial sums of sqrt(d2) over d2 < tau (exact in double for <= 100 float32 values, hence order-free) are added up.
#define XQ_SLICES 128
struct XqState { // per listed query
    unsigned int prefix, mask;
    int rank;
    int less;
    double sum;
};
__global__ void k_xq_init(XqState *__restrict__ st, int count, int mean_k, int n) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= count) return;
    st[q].prefix = 0u;
    st[q].mask = 0u;
    st[q].rank = min(mean_k + 1, n);
    st[q].less = 0;
    st[q].sum = 0.0;
}
__global__ __launch_bounds__(256) void k_xq_hist(const float *__restrict__ xyz, const float4 *__restrict__ sxyz, int n,
                                                  const unsigned int *__restrict__ list, int count,
                                                  const XqState *__restrict__ st, int shift, int width, int *__restrict__ ghist) {
    __shared__ int hist[2048];
    const int per = (n + XQ_SLICES - 1) / XQ_SLICES, q0 = blockIdx.x * per, q1 = min(n, q0 + per);
    for (int item = blockIdx.y; item < count; item += gridDim.y) { // uniform
        const unsigned int orig = list[item];
        const float px = xyz[3 * (size_t)orig], py = xyz[3 * (size_t)orig + 1], pz = xyz[3 * (size_t)orig + 2];
        const unsigned int prefix = st[item].prefix, mask = st[item].mask;
        for (int b = threadIdx.x; b < 2048; b += 256) hist[b] = 0;
        __syncthreads();
        for (int q = q0 + threadIdx.x; q < q1; q += 256) {
            const float4 o = sxyz[q];
            const unsigned int u = __float_as_uint(fdist2(px, py, pz, o.x, o.y, o.z));
            // the distances to one far query share a handful of bins: one LDS atomic per distinct bin of the wave
            int bin = ((u & mask) == prefix) ? (int)((u >> shift) & ((1u << width) - 1u)) : -1;
            while (__ballot(bin >= 0)) { // uniform
                const unsigned long long act = __ballot(bin >= 0);
                const int b0 = __shfl(bin, __ffsll((long long)act) - 1);
                const unsigned long long same = __ballot(bin == b0);
                if (((int)threadIdx.x & 63) == __ffsll((long long)act) - 1) atomicAdd(&hist[b0], __popcll(same));
                if (bin == b0) bin = -1;
            }
        }
        __syncthreads();
        for (int b = threadIdx.x; b < (1 << width); b += 256)
            if (hist[b]) atomicAdd(&ghist[(size_t)item * 2048 + b], hist[b]);
        __syncthreads();
    }
}
__global__ __launch_bounds__(64) void k_xq_pick(XqState *__restrict__ st, int count, int shift, int width,
                                                 int *__restrict__ ghist) {
    const int item = blockIdx.x;
    if (item >= count) return;
    int *h = ghist + (size_t)item * 2048;
    if (threadIdx.x == 0) {
        int rank = st[item].rank, b = 0;
        for (; b < (1 << width) - 1; b++) {
            if (h[b] >= rank) break;
            rank -= h[b];
        }
        st[item].rank = rank;
        st[item].prefix |= (unsigned int)b << shift;
        st[item].mask |= ((1u << width) - 1u) << shift;
    }
    __syncthreads();
    for (int b = threadIdx.x; b < 2048; b += 64) h[b] = 0; // ready for the next pass
}
__global__ __launch_bounds__(256) void k_xq_sum(const float *__restrict__ xyz, const float4 *__restrict__ sxyz, int n,
                                                 const unsigned int *__restrict__ list, int count,
                                                 XqState *__restrict__ st) {
    __shared__ double s_sum[4];
    __shared__ int s_less[4];
    const int per = (n + XQ_SLICES - 1) / XQ_SLICES, q0 = blockIdx.x * per, q1 = min(n, q0 + per);
    for (int item = blockIdx.y; item < count; item += gridDim.y) {
        const unsigned int orig = list[item];
        const float px = xyz[3 * (size_t)orig], py = xyz[3 * (size_t)orig + 1], pz = xyz[3 * (size_t)orig + 2];
        const float tau = __uint_as_float(st[item].prefix);
        double sum = 0.0;
        int less = 0;
        for (int q = q0 + threadIdx.x; q < q1; q += 256) {
            const float4 o = sxyz[q];
            const float d2 = fdist2(px, py, pz, o.x, o.y, o.z);
            if (d2 < tau) {
                sum += (double)sqrtf(d2);
                less++;
            }
        }
        for (int o = 32; o > 0; o >>= 1) {
            sum += __shfl_xor(sum, o);
            less += __shfl_xor(less, o);
        }
        if ((threadIdx.x & 63) == 0) {
            s_sum[threadIdx.x >> 6] = sum;
            s_less[threadIdx.x >> 6] = less;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            const int l = s_less[0] + s_less[1] + s_less[2] + s_less[3];
            if (l) { // at most k values are below tau over ALL slices: the double sum is exact, any order
                atomicAdd(&st[item].sum, (s_sum[0] + s_sum[1]) + (s_sum[2] + s_sum[3]));
                atomicAdd(&st[item].less, l);
            }
        }
        __syncthreads();
    }
}
__global__ void k_xq_finish(const XqState *__restrict__ st, const unsigned int *__restrict__ list, int count, int mean_k,
                            int n, float *__restrict__ dist_orig) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= count) return;
    const int want = min(mean_k + 1, n);
    const float tau = __uint_as_float(st[q].prefix);
    dist_orig[list[q]] = (float)((st[q].sum + (double)(want - st[q].less) * (double)sqrtf(tau)) / mean_k);
}

// ---- sum and sum of squares of the per-point distances as PCL forms them (a sequential loop adding floats to doubles,
// `sq_sum += d * d` with the product in float).  A parallel reduction gives the same bits whenever NO addition rounds:
// every value is a multiple of 2^q (q = the lowest set bit over all values) and the total is below 2^(q + 53), so every
// partial sum, in any order, is representable.  The kernel returns the sums and the two q's; the host checks the
// condition and falls back to the sequential loop otherwise (never on the test or bench clouds).
__device__ __forceinline__ int low_bit_exp(float v) { // exponent of the lowest set bit of a finite float, INT_MAX for 0
    const unsigned int u = __float_as_uint(v) & 0x7fffffffu;
    if (u == 0u) return 0x7fffffff;
    const int e = (int)(u >> 23);
    const unsigned int m = u & 0x7fffffu;
    if (e == 0) return -149 + (__ffs((int)m) - 1);           // subnormal
    return (e - 150) + (m ? __ffs((int)m) - 1 : 23);
}
struct DistStats {
    double sum, sq_sum;
    int q_sum, q_sq;
    int bad; // a negative or non-finite distance (cannot happen; makes the host take the sequential path)
};
__global__ __launch_bounds__(256) void k_dist_stats(const float *__restrict__ dist, int64_t n, DistStats *__restrict__ out) {
    double s = 0.0, s2 = 0.0;
    int q1 = 0x7fffffff, q2 = 0x7fffffff, bad = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float d = dist[i];
        const float dd = __fmul_rn(d, d); // float * float, as in PCL
        bad |= !(d >= 0.0f) || !isfinite(dd);
        s += (double)d;
        s2 += (double)dd;
        q1 = min(q1, low_bit_exp(d));
        q2 = min(q2, low_bit_exp(dd));
    }
    for (int o = 32; o > 0; o >>= 1) {
        s += __shfl_xor(s, o);
        s2 += __shfl_xor(s2, o);
        q1 = min(q1, __shfl_xor(q1, o));
        q2 = min(q2, __shfl_xor(q2, o));
        bad |= __shfl_xor(bad, o);
    }
    __shared__ double s_d[4][2];
    __shared__ int s_i[4][3];
    if ((threadIdx.x & 63) == 0) {
        const int w = threadIdx.x >> 6;
        s_d[w][0] = s, s_d[w][1] = s2;
        s_i[w][0] = q1, s_i[w][1] = q2, s_i[w][2] = bad;
    }
    __syncthreads();
    if (threadIdx.x == 0) { // one set of atomics per workgroup; exact sums are order-free (see above)
        atomicAdd(&out->sum, (s_d[0][0] + s_d[1][0]) + (s_d[2][0] + s_d[3][0]));
        atomicAdd(&out->sq_sum, (s_d[0][1] + s_d[1][1]) + (s_d[2][1] + s_d[3][1]));
        atomicMin(&out->q_sum, min(min(s_i[0][0], s_i[1][0]), min(s_i[2][0], s_i[3][0])));
        atomicMin(&out->q_sq, min(min(s_i[0][1], s_i[1][1]), min(s_i[2][1], s_i[3][1])));
        if (s_i[0][2] | s_i[1][2] | s_i[2][2] | s_i[3][2]) atomicOr(&out->bad, 1);
    }
}

// every `step`-th point into a small buffer (the robust extent's sample)
__global__ void k_sample_points(const float *__restrict__ xyz, int64_t step, int S, float *__restrict__ out) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= S) return;
    const size_t i = (size_t)s * (size_t)step;
    out[3 * s] = xyz[3 * i];
    out[3 * s + 1] = xyz[3 * i + 1];
    out[3 * s + 2] = xyz[3 * i + 2];
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

// computePointNormal's covariance from the nine sums, the plane, flipNormalTowardsViewpoint(origin) + the turn toward CamCenter
__device__ float4 normal_from_sums(double a0, double a1, double a2, double a3, double a4, double a5, double a6, double a7, double a8, int cnt, const float4 p,
                                   float cx, float cy, float cz) {
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
    return out;
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
    normals[__float_as_uint(p.w)] = normal_from_sums(a0, a1, a2, a3, a4, a5, a6, a7, a8, cnt, p, cx, cy, cz);
}

// ---- normals on the pixel lattice (round 6).  The radius search of the normals is a window search too: every point within r of
// P lies inside the (2 w + 1)^2 pixels around P's own as soon as the window's bound (k_sor_window's: the distance from P to every
// ray at a pixel distance >= w + 1) exceeds r -- solved for w per point; the lattice copy, with the removed points blanked, replaces
// the sort of the filtered cloud into a grid of r-cells and the walk over 27 of them.
__device__ __forceinline__ int win_need(const float4 P, const WinGeom &g, double r) {
    const double dx = (double)P.x - g.T[0], dy = (double)P.y - g.T[1], dz = (double)P.z - g.T[2];
    const double F2 = fabs(g.rz[0] * dx + g.rz[1] * dy + g.rz[2] * dz), nP = sqrt(dx * dx + dy * dy + dz * dz);
    const double iw = F2 / fabs(g.qz);
    const double coord = fmax(fmax(fabs((double)P.x), fabs((double)P.y)), fabs((double)P.z)) + nP;
    const double rr = (r * (1.0 + 2e-6) + 4e-7 * coord) / (1.0 - 1e-5); // LB (1 - 1e-5) - 4e-7 coord > r (1 + 2e-6): k_sor_window's margins
    if (!(F2 > rr)) return 1 << 20;                                     // LB(w) = F2 (w + 1) / (nP / iw + w + 1) never reaches it
    const double x = rr * (nP / fmax(iw, 1e-300)) / (F2 - rr);          // LB(w) > rr  <=>  w + 1 > x
    return x < (double)(1 << 20) ? (int)x : 1 << 20;
}
__global__ __launch_bounds__(256) void k_normal_need(const float4 *__restrict__ lat, const unsigned int *__restrict__ cell_of, int64_t n, WinGeom g, double r,
                                                      int *__restrict__ wmax) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int need = 0;
    if (i < n) {
        const float4 P = lat[cell_of[i]];
        if (P.x == P.x) need = win_need(P, g, r);
    }
    for (int o = 32; o > 0; o >>= 1) need = max(need, __shfl_xor(need, o));
    if ((threadIdx.x & 63) == 0 && need > *(volatile int *)wmax) atomicMax(wmax, need); // (same-address atomics retire ~88 per microsecond: one per wave was 0.7 ms)
}
__global__ void k_lattice_drop(const unsigned int *__restrict__ flag, const unsigned int *__restrict__ cell_of, int64_t n, float4 *__restrict__ lat) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && !flag[i]) lat[cell_of[i]].x = __uint_as_float(0x7fc00000u);
}
// thread = a point of the filtered cloud; the arithmetic of k_cloud_normals over the window's points (row by row instead of cell by cell)
__global__ __launch_bounds__(256) void k_normals_lattice(const float4 *__restrict__ lat, const unsigned int *__restrict__ cell_of,
                                                          const int32_t *__restrict__ kept_index, int64_t m, WinGeom g, double r, float r2, float cx, float cy,
                                                          float cz, float4 *__restrict__ normals) {
    const int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    float4 p = make_float4(__uint_as_float(0x7fc00000u), 0.0f, 0.0f, 0.0f);
    unsigned int cell = 0;
    if (o < m) {
        cell = cell_of[kept_index[o]];
        p = lat[cell];
    }
    const bool live = p.x == p.x; // (a non-finite point has no lattice entry: its normal stays NaN)
    int w = live ? win_need(p, g, r) : 0;
    for (int t = 32; t > 0; t >>= 1) w = max(w, __shfl_xor(w, t)); // one window per wave: its points are neighbours on a row
    w = min(w, WIN_PAD);
    if (!live) return;
    double a0 = 0, a1 = 0, a2 = 0, a3 = 0, a4 = 0, a5 = 0, a6 = 0, a7 = 0, a8 = 0;
    int cnt = 0;
    for (int b = -w; b <= w; b++) {
        const float4 *row = lat + (size_t)cell + (ptrdiff_t)b * g.gw;
        for (int a = -w; a <= w; a++) {
            const float4 q = row[a];
            if (!(fdist2(p.x, p.y, p.z, q.x, q.y, q.z) < r2)) continue; // (NaN: a pixel without a point, or a removed one)
            const double x = q.x, y = q.y, z = q.z;
            a0 += x * x; a1 += x * y; a2 += x * z; a3 += y * y; a4 += y * z; a5 += z * z;
            a6 += x; a7 += y; a8 += z;
            cnt++;
        }
    }
    normals[o] = normal_from_sums(a0, a1, a2, a3, a4, a5, a6, a7, a8, cnt, p, cx, cy, cz);
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
    // one set of atomics per WORKGROUP (same-address atomics retire at ~88 per microsecond: per wave they were the kernel)
    __shared__ unsigned int s_r[4][7];
    for (int o = 32; o > 0; o >>= 1) nfin += (unsigned int)__shfl_xor((int)nfin, o);
    for (int a = 0; a < 3; a++)
        for (int o = 32; o > 0; o >>= 1) {
            lo[a] = min(lo[a], (unsigned int)__shfl_xor((int)lo[a], o));
            hi[a] = max(hi[a], (unsigned int)__shfl_xor((int)hi[a], o));
        }
    if ((threadIdx.x & 63) == 0) {
        unsigned int *r = s_r[threadIdx.x >> 6];
        r[0] = lo[0], r[1] = lo[1], r[2] = lo[2], r[3] = hi[0], r[4] = hi[1], r[5] = hi[2], r[6] = nfin;
    }
    __syncthreads();
    if (threadIdx.x < 7) {
        const int a = threadIdx.x;
        unsigned int v = s_r[0][a];
        for (int w = 1; w < (int)(blockDim.x >> 6); w++) v = a < 3 ? min(v, s_r[w][a]) : (a < 6 ? max(v, s_r[w][a]) : v + s_r[w][a]);
        if (a < 3) atomicMin(&bb[a], v);
        else if (a < 6) atomicMax(&bb[a], v);
        else if (v) atomicAdd(&bb[6], v);
    }
}

// ------------------------------------------------------------------------------------------------ host side
// Grow-only device arena owned by the context: one hipMalloc sized for the cloud at hand instead of ~25 hipMalloc /
// hipFree pairs per call (hipFree synchronises the device).  Stack discipline: mark() / release().
struct FilterArena {
    char *base = nullptr;
    size_t cap = 0, off = 0;
    void *h_pinned = nullptr; // small pinned staging block (samples, counters, statistics)
    bool failed = false;
    template <typename T>
    T *get(size_t n) {
        const size_t bytes = (n * sizeof(T) + 255) & ~(size_t)255;
        if (off + bytes > cap) {
            failed = true;
            return nullptr;
        }
        T *p = (T *)(base + off);
        off += bytes;
        return p;
    }
};
FilterArena *filter_arena_create() { return new FilterArena(); }
void filter_arena_destroy(FilterArena *a) {
    if (!a) return;
    if (a->base) (void)hipFree(a->base);
    if (a->h_pinned) (void)hipHostFree(a->h_pinned);
    delete a;
}
#define FA_PINNED_BYTES (3 * 8192 * sizeof(float) + 1024)
// room for `bytes` from offset 0 (contents are scratch: nothing survives a call)
int filter_arena_reserve(FilterArena *a, size_t bytes) {
    if (!a->h_pinned && hipHostMalloc(&a->h_pinned, FA_PINNED_BYTES, hipHostMallocDefault) != hipSuccess) return RSM_E_NOMEM;
    a->off = 0;
    a->failed = false;
    if (bytes <= a->cap) return RSM_OK;
    // grows geometrically: hipFree synchronises the whole device, other contexts' pairs in flight included
    const size_t want = std::max(bytes, a->cap + a->cap / 2);
    if (a->base) (void)hipFree(a->base);
    a->base = nullptr;
    a->cap = 0;
    if (hipMalloc((void **)&a->base, want) == hipSuccess) a->cap = want;
    else if (hipMalloc((void **)&a->base, bytes) == hipSuccess) a->cap = bytes;
    else return RSM_E_NOMEM;
    return RSM_OK;
}
size_t filter_arena_bytes(int64_t n) { // upper bound of one filter call's scratch for an n-point cloud (callers add their own buffers)
    return (size_t)n * 112 + (filter_max_cells(n) + 64) * sizeof(int2) + std::min<size_t>((size_t)64 << 20, ((size_t)32 << 20) + (size_t)n * 16) /* sort / scan temporaries, the exhaustive search's 16 MB of histograms */;
}
void *filter_arena_alloc(FilterArena *a, size_t bytes) { return a->get<char>(bytes); }

namespace {
// robust grid: extents from the 1 % .. 99 % quantiles of a sample (far outliers must not set the cell size)
bool sample_extent(FilterArena *A, const float *d_xyz, int64_t n, hipStream_t st, float lo[3], float hi[3]) {
    const int S = (int)std::min<int64_t>(n, 8192);
    const size_t mark = A->off;
    float *d_s = A->get<float>((size_t)3 * S);
    float *h = (float *)A->h_pinned;
    if (!d_s) return false;
    hipLaunchKernelGGL(k_sample_points, dim3((S + 255) / 256), dim3(256), 0, st, d_xyz, n / S, S, d_s);
    if (hipMemcpyAsync(h, d_s, sizeof(float) * 3 * (size_t)S, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess)
        return false;
    A->off = mark;
    for (int a = 0; a < 3; a++) {
        std::vector<float> v;
        for (int s = 0; s < S; s++)
            if (std::isfinite(h[3 * (size_t)s]) && std::isfinite(h[3 * (size_t)s + 1]) && std::isfinite(h[3 * (size_t)s + 2])) v.push_back(h[3 * (size_t)s + a]);
        if (v.empty()) v.push_back(0.0f);
        const size_t q_lo = (size_t)(0.01 * (v.size() - 1)), q_hi = (size_t)(0.99 * (v.size() - 1)); // (two selections: a full sort of the
        std::nth_element(v.begin(), v.begin() + q_lo, v.end());                                        // three samples took 0.8 ms of every call)
        lo[a] = v[q_lo];
        std::nth_element(v.begin() + q_lo, v.begin() + q_hi, v.end());
        hi[a] = v[q_hi];
    }
    return true;
}
} // namespace

struct FilterGridDev {
    unsigned long long *keys = nullptr;
    unsigned int *vals = nullptr; // original index of every sorted point
    float4 *sxyz = nullptr;
    int2 *table = nullptr; // (first, one past last) sorted index per cell (table_kind 1) or per (y, z) row of cells (2)
    int table_kind = 0;    // 0: none, binary search on all keys
    FGrid g{};
};

// sorts the n points of d_xyz by the key of a grid with cell edge h over the box [bb_lo, bb_hi] (points outside fall
// into the border cells: clamping is non-expansive, so two points within h of each other still sit in adjacent cells);
// nv = finite points (they sort first).  Arena space stays allocated until the caller releases its mark.
static int build_grid(FilterArena *A, const float *d_xyz, int64_t n, int64_t nv, float h, const float bb_lo[3], const float bb_hi[3],
                      hipStream_t st, FilterGridDev &G) {
    G.g.inv_h = 1.0f / h;
    auto dim = [&](int a) { return (int)std::min<double>(1 << 20, std::max<double>(1.0, floor((double)(bb_hi[a] - bb_lo[a]) / h) + 1.0)); };
    int ax[3] = {0, 1, 2};
    std::sort(ax, ax + 3, [&](int a, int b) { return dim(a) != dim(b) ? dim(a) > dim(b) : a < b; }); // most cells first = fastest key digit
    G.g.p0 = ax[0], G.g.p1 = ax[1], G.g.p2 = ax[2];
    G.g.ox = bb_lo[ax[0]], G.g.oy = bb_lo[ax[1]], G.g.oz = bb_lo[ax[2]];
    G.g.nx = dim(ax[0]), G.g.ny = dim(ax[1]), G.g.nz = dim(ax[2]);
    unsigned long long *k1 = A->get<unsigned long long>((size_t)n), *k2 = A->get<unsigned long long>((size_t)n);
    unsigned int *v1 = A->get<unsigned int>((size_t)n), *v2 = A->get<unsigned int>((size_t)n);
    G.sxyz = A->get<float4>((size_t)n);
    if (!k1 || !k2 || !v1 || !v2 || !G.sxyz) return RSM_E_NOMEM;
    const unsigned blocks = (unsigned)((n + 255) / 256);
    hipLaunchKernelGGL(k_cell_keys, dim3(blocks), dim3(256), 0, st, d_xyz, n, G.g, k1, v1);
    const double ncell = (double)G.g.nx * G.g.ny * G.g.nz;
    int key_bits = 1;
    while (key_bits < 64 && ncell > (double)(1ull << key_bits)) key_bits++;
    key_bits = 64; // non-finite points carry the key ~0: all bits take part
    size_t tmp_bytes = 0;
    if (rocprim::radix_sort_pairs(nullptr, tmp_bytes, k1, k2, v1, v2, (size_t)n, 0, key_bits, st) != hipSuccess) return RSM_E_HIP;
    void *tmp = A->get<uint8_t>(tmp_bytes);
    if (!tmp) return RSM_E_NOMEM;
    if (rocprim::radix_sort_pairs(tmp, tmp_bytes, k1, k2, v1, v2, (size_t)n, 0, key_bits, st) != hipSuccess) return RSM_E_HIP;
    hipLaunchKernelGGL(k_gather_sorted, dim3(blocks), dim3(256), 0, st, d_xyz, v2, n, G.sxyz);
    G.keys = k2;
    G.vals = v2;
    G.table = nullptr;
    G.table_kind = 0;
    const double nrow = (double)G.g.ny * G.g.nz;
    const double max_cells = (double)filter_max_cells(n);
    if (nv > 0 && (ncell <= max_cells || nrow <= max_cells)) {
        G.table_kind = ncell <= max_cells ? 1 : 2;
        const size_t nc = (size_t)(G.table_kind == 1 ? ncell : nrow);
        G.table = A->get<int2>(nc);
        if (!G.table) return RSM_E_NOMEM;
        if (hipMemsetAsync(G.table, 0, sizeof(int2) * nc, st) != hipSuccess) return RSM_E_HIP;
        hipLaunchKernelGGL(k_cell_table, dim3((unsigned)((nv + 255) / 256)), dim3(256), 0, st, k2, (int)nv,
                           (unsigned long long)(G.table_kind == 1 ? 1 : G.g.nx), G.table);
    }
    return RSM_OK;
}

// (pixel-window pass) every finite point starts undecided; the others take no part in the searches (distance 0, as PCL)
__global__ void k_finite_flags(const float *__restrict__ xyz, int64_t n, unsigned int *__restrict__ flag, unsigned int *__restrict__ iota) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    flag[i] = (isfinite(xyz[3 * i]) && isfinite(xyz[3 * i + 1]) && isfinite(xyz[3 * i + 2])) ? 1u : 0u;
    iota[i] = (unsigned int)i;
}

__global__ void k_compact_list(const unsigned int *__restrict__ src, const unsigned int *__restrict__ flag, const unsigned int *__restrict__ pos,
                               int n, unsigned int *__restrict__ dst, int *__restrict__ count) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (flag[i]) dst[pos[i]] = src[i];
    if (i == n - 1) *count = (int)(pos[i] + flag[i]);
}

static void launch_knn(const float *d_xyz, const FilterGridDev &G, int nv, float h, int mean_k, const unsigned int *queries, int nq,
                       float *d_dist, unsigned int *d_cnt /* undecided flags */, hipStream_t st) {
    const dim3 grid((unsigned)((nq + 3) / 4));
    if (G.table_kind == 1)
        hipLaunchKernelGGL(k_sor_knn<1>, grid, dim3(256), 0, st, d_xyz, G.sxyz, G.keys, G.table, nv, G.g, h * h, mean_k, queries, nq, d_dist, d_cnt);
    else if (G.table_kind == 2)
        hipLaunchKernelGGL(k_sor_knn<2>, grid, dim3(256), 0, st, d_xyz, G.sxyz, G.keys, G.table, nv, G.g, h * h, mean_k, queries, nq, d_dist, d_cnt);
    else
        hipLaunchKernelGGL(k_sor_knn<0>, grid, dim3(256), 0, st, d_xyz, G.sxyz, G.keys, G.table, nv, G.g, h * h, mean_k, queries, nq, d_dist, d_cnt);
}

// d_xyz: n x 3 float (device).  Outputs (device): kept_index [n] (first *n_kept valid), fxyz [3n], normals [n] float4.
// A: reserved by the caller (filter_arena_reserve) with at least filter_arena_bytes(n) free.
// pre (optional): the cloud is the depth map of a matched pair -- its pixel lattice decides most queries (k_sor_window),
// the grid ladder only sees the rest.
int filter_cloud_device(FilterArena *A, const float *d_xyz, int64_t n, int mean_k, double std_mul, double normal_radius,
                        const float cam_center[3], int32_t *d_kept_index, float *d_fxyz, float4 *d_normals, int64_t *n_kept,
                        double stats[4], hipStream_t st, const FilterLattice *pre) {
    *n_kept = 0;
    if (n <= 0) return RSM_OK;
    if (n >= (1ll << 31) || mean_k < 1 || !A) return RSM_E_INVALID;
    float lo[3], hi[3], flo[3], fhi[3];
    // exact bounding box (the sample's extremes are not the cloud's) and the number of finite points: only they take
    // part in the searches (PCL: the k-d tree skips the others)
    unsigned int *d_bb = A->get<unsigned int>(8);
    int *d_cnt = A->get<int>(4);
    DistStats *d_stats = A->get<DistStats>(1);
    float *d_dist = A->get<float>((size_t)n);
    unsigned int *d_redo = A->get<unsigned int>((size_t)n), *d_redo2 = A->get<unsigned int>((size_t)n);
    unsigned int *d_flag = A->get<unsigned int>((size_t)n), *d_pos = A->get<unsigned int>((size_t)n);
    if (A->failed) return RSM_E_NOMEM;
    unsigned int *h_bb = (unsigned int *)((char *)A->h_pinned + 3 * 8192 * sizeof(float));
    int *h_cnt = (int *)(h_bb + 8);
    DistStats *h_stats = (DistStats *)(h_cnt + 4);
    int64_t nv = 0;
    {
        const unsigned int init[7] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0u, 0u, 0u, 0u};
        memcpy(h_bb, init, sizeof init);
        if (hipMemcpyAsync(d_bb, h_bb, sizeof init, hipMemcpyHostToDevice, st) != hipSuccess) return RSM_E_HIP;
        hipLaunchKernelGGL(k_bbox, dim3((unsigned)std::min<int64_t>((n + 255) / 256, 1024)), dim3(256), 0, st, d_xyz, n, d_bb);
        if (hipMemcpyAsync(h_bb, d_bb, sizeof init, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return RSM_E_HIP;
        nv = h_bb[6];
        for (int a = 0; a < 3; a++) {
            flo[a] = nv ? ord_to_float(h_bb[a]) : 0.0f;
            fhi[a] = nv ? ord_to_float(h_bb[3 + a]) : 0.0f;
        }
    }
    // first cell edge: ~sqrt(k + 1) point spacings of a surface patch whose area is the product of the two largest
    // robust extents; queries that cannot be decided inside their 27 cells (fewer than k + 1 points within h: thick or
    // sparse parts of the cloud, patch corners, isolated points) are retried on a grid with twice the edge; what is left
    // after KNN_LEVELS grids (or once only a handful remain) is searched against all points by the whole chip.
    // One cell edge does not fit a whole cloud: a perspective depth map is 25 times denser (per unit of space) in its near
    // part than in its far part (C2: 1.25 vs 6 units between neighbours), and the surface estimate above is off for a deep
    // or thick one.  The search therefore runs over a ladder of grids from fine to coarse, h doubling: a query is decided
    // at the first level whose 27 cells hold its k + 1 nearest (there its candidates number a few hundred whatever the
    // local density); an undecided attempt costs one table lookup and one count over the few candidates a too-fine grid
    // offers.  Every level re-sorts the points (0.65 ms for 5.5 M): cheap next to one query pass with a wrong h
    // (a single level at the surface estimate ran the near part's queries over ~50 000 candidates each: 390 ms).
    // The ladder starts one octave below the surface estimate (on C2 two octaves below decides 10 % of the queries for
    // a quarter of the search time, three octaves below nothing).
    // All of it only when a grid is built at all: the cloud of a matched pair is searched on its pixel lattice and the sample, its
    // round trip and its quantiles stay out of the call.
    float h = 0.0f, h_floor = 0.0f, glo[3] = {0.0f, 0.0f, 0.0f}, ghi[3] = {0.0f, 0.0f, 0.0f};
    bool have_grid_box = false;
    auto grid_box = [&]() -> bool {
        if (have_grid_box) return true;
        if (!sample_extent(A, d_xyz, n, st, lo, hi)) return false;
        double e[3] = {(double)hi[0] - lo[0], (double)hi[1] - lo[1], (double)hi[2] - lo[2]};
        std::sort(e, e + 3);
        const double area = std::max(e[2] * e[1], 1e-12), spacing = sqrt(area / (0.98 * 0.98 * 0.98 * (double)std::max<int64_t>(nv, 1)));
        h = (float)(spacing * sqrt((double)(mean_k + 1)));
        if (!(h > 0.0f) || !std::isfinite(h)) h = 1.0f;
        // the grid box: the robust extent grown by a few cells, inside the exact bounding box (far outliers are clamped
        // into the border cells instead of blowing up the cell count)
        for (int a = 0; a < 3; a++) {
            glo[a] = std::max(flo[a], lo[a] - 4.0f * h);
            ghi[a] = std::min(fhi[a], hi[a] + 4.0f * h);
            if (!(ghi[a] >= glo[a])) ghi[a] = glo[a];
        }
        h *= 0.5f;
        if (h_floor > h) h = h_floor;
        have_grid_box = true;
        return true;
    };
    const unsigned blocks = (unsigned)((n + 255) / 256);
    const int KNN_LEVELS = 12;
    if (hipMemsetAsync(d_dist, 0, sizeof(float) * (size_t)n, st) != hipSuccess) return RSM_E_HIP; // non-finite points: distance 0, as PCL
    int nq = (int)nv, redo_n = 0;
    const unsigned int *queries = nullptr;
    int s = RSM_OK;
    const size_t mark = A->off;
    bool prepassed = false, wave_done = false;
    float4 *lat_keep = nullptr;          // the lattice copy, kept for the normals while no grid level has taken its arena space
    const unsigned int *cell_keep = nullptr;
    size_t mark_lat = mark;
    const float4 *lat_all = nullptr;
    int64_t lat_cells = 0;
    if (pre && mean_k <= 128 && nv > 0) { // the pixel-window pass: what it cannot decide becomes the ladder's first query list
        float4 *lat = A->get<float4>(cloud_lattice_bytes(pre->XL, pre->XR, pre->YL, pre->YR) / sizeof(float4));
        unsigned int *cell_of = A->get<unsigned int>((size_t)n);
        if (!lat || !cell_of) return RSM_E_NOMEM;
        lat_keep = lat;
        cell_keep = cell_of;
        mark_lat = A->off;
        hipLaunchKernelGGL(k_finite_flags, dim3(blocks), dim3(256), 0, st, d_xyz, n, d_flag, d_redo2);
        launch_cloud_lattice(pre->flags, pre->row_offset, pre->W, pre->XL, pre->XR, pre->YL, pre->YR, pre->xyz64, n, lat, cell_of, st);
        // which window?  pre->radius > 0: the caller's; 0: the smallest of 7 / 12 / 16 pixels that decides >= 85 % of a sparse
        // sample of the tiles (every 6th in x and y: ~3 % of the queries) -- a thin sheet (depth noise below the lateral
        // spacing, the reference's rig geometry) is decided inside 15 x 15 pixels, a thick one (the synthetic bench rig: a
        // disparity step of 0.05 pixel is 6 lateral spacings deep) needs 33 x 33; nothing decides enough: no window pass
        int radius = pre->radius;
        if (radius <= 0) {
            unsigned int *d_pc = (unsigned int *)d_cnt; // (d_cnt has 4 ints)
            unsigned int *h_pc = (unsigned int *)h_cnt;
            const int cand[3] = {7, 12, 16};
            double best = 0.0;
            for (int ci = 0; ci < 3 && radius <= 0; ci++) {
                if (hipMemsetAsync(d_pc, 0, 3 * sizeof(unsigned int), st) != hipSuccess) return RSM_E_HIP;
                launch_sor_window(lat, pre->XL, pre->XR, pre->YL, pre->YR, pre->qz, pre->R, pre->T, mean_k, cand[ci], d_dist, d_flag, st, 6, d_pc);
                if (hipMemcpyAsync(h_pc, d_pc, 3 * sizeof(unsigned int), hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess)
                    return RSM_E_HIP;
                best = h_pc[0] ? (double)h_pc[1] / (double)h_pc[0] : 0.0;
                if (best >= 0.85 || (ci == 2 && best >= 0.5)) {
                    radius = cand[ci];
                    // every query this window leaves over has its (k+1)-th neighbour beyond the window's bound (closer points
                    // cannot hide outside it): the ladder need not start below that scale
                    float sum_lim;
                    memcpy(&sum_lim, &h_pc[2], sizeof sum_lim);
                    // (1.25 x the mean bound: a level at the bound itself would decide almost none of them)
                    const float hf = 1.25f * sum_lim / (float)h_pc[0];
                    if (std::isfinite(hf)) h_floor = hf;
                }
            }
        }
        if (pre->radius_out) *pre->radius_out = radius > 0 ? radius : 0;
        if (radius > 0) {
            launch_sor_window(lat, pre->XL, pre->XR, pre->YL, pre->YR, pre->qz, pre->R, pre->T, mean_k, radius, d_dist, d_flag, st);
            size_t tb = 0;
            if (rocprim::exclusive_scan(nullptr, tb, d_flag, d_pos, 0u, (size_t)n, rocprim::plus<unsigned int>(), st) != hipSuccess) return RSM_E_HIP;
            void *tp = A->get<uint8_t>(tb);
            if (!tp) return RSM_E_NOMEM;
            if (rocprim::exclusive_scan(tp, tb, d_flag, d_pos, 0u, (size_t)n, rocprim::plus<unsigned int>(), st) != hipSuccess) return RSM_E_HIP;
            hipLaunchKernelGGL(k_compact_list, dim3(blocks), dim3(256), 0, st, d_redo2, d_flag, d_pos, (int)n, d_redo, d_cnt);
            if (hipMemcpyAsync(h_cnt, d_cnt, sizeof(int), hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return RSM_E_HIP;
            nq = h_cnt[0];
            queries = d_redo;
            prepassed = true;
            if (pre->tile_left_out) *pre->tile_left_out = nq;
            // What a window below 24 pixels leaves over (a thick sheet's points whose neighbours spread wider: scattered, a seventh
            // of C2's cloud) gets the 49 x 49 window a thread each, straight from the lattice copy (k_sor_window_list); only what
            // that leaves as well -- depth edges, holes, outliers -- goes to the grid ladder.
            // What a window below 24 pixels leaves over (a thick sheet's points whose neighbours spread wider: scattered, a seventh
            // of C2's cloud) gets the 49 x 49 window a thread each, straight from the lattice copy (k_sor_window_list); what THAT
            // leaves -- the points within a dozen pixels of the mask's border and of depth edges, whose neighbours lie on one
            // side: 1.6 % of C2's cloud, the queries that cost the grid ladder most (coarse levels: tens of thousands of
            // candidates in the 27 cells of each) -- an 81 x 81 window; only the rest (holes, outliers, islands) goes to the ladder.
            unsigned int *cur = d_redo, *oth = d_redo2;
            const int radii[2] = {24, 40};
            int last = radius;
            for (int pass = 0; pass < 2 && nq > 0; pass++) {
                if (radii[pass] <= last || !((pre->list_pass >> pass) & 1)) continue;
                if ((pre->list_pass >> (3 + pass)) & 1) // (A/B: the wave form at this radius instead of the thread form)
                    launch_sor_window_wave(lat, cell_of, pre->XL, pre->XR, pre->YL, pre->YR, pre->qz, pre->R, pre->T, mean_k, radii[pass], cur, nq, d_dist, d_flag, st, pre->wg_max);
                else
                    launch_sor_window_list(lat, cell_of, pre->XL, pre->XR, pre->YL, pre->YR, pre->qz, pre->R, pre->T, mean_k, radii[pass], cur, nq, d_dist, d_flag, st);
                if (rocprim::exclusive_scan(tp, tb, d_flag, d_pos, 0u, (size_t)nq, rocprim::plus<unsigned int>(), st) != hipSuccess) return RSM_E_HIP;
                hipLaunchKernelGGL(k_compact_list, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, st, cur, d_flag, d_pos, nq, oth, d_cnt);
                if (hipMemcpyAsync(h_cnt, d_cnt, sizeof(int), hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return RSM_E_HIP;
                nq = h_cnt[0];
                std::swap(cur, oth);
                last = radii[pass]; // (the ladder keeps its start: what is left now is few, and a level whose 27 cells hold tens of
                                    // thousands of candidates costs a wave-per-query search milliseconds however few the queries)
            }
            // What the list passes leave (C2: 564 of 5.5 M -- mask corners, hole rims, outliers): a wave each over windows of 80, 160,
            // 320 ... pixels of the lattice copy, while there are few enough for that to be cheaper than a grid level (a level is a
            // sort of the whole cloud plus, for these queries, 27 cells of tens of thousands of candidates each: 4 ms a level on C2).
            if ((pre->list_pass & 4) && last >= 24) {
                const int span = std::max(pre->XR - pre->XL, pre->YR - pre->YL) + 1;
                for (int wr = std::max(80, 2 * last); nq > 0 && nq <= 65536; wr *= 2) {
                    launch_sor_window_wave(lat, cell_of, pre->XL, pre->XR, pre->YL, pre->YR, pre->qz, pre->R, pre->T, mean_k, wr, cur, nq, d_dist, d_flag, st, pre->wg_max);
                    if (rocprim::exclusive_scan(tp, tb, d_flag, d_pos, 0u, (size_t)nq, rocprim::plus<unsigned int>(), st) != hipSuccess) return RSM_E_HIP;
                    hipLaunchKernelGGL(k_compact_list, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, st, cur, d_flag, d_pos, nq, oth, d_cnt);
                    if (hipMemcpyAsync(h_cnt, d_cnt, sizeof(int), hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return RSM_E_HIP;
                    nq = h_cnt[0];
                    std::swap(cur, oth);
                    if (wr >= span) break; // the window already covers the whole box: what is left has fewer than k + 1 points within its bound
                }
                wave_done = true;
            }
            queries = cur;
            if (pre->undecided_out) *pre->undecided_out = nq;
            lat_all = lat;
            lat_cells = (int64_t)(pre->XR - pre->XL + 1 + 2 * WIN_PAD) * (int64_t)(pre->YR - pre->YL + 1 + 2 * WIN_PAD);
        }
    }
    FilterGridDev G;
    G.sxyz = nullptr;
    // after the wave passes a handful is left at most (isolated points): straight to the whole-chip search, over the lattice copy
    // as the point array (its empty pixels are NaN: never below any threshold) -- no grid level, no sort
    const bool skip_ladder = wave_done && nq <= 48 && lat_all != nullptr;
    if (nq > 0 && !skip_ladder && !grid_box()) return RSM_E_HIP;
    for (int level = 0; level < KNN_LEVELS && nq > 0 && !skip_ladder; level++, h *= 2.0f) {
        A->off = mark; // the previous level's grid is done (its kernels are ordered before this level's on the stream)
        lat_keep = nullptr;
        s = build_grid(A, d_xyz, n, nv, h, glo, ghi, st, G);
        if (s != RSM_OK) return s;
        if (level == 0 && !prepassed) queries = G.vals; // every point, in grid order (coherent waves)
        unsigned int *out_list = (queries == d_redo) ? d_redo2 : d_redo; // never the list being read
        launch_knn(d_xyz, G, (int)nv, h, mean_k, queries, nq, d_dist, d_flag, st);
        { // undecided queries -> the next level's list, in order
            size_t tb = 0;
            if (rocprim::exclusive_scan(nullptr, tb, d_flag, d_pos, 0u, (size_t)nq, rocprim::plus<unsigned int>(), st) != hipSuccess) return RSM_E_HIP;
            void *tp = A->get<uint8_t>(tb);
            if (!tp) return RSM_E_NOMEM;
            if (rocprim::exclusive_scan(tp, tb, d_flag, d_pos, 0u, (size_t)nq, rocprim::plus<unsigned int>(), st) != hipSuccess) return RSM_E_HIP;
            hipLaunchKernelGGL(k_compact_list, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, st, queries, d_flag, d_pos, nq, out_list, d_cnt);
        }
        if (hipMemcpyAsync(h_cnt, d_cnt, sizeof(int), hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess)
            return RSM_E_HIP;
        redo_n = h_cnt[0];
        queries = out_list;
        nq = redo_n;
        if (nq <= 48) break; // cheaper to finish against all points than to sort again (a coarser level costs ~0.8 ms, the
                             // whole-chip search ~25 us per query)
    }
    redo_n = nq;
    if (nq > 0) { // against all points; the last grid's sorted copy is still in the arena (any ordering of the points serves)
        const int XQ_BATCH = 2048; // queries per round (their histograms: 16 MB)
        XqState *d_xq = A->get<XqState>((size_t)XQ_BATCH);
        int *d_hist = A->get<int>((size_t)XQ_BATCH * 2048);
        if (!d_xq || !d_hist) return RSM_E_NOMEM;
        if (hipMemsetAsync(d_hist, 0, sizeof(int) * (size_t)XQ_BATCH * 2048, st) != hipSuccess) return RSM_E_HIP;
        const int shifts[3] = {20, 9, 0}, widths[3] = {11, 11, 9};
        const float4 *arr = skip_ladder ? lat_all : G.sxyz; // any array holding every finite point once
        const int arr_n = skip_ladder ? (int)lat_cells : (int)nv;
        if (!arr || lat_cells >= (1ll << 31)) return RSM_E_INVALID;
        for (int q0 = 0; q0 < nq; q0 += XQ_BATCH) {
            const int cnt = std::min(XQ_BATCH, nq - q0), qy = std::min(cnt, 1024);
            const unsigned int *list = queries + q0;
            hipLaunchKernelGGL(k_xq_init, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, st, d_xq, cnt, mean_k, (int)nv);
            for (int pass = 0; pass < 3; pass++) {
                hipLaunchKernelGGL(k_xq_hist, dim3(XQ_SLICES, (unsigned)qy), dim3(256), 0, st, d_xyz, arr, arr_n, list, cnt, d_xq, shifts[pass],
                                   widths[pass], d_hist);
                hipLaunchKernelGGL(k_xq_pick, dim3((unsigned)cnt), dim3(64), 0, st, d_xq, cnt, shifts[pass], widths[pass], d_hist);
            }
            hipLaunchKernelGGL(k_xq_sum, dim3(XQ_SLICES, (unsigned)qy), dim3(256), 0, st, d_xyz, arr, arr_n, list, cnt, d_xq);
            hipLaunchKernelGGL(k_xq_finish, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, st, d_xq, list, cnt, mean_k, (int)nv, d_dist);
        }
    }
    A->off = lat_keep ? mark_lat : mark;
    // mean / stddev exactly as PCL forms them (a sequential loop over the per-point distances in point order): on the
    // device when no addition can round (k_dist_stats), else on the host
    double sum = 0.0, sq_sum = 0.0;
    {
        DistStats init{0.0, 0.0, 0x7fffffff, 0x7fffffff, 0};
        *h_stats = init;
        if (hipMemcpyAsync(d_stats, h_stats, sizeof init, hipMemcpyHostToDevice, st) != hipSuccess) return RSM_E_HIP;
        hipLaunchKernelGGL(k_dist_stats, dim3((unsigned)std::min<int64_t>((n + 255) / 256, 1024)), dim3(256), 0, st, d_dist, n, d_stats);
        h_cnt[3] = 1 << 20;
        if (lat_keep && d_normals && pre->normals_wmax > 0) { // the widest window a normal's radius search needs on the lattice (read back with the statistics)
            if (hipMemsetAsync(d_cnt + 3, 0, sizeof(int), st) != hipSuccess) return RSM_E_HIP;
            hipLaunchKernelGGL(k_normal_need, dim3(blocks), dim3(256), 0, st, lat_keep, cell_keep, n, win_geom(pre->XL, pre->XR, pre->YL, pre->YR, pre->qz, pre->R, pre->T),
                               normal_radius, d_cnt + 3);
            if (hipMemcpyAsync(h_cnt + 3, d_cnt + 3, sizeof(int), hipMemcpyDeviceToHost, st) != hipSuccess) return RSM_E_HIP;
        }
        if (hipMemcpyAsync(h_stats, d_stats, sizeof init, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess)
            return RSM_E_HIP;
        auto exact = [](double total, int q) { // every partial sum is a multiple of 2^q below 2^(q + 53)
            if (q == 0x7fffffff) return true;   // all zero
            return total * (1.0 + 1e-9) < ldexp(1.0, q + 53);
        };
        if (!h_stats->bad && exact(h_stats->sum, h_stats->q_sum) && exact(h_stats->sq_sum, h_stats->q_sq)) {
            sum = h_stats->sum;
            sq_sum = h_stats->sq_sum;
        } else {
            std::vector<float> hd((size_t)n);
            if (hipMemcpyAsync(hd.data(), d_dist, sizeof(float) * (size_t)n, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess)
                return RSM_E_HIP;
            for (int64_t i = 0; i < n; i++) {
                sum += hd[(size_t)i];
                sq_sum += hd[(size_t)i] * hd[(size_t)i]; // float * float, as in PCL
            }
        }
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
    hipLaunchKernelGGL(k_keep_flags, dim3(blocks), dim3(256), 0, st, d_dist, n, thr, d_flag);
    size_t tb = 0;
    if (rocprim::exclusive_scan(nullptr, tb, d_flag, d_pos, 0u, (size_t)n, rocprim::plus<unsigned int>(), st) != hipSuccess) return RSM_E_HIP;
    void *tp = A->get<uint8_t>(tb);
    if (!tp) return RSM_E_NOMEM;
    if (rocprim::exclusive_scan(tp, tb, d_flag, d_pos, 0u, (size_t)n, rocprim::plus<unsigned int>(), st) != hipSuccess) return RSM_E_HIP;
    hipLaunchKernelGGL(k_compact_kept, dim3(blocks), dim3(256), 0, st, d_xyz, d_flag, d_pos, n, d_fxyz, d_kept_index);
    unsigned int *h_last = (unsigned int *)h_cnt;
    if (hipMemcpyAsync(&h_last[0], d_pos + (n - 1), 4, hipMemcpyDeviceToHost, st) != hipSuccess ||
        hipMemcpyAsync(&h_last[1], d_flag + (n - 1), 4, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess)
        return RSM_E_HIP;
    const int64_t m = (int64_t)h_last[0] + h_last[1];
    *n_kept = m;
    A->off = lat_keep ? mark_lat : mark;
    if (m == 0 || !d_normals) {
        A->off = mark;
        return RSM_OK;
    }
    // normals of the filtered cloud.  On the lattice copy when it is still there and a window of at most pre->normals_wmax pixels holds every
    // normal's neighbourhood (k_normal_need: a property of the rig -- the search radius in pixel spacings at the nearest point)
    if (pre && pre->normals_out) {
        pre->normals_out[0] = 0;
        pre->normals_out[1] = lat_keep ? h_cnt[3] : -1;
    }
    if (lat_keep && h_cnt[3] <= pre->normals_wmax) {
        if (pre->normals_out) pre->normals_out[0] = std::max(h_cnt[3], 1);
        const WinGeom g = win_geom(pre->XL, pre->XR, pre->YL, pre->YR, pre->qz, pre->R, pre->T);
        hipLaunchKernelGGL(k_lattice_drop, dim3(blocks), dim3(256), 0, st, d_flag, cell_keep, n, lat_keep);
        if (hipMemsetD32Async((hipDeviceptr_t)d_normals, 0x7fc00000, (size_t)4 * m, st) != hipSuccess) return RSM_E_HIP;
        hipLaunchKernelGGL(k_normals_lattice, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, st, lat_keep, cell_keep, d_kept_index, m, g, normal_radius,
                           (float)(normal_radius * normal_radius), cam_center[0], cam_center[1], cam_center[2], d_normals);
        if (hipStreamSynchronize(st) != hipSuccess || hipGetLastError() != hipSuccess) return RSM_E_HIP;
        A->off = mark;
        return RSM_OK;
    }
    A->off = mark;
    // normals of the filtered cloud: grid with cell edge = search radius; the non-finite points (all kept: distance 0)
    // sort behind the finite ones and keep the NaN normal the buffer is filled with
    FilterGridDev G2;
    const int64_t mv = m - (n - nv);
    s = build_grid(A, d_fxyz, m, 0 /* no table: the normals walk their ranges by binary search */, (float)normal_radius, flo, fhi, st, G2);
    if (s != RSM_OK) return s;
    const float r2 = (float)(normal_radius * normal_radius);
    if (hipMemsetD32Async((hipDeviceptr_t)d_normals, 0x7fc00000, (size_t)4 * m, st) != hipSuccess) return RSM_E_HIP;
    if (mv > 0)
        hipLaunchKernelGGL(k_cloud_normals, dim3((unsigned)((mv + 255) / 256)), dim3(256), 0, st, G2.sxyz, G2.keys, (int)mv, G2.g, r2,
                           cam_center[0], cam_center[1], cam_center[2], d_normals);
    if (hipStreamSynchronize(st) != hipSuccess || hipGetLastError() != hipSuccess) return RSM_E_HIP;
    A->off = mark;
    return RSM_OK;
}

void launch_f64_to_f32x3(const double *src, int64_t n, float *dst, hipStream_t st) {
    if (n > 0) hipLaunchKernelGGL(k_f64_to_f32x3, dim3((unsigned)((3 * n + 255) / 256)), dim3(256), 0, st, src, 3 * n, dst);
}
void launch_pack_filtered16(const double *xyz, const uint8_t *bgr, const int32_t *kept, int64_t m, void *dst, hipStream_t st) {
    if (m > 0) hipLaunchKernelGGL(k_pack_filtered16, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, st, xyz, bgr, kept, m, (uint4 *)dst);
}
