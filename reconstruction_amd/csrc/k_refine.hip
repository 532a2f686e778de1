// k_refine.hip -- CStereoMatching::DisparityRefine, reconstruction/CStereoMatching.cpp:572-680.
//
// fp64 Jacobi sweeps (30 + 30*level of them, .cpp:95) over the interior of the own margin.
// Per pixel and sweep the reference recomputes three 3x3x3 NCC values at the integer shifts
// iMatch + {0,1,2}, iMatch = int(dCenter - 1.5) + x (.cpp:625-630), and derives the data term
//   index 0: pwp = xi1-xi0, pdp = dCenter - 0.5      index 2: pwp = xi1-xi2, pdp = dCenter + 0.5
//   index 1: pwp = (xi0+xi2)/2 - xi1, pdp = dCenter + 0.5*(xi0-xi2)/(xi0+xi2-2*xi1), 0 if pwp == 0
// (pwp, pdp - dCenter) depend only on (x, y, iMatch), not on the sweep: they are cached per pixel
// keyed by iMatch and recomputed only when int(dCenter - 1.5) changes -- the same fp64 expression
// tree is evaluated either way, so the sweep result is bit-identical to recomputing every time.
// The three NCC values are computed on a cache miss only, as a bit-faithful fp64 restatement of the
// reference's WindowToVec + arma::dot (refine_common.h).
// Right-window reads have no bounds check in the reference (.cpp:628): emulated on the flat
// row-major buffer, bytes outside the whole image read 0 (same rule as oracle/stereo_oracle.c).
//
// The cache (DirArgs::rf_key / rf_ent): two ways per pixel, indexed by the parity of iMatch - x (the settled iteration flips
// between two ADJACENT iMatch values).  A way's entry is one 16-byte record (pwp, delta) in rf_ent[way * rf_stride + pixel];
// both ways' keys (iMatch - x as int16, RF_NOKEY = empty) share the dword rf_key[pixel] (way 0 in the low half) and are
// written as separate 16-bit halves.
//
// This file: the first sweep (k_refine_first), the single sweep (k_refine_sweep, used while the cache fills and on the small
// levels), the scatter of a time-skewed launch's update list (k_refine_apply), and the test entries.  The settled sweeps of
// the large levels run T per launch in k_refine_skew.hip.
#include "refine_common.h"

// test entry: the specified exp on an array (rsm_stage_exp_neg); flag = 1: the t < 512 form on every argument below 512
__global__ void k_exp_neg(const double *t, double *out, long long n, int small_form) {
    __shared__ double2 s_exp[128];
    exp_tab_stage(s_exp);
    __syncthreads();
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (small_form && t[i] < 512.0) ? exp_neg_small(t[i], s_exp) : exp_neg(t[i], s_exp);
}
void launch_exp_neg(const double *t, double *out, long long n, hipStream_t st, int small_form) {
    if (n > 0) hipLaunchKernelGGL(k_exp_neg, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, t, out, n, small_form);
}

__global__ void k_refine_init(StageArgs a) {
    const DirArgs &d = a.d[blockIdx.z];
    const size_t n = (size_t)a.W * a.H;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        const double v = (double)d.d16_in[i]; // convertTo CV_64F, .cpp:585
        d.f64_a[i] = v;
        d.f64_b[i] = v; // copyTo, .cpp:587
        d.rf_key[i] = RF_NOKEY2; // both ways empty
    }
}

void launch_refine_init(const StageArgs &a, hipStream_t st) {
    const size_t n = (size_t)a.W * a.H;
    size_t blocks = (n + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(k_refine_init, dim3((unsigned)blocks, 1, a.ndir), dim3(256), 0, st, a);
}

// test entry (rsm_stage_div_unscaled): the trimmed division beside the compiler's on arrays
__global__ void k_div_unscaled(const double *a, const double *b, double *q_fast, double *q_ieee, long long n) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        q_fast[i] = div_unscaled(a[i], b[i]);
        q_ieee[i] = a[i] / b[i];
    }
}
void launch_div_unscaled(const double *a, const double *b, double *q_fast, double *q_ieee, long long n, hipStream_t st) {
    if (n > 0) hipLaunchKernelGGL(k_div_unscaled, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, a, b, q_fast, q_ieee, n);
}

// test entry (rsm_stage_refine_xi): the matching costs xi (.cpp:624-629) as each of the three device restatements of the data
// term computes them, for every row y in [1, H-1), own column x in [1, W-1) and other-view window left edge col in [0, W-3]:
// out[c][entry] = xi(x, y, col + c), c = 0..2, entry = ((y-1) (W-2) + (x-1)) (W-2) + col.  form 0: refine_left + refine_cost_left
// (k_refine_first), 1: refine_data_term_packed (the sweep kernels' lane-per-miss service), 2: refine_data_term_quad (four lanes
// per miss: k_refine_sweep, k_refine_skew).
__global__ void k_refine_xi(const uint32_t *A, const uint32_t *B, int W, int H, int form, double *out, long long n) {
    const long long tid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long e = form == 2 ? (tid >> 2) : tid;
    const long long ec = e < n ? e : n - 1; // (whole quads stay active: the quad routine shuffles)
    const int nw = W - 2;
    const int col = (int)(ec % nw), x = 1 + (int)((ec / nw) % nw), y = 1 + (int)(ec / ((long long)nw * nw));
    double xs[3] = {0.0, 0.0, 0.0};
    if (form == 0) {
        RfLeft L;
        refine_left(A, W, x, y, L);
        for (int c = 0; c < 3; c++) xs[c] = refine_cost_left(L, B, W, H, y, col + c);
    } else if (form == 1) {
        double p, q;
        refine_data_term_packed(A, B, W, H, x, y, col, p, q, xs);
    } else {
        double p, q, own = 0.0;
        refine_data_term_quad(A, B, W, H, x, y, col, (int)(tid & 3), p, q, &own);
        const int qd = (int)(tid & 3);
        if (e < n && qd >= 1) out[(long long)(qd - 1) * n + e] = own;
        return;
    }
    if (e < n)
        for (int c = 0; c < 3; c++) out[(long long)c * n + e] = xs[c];
}
void launch_refine_xi(const uint32_t *A, const uint32_t *B, int W, int H, int form, double *out, hipStream_t st) {
    const long long n = (long long)(H - 2) * (W - 2) * (W - 2);
    if (n <= 0) return;
    const long long threads = form == 2 ? 4 * n : n;
    hipLaunchKernelGGL(k_refine_xi, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st, A, B, W, H, form, out, n);
}

// First sweep of a level: every cache entry is empty, so instead of a worklist the kernel walks the whole
// interior (and also does the mode 0 copy-through).
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_refine_first(StageArgs a) {
    __shared__ double2 s_exp[128]; // the specified exp's table (exp_tab_stage)
    exp_tab_stage(s_exp);
    __syncthreads();
    const int W = a.W, H = a.H;
    const DirArgs &d = a.d[blockIdx.z];
    const int x = d.own.XL + 1 + blockIdx.x * blockDim.x + threadIdx.x;
    const int y = d.own.YL + 1 + blockIdx.y;
    if (x > d.own.XR - 1 || y > d.own.YR - 1) return;
    const size_t pix = (size_t)y * W + x;
    const double *in = d.f64_a;
    const double dC = in[pix];
    if (dC == (double)NOMATCH) return;
    const double dE = in[pix + 1], dW = in[pix - 1], dN = in[pix - W], dS = in[pix + W];
    const int mode = (int)(dE != (double)NOMATCH && dW != (double)NOMATCH) +
                     (int)(dS != (double)NOMATCH && dN != (double)NOMATCH) * 2;
    if (mode == 0) {
        d.f64_b[pix] = dC;
        return;
    }
    const int key = (int)(dC - 1.5) + x;
    double pwp, delta, xs[3] = {0.0, 0.0, 0.0};
    RfLeft L;
    refine_left(d.img4_own, W, x, y, L);
#pragma unroll 1
    for (int c = 0; c < 3; c++) { // (rolled: one copy of the 27-element loop; the left differences stay in registers)
        xs[0] = xs[1];
        xs[1] = xs[2];
        xs[2] = refine_cost_left(L, d.img4_oth, W, H, y, key + c);
    }
    refine_entry(xs[0], xs[1], xs[2], pwp, delta);
    rf_store(d, a.rf_stride, pix, key - x, pwp, delta);
    const double val = refine_update(mode, dC, dE, dW, dN, dS, pwp, delta, a.ws, s_exp);
    d.f64_b[pix] = val;
    // The second cache way, filled ahead of its first use.  The level starts from integers, so int(dC - 1.5) sits in the
    // middle of its interval and the pixel leaves it on the side its first update points to: measured with the oracle
    // (tests/tools/analyze_keys.py), the next iMatch a pixel needs is the neighbour in that direction for > 99.9 % of the
    // pixels -- in the sweep kernels each of them would be a cache miss (C2's top level: 4.8 M in the third sweep alone).
    // The neighbour's data term shares two of its three matching costs with this one: one more right window.
    if (a.flag3 & 2) return; // option refine_prefill = 0 (A/B)
    const int rel0 = key - x, rel1 = (int)(val - 1.5);
    const int rel2 = rel1 != rel0 ? rel1 : (val > dC ? rel0 + 1 : (val < dC ? rel0 - 1 : rel0));
    if (rel2 == rel0 + 1 || rel2 == rel0 - 1) {
        const bool up = rel2 > rel0;
        const double xe = refine_cost_left(L, d.img4_oth, W, H, y, up ? key + 3 : key - 1);
        double p2, q2;
        refine_entry(up ? xs[1] : xe, up ? xs[2] : xs[0], up ? xe : xs[1], p2, q2);
        rf_store(d, a.rf_stride, pix, rel2, p2, q2);
    }
}

// One Jacobi sweep f64_a -> f64_b: RF_PPT vertically adjacent pixels per thread, every load issued up front.
// The data term (pwp, delta) of a pixel is cached, two entries per pixel indexed by the parity of iMatch - x: the
// iteration settles into flipping between two ADJACENT iMatch values, so both stay resident (C2's top level: 40 % of the
// pixels miss in sweep 2, 2 % in sweep 10, 0.03 % in sweep 40, a few hundred pixels per sweep at the end).  A pixel whose entry belongs to
// another iMatch is a miss.  Misses are served inside the workgroup: compacted through an LDS list and computed
// four lanes per entry (a lane each when the list is long); computing them in place would run the data-term
// routine in every second wave for one or two lanes.
// TOP only gives the top level's launches (the dominant kernel) their own name in rocprof traces.
#define RFW_CAP (64 * RF_PPT) // every pixel of a wave may miss
template <int TOP>
__global__ __launch_bounds__(256) void k_refine_sweep(StageArgs a) {
    // Misses are served per WAVE (own list, own results, no workgroup barrier): a wave that has none -- nearly all of
    // them once the iteration has settled -- never waits for a neighbour's ~3 us service chain.
    __shared__ uint32_t s_list[4][RFW_CAP]; // slot of the owner (lane * RF_PPT + i) | (iMatch - x) << 16
    __shared__ double s_res[4][RFW_CAP][2];
    __shared__ double2 s_exp[128]; // the specified exp's table (exp_tab_stage)
    exp_tab_stage(s_exp);
    __syncthreads(); // (the only workgroup barrier: before any wave leaves)
    const DirArgs &d = a.d[blockIdx.z];
    const int W = a.W, H = a.H;
    const int x = d.own.XL + 1 + blockIdx.x * 256 + (int)threadIdx.x;
    const int y0 = d.own.YL + 1 + blockIdx.y * RF_PPT; // the interior's rows
    const int ylast = d.own.YR - 1;
    if (y0 > ylast) return; // uniform
    const bool colok = x <= d.own.XR - 1;
    const int xs = colok ? x : d.own.XL + 1; // out-of-range lanes shadow a valid column (no stores)
    const double *__restrict__ in = d.f64_a;
    double *__restrict__ out = d.f64_b;
    double col[RF_PPT + 2], dE[RF_PPT], dW[RF_PPT];
    uint32_t kk[RF_PPT];
#pragma unroll
    for (int i = 0; i < RF_PPT + 2; i++) {
        const int yy = min(y0 - 1 + i, ylast + 1);
        col[i] = in[(size_t)yy * W + xs];
    }
#pragma unroll
    for (int i = 0; i < RF_PPT; i++) {
        const int yy = min(y0 + i, ylast);
        dE[i] = in[(size_t)yy * W + xs + 1];
        dW[i] = in[(size_t)yy * W + xs - 1];
        kk[i] = d.rf_key[(size_t)yy * W + xs]; // both ways' keys: their address does not depend on the state
    }
    { // a wave whose 64 x RF_PPT pixels are all NOMATCH (outside an elliptic mask, a hole) has nothing to update: the cache
      // addresses below depend on the state anyway, so leaving here costs no extra round trip and saves its entry loads
        bool any = false;
#pragma unroll
        for (int i = 0; i < RF_PPT; i++) any |= colok && (y0 + i <= ylast) && col[i + 1] != (double)NOMATCH;
        if (!__ballot(any)) return; // wave-uniform; no workgroup barrier follows
    }
    int rel[RF_PPT], crel[RF_PPT];
    double pwp[RF_PPT], delta[RF_PPT];
#pragma unroll
    for (int i = 0; i < RF_PPT; i++) {
        const int yy = min(y0 + i, ylast);
        rel[i] = (int)(col[i + 1] - 1.5); // .cpp:625 (iMatch - x)
        const double2 e = d.rf_ent[(size_t)yy * W + xs + (size_t)(rel[i] & 1) * a.rf_stride]; // the way the state selects
        crel[i] = (int)(int16_t)(kk[i] >> ((rel[i] & 1) << 4));
        pwp[i] = e.x;
        delta[i] = e.y;
    }
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int xw0 = d.own.XL + 1 + blockIdx.x * 256 + wid * 64; // column of this wave's lane 0
    unsigned live = 0, miss = 0;
    int mode[RF_PPT], pos[RF_PPT];
    int n = 0; // misses of this wave (uniform)
#pragma unroll
    for (int i = 0; i < RF_PPT; i++) {
        const double dC = col[i + 1], dN = col[i], dS = col[i + 2];
        const bool lv = colok && (y0 + i <= ylast) && dC != (double)NOMATCH; // .cpp:613
        mode[i] = (int)(dE[i] != (double)NOMATCH && dW[i] != (double)NOMATCH) +
                  (int)(dS != (double)NOMATCH && dN != (double)NOMATCH) * 2; // .cpp:620
        const bool ms = lv && mode[i] != 0 && crel[i] != rel[i];
        const unsigned long long mm = __ballot(ms);
        pos[i] = n + __popcll(mm & ((1ull << lane) - 1ull));
        if (ms) s_list[wid][pos[i]] = (uint32_t)(lane * RF_PPT + i) | ((uint32_t)(rel[i] & 0xffff) << 16);
        n += __popcll(mm);
        live |= (unsigned)lv << i;
        miss |= (unsigned)ms << i;
    }
    if (n) { // wave-uniform
        __builtin_amdgcn_wave_barrier();
        for (int done = 0; done < n;) { // uniform
            if (n - done > 16) { // a lane per entry: one round serves up to 64 with 1/3 of the quads' instructions
                const int e = done + lane;
                if (e < n) {
                    const uint32_t en = s_list[wid][e];
                    const int slot = en & 0xffff, r = (int)(int16_t)(en >> 16);
                    const int ex = xw0 + slot / RF_PPT, ey = y0 + slot % RF_PPT;
                    double p, q;
                    refine_data_term_packed(d.img4_own, d.img4_oth, W, H, ex, ey, r + ex, p, q);
                    s_res[wid][e][0] = p;
                    s_res[wid][e][1] = q;
                }
                done += 64;
            } else { // four lanes per entry: a third of the latency, which is what a handful of misses costs
                const int e = done + (lane >> 2);
                const bool ok = e < n;
                const uint32_t en = s_list[wid][ok ? e : done];
                const int slot = en & 0xffff, r = (int)(int16_t)(en >> 16);
                const int ex = xw0 + slot / RF_PPT, ey = y0 + slot % RF_PPT;
                double p, q;
                refine_data_term_quad(d.img4_own, d.img4_oth, W, H, ex, ey, r + ex, lane & 3, p, q);
                if (ok && (lane & 3) == 0) {
                    s_res[wid][e][0] = p;
                    s_res[wid][e][1] = q;
                }
                done += 16;
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
#pragma unroll
    for (int i = 0; i < RF_PPT; i++) {
        if (!((live >> i) & 1u)) continue;
        const size_t pix = (size_t)(y0 + i) * W + x;
        if ((miss >> i) & 1u) {
            pwp[i] = s_res[wid][pos[i]][0];
            delta[i] = s_res[wid][pos[i]][1];
            rf_store(d, a.rf_stride, pix, rel[i], pwp[i], delta[i]);
        }
        out[pix] = (mode[i] == 0) ? col[i + 1] /* .cpp:655 */
                                  : refine_update(mode[i], col[i + 1], dE[i], dW[i], col[i], col[i + 2], pwp[i], delta[i], a.ws, s_exp);
    }
}

// scatters the update list of the k_refine_skew launch a.flag3 into the cache and clears the other counter set
__global__ __launch_bounds__(256) void k_refine_apply(StageArgs a) {
    const int32_t *cnt = a.upd_cnt + (a.flag3 & 1) * RF_UPD_SHARDS;
    if (blockIdx.x == 0 && threadIdx.x < RF_UPD_SHARDS) a.upd_cnt[((a.flag3 + 1) & 1) * RF_UPD_SHARDS + threadIdx.x] = 0;
    for (int sh = blockIdx.y; sh < RF_UPD_SHARDS; sh += gridDim.y) {
        const int n = min(cnt[sh], a.upd_cap);
        for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
            const RfUpd u = a.upd_list[(size_t)sh * a.upd_cap + i];
            const DirArgs &d = a.d[u.pix >> 31];
            rf_store(d, a.rf_stride, (size_t)(u.pix & 0x7fffffffu), u.rel, u.pwp, u.delta);
        }
    }
}

// scatters a time-skewed launch's update list into the cache
void launch_refine_apply(const StageArgs &a, hipStream_t st) { hipLaunchKernelGGL(k_refine_apply, dim3(8, RF_UPD_SHARDS), dim3(256), 0, st, a); }

// One sweep f64_a -> f64_b over the interior (a.flag2 = sweep index: 0 = k_refine_first; a.flag = top level); ev0 / ev1 (optional)
// bracket the launch.
void launch_refine_sweep(const StageArgs &a, hipStream_t st, hipEvent_t ev0, hipEvent_t ev1) {
    int rows = 0, cols = 0;
    for (int v = 0; v < a.ndir; v++) {
        rows = max(rows, a.d[v].own.YR - a.d[v].own.YL - 1);
        cols = max(cols, a.d[v].own.XR - a.d[v].own.XL - 1);
    }
    if (rows <= 0 || cols <= 0) return;
    const dim3 grid((cols + 255) / 256, rows, a.ndir);
    if (ev0) (void)hipEventRecord(ev0, st);
    if (a.flag2 == 0) { // first sweep: every pixel needs its data term
        hipLaunchKernelGGL(k_refine_first, grid, dim3(256), 0, st, a);
    } else {
        const dim3 sgrid(grid.x, (grid.y + RF_PPT - 1) / RF_PPT, grid.z);
        if (a.flag) hipLaunchKernelGGL(k_refine_sweep<1>, sgrid, dim3(256), 0, st, a);
        else hipLaunchKernelGGL(k_refine_sweep<0>, sgrid, dim3(256), 0, st, a);
    }
    if (ev1) (void)hipEventRecord(ev1, st);
}
