// k_constraints.hip -- the integer constraint / filter stages of MatchOneLayer.
//   smooth        CStereoMatching::SmoothConstraint      .cpp:370-448  (gather form of the scatter)
//   order         CStereoMatching::OrderConstraint       .cpp:310-368  (one workgroup per row)
//   uniq<T>       CStereoMatching::UniquenessContraint_  .cpp:463-497  (one wave per row, carry chain)
//   set_boundary  CStereoMatching::SetBoundary_smooth    .cpp:817-942  (column sweeps, then row sweeps)
//   median        CStereoMatching::MedianFilter          .cpp:763-815
#include "rsm_dev.h"

__device__ __forceinline__ bool in_margin(const Mg &m, int x, int y) {
    return x >= m.XL && x <= m.XR && y >= m.YL && y <= m.YR;
}

// ---------------------------------------------------------------- SmoothConstraint
// The reference scatters neighbour counts into a CV_8UC2 map serially (.cpp:380-432) and then kills
// pixels with total==0 || 2*differ>total (.cpp:443).  Gather form per target pixel (X,Y); "src in M"
// means the scattering pixel lies inside the own margin.  The SE total counts land at BYTE index x
// of row y and x+2 of row y+1 (.cpp:423-424), i.e. at pixel x/2 (total if x even, differ if odd):
// the four "slip" terms below.
__global__ void k_smooth(StageArgs a) {
    const DirArgs &d = a.d[blockIdx.z];
    const int W = a.W, H = a.H;
    const int X = blockIdx.x * blockDim.x + threadIdx.x;
    const int Y = blockIdx.y;
    if (X >= W || Y >= H) return;
    const int16_t *D = d.d16_in;
    const size_t pix = (size_t)Y * W + X;
    const int16_t dc = D[pix];
    const Mg M = d.own;
    if (!in_margin(M, X, Y)) {
        d.d16_out[pix] = dc;
        return;
    }
    auto val = [&](int x, int y) -> int {
        return (x >= 0 && x < W && y >= 0 && y < H) ? (int)D[(size_t)y * W + x] : NOMATCH;
    };
    auto differ = [](int p, int q) -> int { return abs(p - q) > 1; };
    int T = 0, F = 0;
    const bool vc = dc != NOMATCH;
    if (vc) {
        // (X,Y) as source (it is inside M): E, SW, S totals; E, SW, S, SE differs
        const int e = val(X + 1, Y), sw = val(X - 1, Y + 1), s = val(X, Y + 1), se = val(X + 1, Y + 1);
        if (e != NOMATCH) { T++; F += differ(dc, e); }
        if (sw != NOMATCH) { T++; F += differ(dc, sw); }
        if (s != NOMATCH) { T++; F += differ(dc, s); }
        if (se != NOMATCH) { F += differ(dc, se); }
        // (X,Y) as target of W (its E), NE (its SW), N (its S), NW (its SE: differ only)
        const int w = val(X - 1, Y), ne = val(X + 1, Y - 1), nn = val(X, Y - 1), nw = val(X - 1, Y - 1);
        if (w != NOMATCH && in_margin(M, X - 1, Y)) { T++; F += differ(w, dc); }
        if (ne != NOMATCH && in_margin(M, X + 1, Y - 1)) { T++; F += differ(ne, dc); }
        if (nn != NOMATCH && in_margin(M, X, Y - 1)) { T++; F += differ(nn, dc); }
        if (nw != NOMATCH && in_margin(M, X - 1, Y - 1)) { F += differ(nw, dc); }
    }
    // slip terms (independent of the validity of (X,Y))
    if (in_margin(M, 2 * X, Y) && val(2 * X, Y) != NOMATCH && val(2 * X + 1, Y + 1) != NOMATCH) T++;
    if (in_margin(M, 2 * X - 2, Y - 1) && val(2 * X - 2, Y - 1) != NOMATCH && val(2 * X - 1, Y) != NOMATCH) T++;
    if (in_margin(M, 2 * X + 1, Y) && val(2 * X + 1, Y) != NOMATCH && val(2 * X + 2, Y + 1) != NOMATCH) F++;
    if (in_margin(M, 2 * X - 1, Y - 1) && val(2 * X - 1, Y - 1) != NOMATCH && val(2 * X, Y) != NOMATCH) F++;
    T &= 255; // the reference counters are uchar
    F &= 255;
    d.d16_out[pix] = (T == 0 || (F << 1) > T) ? (int16_t)NOMATCH : dc;
}

void launch_smooth(const StageArgs &a, hipStream_t st) {
    dim3 grid((a.W + 255) / 256, a.H, a.ndir);
    hipLaunchKernelGGL(k_smooth, grid, dim3(256), 0, st, a);
}

// ---------------------------------------------------------------- OrderConstraint
// One 256-thread workgroup per row. Valid pixels are compacted in ascending x into LDS
// (m = p[x] + x). Crossing count of i = #{j<i : m_j > m_i} + #{j>i : m_j < m_i} -- the row sums of the
// reference's symmetric matrix A (.cpp:337-353) without ever materialising it. Then the greedy loop of
// .cpp:354-364: remove the first pixel with the largest count until no crossings remain.
__global__ __launch_bounds__(256) void k_order(StageArgs a, int maxL) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const DirArgs &d = a.d[blockIdx.z];
    const int y = d.own.YL + blockIdx.x;
    if (y > d.own.YR) return;
    const int W = a.W, XL = d.own.XL, XR = d.own.XR;
    int *cnt = (int *)smem;                  // [maxL]
    int16_t *line = (int16_t *)(cnt + maxL); // [maxL]
    int16_t *idx = line + maxL;              // [maxL]
    __shared__ int s_n, s_flag, s_best[4], s_bidx[4], s_wsum[4];
    int16_t *p = d.d16_in + (size_t)y * W;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    if (tid == 0) {
        s_n = 0;
        s_flag = 0;
    }
    __syncthreads();
    // ordered compaction, 256 columns per pass
    for (int x0 = XL; x0 <= XR; x0 += 256) {
        const int x = x0 + tid;
        const int v = (x <= XR) ? (int)p[x] : NOMATCH;
        const bool valid = v != NOMATCH;
        const unsigned long long m = __ballot(valid);
        const int before = __popcll(m & ((1ull << lane) - 1ull));
        if (lane == 0) s_wsum[wid] = __popcll(m);
        __syncthreads();
        int base = s_n;
        for (int w = 0; w < wid; w++) base += s_wsum[w];
        if (valid) {
            line[base + before] = (int16_t)(v + x);
            idx[base + before] = (int16_t)x;
        }
        __syncthreads();
        if (tid == 0) s_n += s_wsum[0] + s_wsum[1] + s_wsum[2] + s_wsum[3];
        __syncthreads();
    }
    const int n = s_n;
    if (n < 2) return;
    // fast path: already non-decreasing -> no crossings
    for (int i = tid; i + 1 < n; i += 256)
        if (line[i] > line[i + 1]) s_flag = 1;
    __syncthreads();
    if (!s_flag) return;
    for (int i = tid; i < n; i += 256) {
        const int mi = line[i];
        int c = 0;
        for (int j = 0; j < i; j++) c += line[j] > mi;
        for (int j = i + 1; j < n; j++) c += line[j] < mi;
        cnt[i] = c;
    }
    __syncthreads();
    for (;;) {
        // first index of the maximum count (op_max: strict '>' scan -> first maximum)
        int bv = -1, bi = 0x7fffffff;
        for (int i = tid; i < n; i += 256) {
            const int c = cnt[i];
            if (c > bv) {
                bv = c;
                bi = i;
            }
        }
        for (int o = 32; o > 0; o >>= 1) {
            const int ov = __shfl_xor(bv, o), oi = __shfl_xor(bi, o);
            if (ov > bv || (ov == bv && oi < bi)) {
                bv = ov;
                bi = oi;
            }
        }
        if (lane == 0) {
            s_best[wid] = bv;
            s_bidx[wid] = bi;
        }
        __syncthreads();
        bv = s_best[0];
        bi = s_bidx[0];
        for (int w = 1; w < 4; w++)
            if (s_best[w] > bv || (s_best[w] == bv && s_bidx[w] < bi)) {
                bv = s_best[w];
                bi = s_bidx[w];
            }
        if (bv <= 0) break; // ones_count == 0 (.cpp:354)
        const int mb = line[bi];
        __syncthreads(); // everyone has read s_best / line[bi]
        for (int j = tid; j < n; j += 256) {
            const int c = cnt[j];
            if (c < 0 || j == bi) continue; // removed pixels have no edges left
            const int mj = line[j];
            if ((j < bi && mj > mb) || (j > bi && mj < mb)) cnt[j] = c - 1;
        }
        __syncthreads();
        if (tid == 0) {
            cnt[bi] = -1; // dead: row/col of A zeroed (.cpp:359-361)
            p[idx[bi]] = (int16_t)NOMATCH;
        }
        __syncthreads();
    }
}

void launch_order(const StageArgs &a, hipStream_t st) {
    int rows = 0, maxL = 0;
    for (int v = 0; v < a.ndir; v++) {
        rows = max(rows, a.d[v].own.YR - a.d[v].own.YL + 1);
        maxL = max(maxL, a.d[v].own.XR - a.d[v].own.XL + 1);
    }
    if (rows <= 0 || maxL <= 0) return;
    maxL = (maxL + 7) & ~7;
    const size_t lds = (size_t)maxL * (4 + 2 + 2);
    hipLaunchKernelGGL(k_order, dim3(rows, 1, a.ndir), dim3(256), lds, st, a, maxL);
}

// ---------------------------------------------------------------- UniquenessContraint_<T>
// Sequential dependency of the reference: the rescue test at .cpp:492 reads p[x-1] AFTER it may have
// been killed.  With  g(x) = valid && (direct || rescue by the untouched p[x+1])  and
// h(x) = valid && !direct && |q[bL+1] + p_orig[x-1]| < 2,  alive(x) = g(x) || (h(x) && alive(x-1)):
// a carry chain, resolved per 64-pixel chunk with ballots and a Kogge-Stone prefix on the masks.
template <typename T>
__device__ __forceinline__ bool close2(T q, T p);
template <>
__device__ __forceinline__ bool close2<int>(int q, int p) { return abs(q + p) < 2; }
template <>
__device__ __forceinline__ bool close2<double>(double q, double p) { return fabs(q + p) < 2; }

template <typename T, typename CT> // T storage (int16_t/double), CT compute (int/double)
__global__ void k_uniq(T *__restrict__ P, const T *__restrict__ Q, int W, int H, Mg own, Mg oth) {
    const int lane = threadIdx.x & 63;
    const int y = own.YL + blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (y > own.YR) return;
    const int XL = own.XL, XR = own.XR, XL1 = oth.XL, XR1 = oth.XR;
    const long long total = (long long)W * H;
    T *p = P + (size_t)y * W;
    const T *q = Q + (size_t)y * W;
    unsigned long long cin = 1ull; // p[XL-1] is never modified: "alive" as far as the chain goes
    CT prev_last = (XL - 1 >= 0) ? (CT)p[XL - 1] : (CT)NOMATCH;
    for (int x0 = XL; x0 <= XR; x0 += 64) {
        const int x = x0 + lane;
        const CT raw = (x <= XR + 1 && x < W) ? (CT)p[x] : (CT)NOMATCH;
        const bool valid = (x <= XR) && (raw != (CT)NOMATCH);
        CT pm1 = (CT)__shfl_up(raw, 1);
        if (lane == 0) pm1 = prev_last;
        CT pp1 = (CT)__shfl_down(raw, 1);
        if (lane == 63) pp1 = (x + 1 <= XR + 1 && x + 1 < W) ? (CT)p[x + 1] : (CT)NOMATCH;
        bool g = false, h = false;
        if (valid) {
            const int bL = max((int)(raw + 0.5) + x - 1, XL1); // .cpp:482 (C truncation)
            const int bR = min(bL + 2, XR1);
            bool direct = false;
            for (int i = bL; i <= bR; i++) direct = direct || close2<CT>((CT)q[i], raw);
            if (direct) g = true;
            else {
                const long long fi = (long long)y * W + bL + 1; // flat-buffer emulation of q[bL+1]
                const CT qv = (fi >= 0 && fi < total) ? (CT)Q[fi] : (CT)NOMATCH;
                g = close2<CT>(qv, pp1);
                h = close2<CT>(qv, pm1);
            }
        }
        unsigned long long G = __ballot(g), Pm = __ballot(h);
        G |= Pm & cin; // carry into lane 0
        for (int s = 1; s < 64; s <<= 1) {
            G |= Pm & (G << s);
            Pm &= (Pm << s);
        }
        const bool alive = (G >> lane) & 1ull;
        if (valid && !alive) p[x] = (T)NOMATCH;
        cin = (G >> 63) & 1ull;
        prev_last = (CT)__shfl(raw, 63);
    }
}

void launch_uniq_s16(int16_t *p, const int16_t *q, int W, int H, Mg own, Mg oth, hipStream_t st) {
    const int rows = own.YR - own.YL + 1;
    if (rows <= 0 || own.XR < own.XL) return;
    hipLaunchKernelGGL((k_uniq<int16_t, int>), dim3((rows + 3) / 4), dim3(256), 0, st, p, q, W, H, own, oth);
}
void launch_uniq_f64(double *p, const double *q, int W, int H, Mg own, Mg oth, hipStream_t st) {
    const int rows = own.YR - own.YL + 1;
    if (rows <= 0 || own.XR < own.XL) return;
    hipLaunchKernelGGL((k_uniq<double, double>), dim3((rows + 3) / 4), dim3(256), 0, st, p, q, W, H, own, oth);
}

// ---------------------------------------------------------------- SetBoundary_smooth<short>
#define MAX_DISPARITY 2 // .cpp:4
// vertical sweeps (.cpp:842-901): columns are independent -> one thread per column.
__global__ void k_setb_vert(StageArgs a) {
    const DirArgs &d = a.d[blockIdx.z];
    const int x = d.own.XL + blockIdx.x * blockDim.x + threadIdx.x;
    if (x > d.own.XR) return;
    const int W = a.W, YL = d.own.YL, YR = d.own.YR;
    const uint8_t *mk = d.mask_own;
    const int16_t *src = d.d16_in;
    int16_t *BL = d.BL, *BR = d.BR;
    int bl = -10000, br = 10000; // .cpp:832-833
    for (int y = YL; y <= YR - 1; y++) {
        const size_t o = (size_t)y * W + x;
        int nbl = -10000, nbr = 10000;
        if (mk[o] == 255) {
            const int ref = src[o];
            if (ref == NOMATCH) {
                nbl = max(bl - MAX_DISPARITY, nbl);
                nbr = min(br + MAX_DISPARITY, nbr);
            } else {
                bl = ref;
                br = ref;
                nbl = max(ref - MAX_DISPARITY, nbl);
                nbr = min(ref + MAX_DISPARITY, nbr);
            }
        }
        BL[o] = (int16_t)bl;
        BR[o] = (int16_t)br;
        bl = nbl;
        br = nbr;
    }
    BL[(size_t)YR * W + x] = (int16_t)bl;
    BR[(size_t)YR * W + x] = (int16_t)br;
    for (int y = YR; y >= YL + 1; y--) {
        const size_t o = (size_t)y * W + x;
        int ubl = BL[o - W], ubr = BR[o - W];
        if (mk[o] == 255) {
            const int ref = src[o];
            if (ref == NOMATCH) {
                ubl = max(bl - MAX_DISPARITY, ubl);
                ubr = min(br + MAX_DISPARITY, ubr);
            } else {
                bl = ref;
                br = ref;
                ubl = max(ref - MAX_DISPARITY, ubl);
                ubr = min(ref + MAX_DISPARITY, ubr);
            }
        }
        BL[o] = (int16_t)bl;
        BR[o] = (int16_t)br;
        bl = ubl;
        br = ubr;
    }
    BL[(size_t)YL * W + x] = (int16_t)bl;
    BR[(size_t)YL * W + x] = (int16_t)br;
}

// horizontal sweeps (.cpp:903-941): one thread per row, literal.
__global__ void k_setb_horiz(StageArgs a) {
    const DirArgs &d = a.d[blockIdx.z];
    const int y = d.own.YL + blockIdx.x * blockDim.x + threadIdx.x;
    if (y > d.own.YR) return;
    const int W = a.W, XL = d.own.XL, XR = d.own.XR, XL1 = d.oth.XL, XR1 = d.oth.XR;
    const uint8_t *mk = d.mask_own + (size_t)y * W;
    int16_t *bl = d.BL + (size_t)y * W, *br = d.BR + (size_t)y * W;
    int cbl = bl[XL], cbr = br[XL];
    for (int x = XL; x <= XR - 1; x++) {
        int nbl = bl[x + 1], nbr = br[x + 1];
        if (mk[x] == 255) {
            nbl = max(cbl - 1, nbl);
            nbr = min(cbr + MAX_DISPARITY, nbr);
            bl[x + 1] = (int16_t)nbl;
            br[x + 1] = (int16_t)nbr;
        }
        cbl = nbl;
        cbr = nbr;
    }
    // cbl/cbr now hold the values at XR
    for (int x = XR; x >= XL + 1; x--) {
        int lbl = bl[x - 1], lbr = br[x - 1];
        if (mk[x] == 255) {
            int A = (int16_t)(cbl + x), B = (int16_t)(cbr + x);
            if (A < XL1) A = XL1;
            if (B > XR1) B = XR1;
            bl[x] = (int16_t)A;
            br[x] = (int16_t)B;
            lbl = max(A - x - MAX_DISPARITY, lbl);
            lbr = min(B - x + 1, lbr);
            bl[x - 1] = (int16_t)lbl;
            br[x - 1] = (int16_t)lbr;
        }
        cbl = lbl;
        cbr = lbr;
    }
    if (mk[XL] == 255) {
        int A = (int16_t)(cbl + XL), B = (int16_t)(cbr + XL);
        if (A < XL1) A = XL1;
        if (B > XR1) A = XR1; // .cpp:938-939: the reference assigns bl here (typo kept)
        bl[XL] = (int16_t)A;
        br[XL] = (int16_t)B;
    }
}

void launch_set_boundary(const StageArgs &a, hipStream_t st) {
    int rows = 0, cols = 0;
    for (int v = 0; v < a.ndir; v++) {
        rows = max(rows, a.d[v].own.YR - a.d[v].own.YL + 1);
        cols = max(cols, a.d[v].own.XR - a.d[v].own.XL + 1);
    }
    if (rows <= 0 || cols <= 0) return;
    hipLaunchKernelGGL(k_setb_vert, dim3((cols + 63) / 64, 1, a.ndir), dim3(64), 0, st, a);
    hipLaunchKernelGGL(k_setb_horiz, dim3((rows + 63) / 64, 1, a.ndir), dim3(64), 0, st, a);
}

// ---------------------------------------------------------------- MedianFilter (1 iteration)
__global__ void k_median(StageArgs a) {
    const DirArgs &d = a.d[blockIdx.z];
    const int x = d.own.XL + blockIdx.x * blockDim.x + threadIdx.x;
    const int y = d.own.YL + blockIdx.y;
    if (x > d.own.XR || y > d.own.YR) return;
    const int W = a.W;
    if (d.mask_own[(size_t)y * W + x] != 255) return; // output stays NOMATCH (.cpp:772,788)
    const int16_t *D = d.d16_in;
    int u[6], k = 0;
    for (int i = x - 1; i < x + 1; i++) // .cpp:792: columns x-1 and x only
        for (int j = -1; j <= 1; j++) {
            const int v = D[(size_t)(y + j) * W + i];
            if (v != NOMATCH) u[k++] = v;
        }
    const int c = D[(size_t)y * W + x];
    int out = NOMATCH;
    const bool take = (c == NOMATCH) ? (k >= 4) : (k > 2);
    if (take) {
        for (int i = 1; i < k; i++) { // insertion sort, k <= 6
            const int v = u[i];
            int j = i - 1;
            while (j >= 0 && u[j] > v) {
                u[j + 1] = u[j];
                j--;
            }
            u[j + 1] = v;
        }
        const int half = k / 2;
        // arma::median: odd -> middle; even -> lo + (hi - lo)/2 (op_mean::robust_mean, integer division)
        out = (k & 1) ? u[half] : (u[half - 1] + (u[half] - u[half - 1]) / 2);
    }
    d.d16_out[(size_t)y * W + x] = (int16_t)out;
}

void launch_median(const StageArgs &a, hipStream_t st) {
    int rows = 0, cols = 0;
    for (int v = 0; v < a.ndir; v++) {
        rows = max(rows, a.d[v].own.YR - a.d[v].own.YL + 1);
        cols = max(cols, a.d[v].own.XR - a.d[v].own.XL + 1);
    }
    if (rows <= 0 || cols <= 0) return;
    hipLaunchKernelGGL(k_median, dim3((cols + 255) / 256, rows, a.ndir), dim3(256), 0, st, a);
}
