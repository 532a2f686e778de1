// k_match.hip -- the NCC patch-cost kernels.
//   next_valid + hl_interval  replace the carried boundary_L / boundary_R bookkeeping of
//                             CStereoMatching::HighLevelInitialMatch, .cpp:259-288
//   ncc_argmax                replaces the candidate loops of LowestLevelInitialMatch (.cpp:197-223),
//                             HighLevelInitialMatch (.cpp:289-302) and Rematch (.cpp:531-565)
//
// NCC arithmetic: the reference recomputes mean / norm / dot in fp64 per candidate
// (CManageData.cpp:81-90).  Here every window statistic is an exact integer:
//   score = (n*Sab - Sa*Sb) / sqrt((n*Saa - Sa^2) * (n*Sbb - Sb^2)),   zero variance -> 0
// with Sa,Saa,Sb,Sbb from the per-level tables (k_pyramid.hip) and Sab accumulated in int32
// from LDS-staged rows of both views; one fp64 divide+sqrt per candidate; strict '>' from -1
// in ascending column order, exactly as the reference.
#include "rsm_dev.h"

// ---------------------------------------------------------------- next valid parent column
// nv[row][t] = smallest i > t with parent[row][i] != NOMATCH, or INT_MAX. One wave per parent row.
__global__ void k_next_valid(const double *__restrict__ parent, int Wp, int Hp, int32_t *__restrict__ nv) {
    const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= Hp) return;
    const double *p = parent + (size_t)row * Wp;
    int32_t *o = nv + (size_t)row * Wp;
    int carry = 0x7fffffff;
    const int nchunks = (Wp + 63) >> 6;
    for (int c = nchunks - 1; c >= 0; c--) {
        const int t = (c << 6) + lane;
        const bool valid = (t < Wp) && (p[t] != (double)NOMATCH);
        const unsigned long long m = __ballot(valid);
        const unsigned long long above = (lane == 63) ? 0ull : (m & ~((2ull << lane) - 1ull));
        const int res = above ? ((c << 6) + __builtin_ctzll(above)) : carry;
        if (t < Wp) o[t] = res;
        if (m) carry = (c << 6) + __builtin_ctzll(m);
    }
}

void launch_next_valid(const double *parent, int Wp, int Hp, int32_t *nv, hipStream_t st) {
    hipLaunchKernelGGL(k_next_valid, dim3((Hp + 3) / 4), dim3(256), 0, st, parent, Wp, Hp, nv);
}

// ---------------------------------------------------------------- HighLevel candidate intervals
// One wave per row. State (boundary_L, boundary_R) is carried across masked pixels exactly as the
// sequential reference loop: a pixel with a valid parent refreshes both; a pixel with a NOMATCH
// parent keeps L and refreshes R only if a valid parent exists to its right (<= XR>>1).
__global__ void k_hl_interval(StageArgs a) {
    const DirArgs &d = a.d[blockIdx.z];
    const int lane = threadIdx.x & 63;
    const int y = d.own.YL + blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (y > d.own.YR) return;
    const int W = a.W, Wp = a.Wp, off = a.offset;
    const int XL = d.own.XL, XR = d.own.XR, XL1 = d.oth.XL, XR1 = d.oth.XR;
    const int pr = (y + 1) >> 1; // int((y+1)/2.0), .cpp:259
    const double *s = d.parent + (size_t)pr * Wp;
    const int32_t *nv = d.parent_nv + (size_t)pr * Wp;
    const uint8_t *mk = d.mask_own + (size_t)y * W;
    int carryL = XL1, carryR = XR1; // .cpp:260-261
    for (int x0 = XL; x0 <= XR; x0 += 64) {
        const int x = x0 + lane;
        const bool masked = (x <= XR) && (mk[x] == 255);
        bool freshL = false, freshR = false;
        int Lv = 0, Rv = 0;
        if (masked) {
            const int t = (x + 1) >> 1; // .cpp:267
            const double sv = s[t];
            if (sv != (double)NOMATCH) { // .cpp:286-287
                const int c = x + (int)(sv * 2 + 0.5);
                Lv = max(c - off, XL1);
                Rv = min(c + off, XR1);
                freshL = freshR = true;
            } else { // .cpp:275-282
                const int i = nv[t];
                if (i <= (XR >> 1)) {
                    Rv = min(i + (int)(s[i] * 2) + off + 1, XR1);
                    freshR = true;
                }
            }
        }
        const unsigned long long le = (lane == 63) ? ~0ull : ((2ull << lane) - 1ull);
        const unsigned long long mL = __ballot(freshL), mR = __ballot(freshR);
        const unsigned long long bl = mL & le, br = mR & le;
        const int srcL = bl ? (63 - __builtin_clzll(bl)) : 0;
        const int srcR = br ? (63 - __builtin_clzll(br)) : 0;
        const int gl = __shfl(Lv, srcL), gr = __shfl(Rv, srcR);
        const int L = bl ? gl : carryL;
        const int R = br ? gr : carryR;
        if (masked) {
            d.BL[(size_t)y * W + x] = (int16_t)L;
            d.BR[(size_t)y * W + x] = (int16_t)R;
        }
        if (mL) carryL = __shfl(Lv, 63 - __builtin_clzll(mL));
        if (mR) carryR = __shfl(Rv, 63 - __builtin_clzll(mR));
    }
}

void launch_hl_interval(const StageArgs &a, hipStream_t st) {
    int rows = 0;
    for (int v = 0; v < a.ndir; v++) rows = max(rows, a.d[v].own.YR - a.d[v].own.YL + 1);
    if (rows <= 0) return;
    hipLaunchKernelGGL(k_hl_interval, dim3((rows + 3) / 4, 1, a.ndir), dim3(256), 0, st, a);
}

// ---------------------------------------------------------------- NCC interval argmax
#define NCC_TX 256  // pixels of one row per block
#define NCC_CH 512  // candidate columns staged per pass

__global__ __launch_bounds__(NCC_TX) void k_ncc_argmax(StageArgs a, int mode, int strideA, int strideB) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    int *s_lohi = (int *)smem;  // [0]=min L, [1]=max R over the block
    uint8_t *sA = smem + 16;
    const DirArgs &d = a.d[blockIdx.z];
    const int W = a.W, r = a.r, ws = 2 * r + 1;
    const int n = ws * ws * 3;
    const int y = d.own.YL + blockIdx.y;
    const int x0 = d.own.XL + blockIdx.x * NCC_TX;
    if (y > d.own.YR || x0 > d.own.XR) return; // uniform per block
    uint8_t *sB = sA + (size_t)ws * strideA;
    const int x = x0 + threadIdx.x;
    const size_t pix = (size_t)y * W + x;

    bool active = (x <= d.own.XR) && (d.mask_own[pix] == 255);
    if (mode == 2 && active) active = (d.d16_in[pix] == NOMATCH); // Rematch: .cpp:538
    int L = 0x7fffffff, R = -1;
    if (active) {
        if (mode == 0) {
            L = d.oth.XL; // .cpp:207
            R = d.oth.XR;
        } else {
            L = d.BL[pix];
            R = d.BR[pix];
        }
        L = max(L, r); // windows that would leave the image are skipped (UB in the reference)
        R = min(R, W - 1 - r);
        if (L > R) active = false;
    }
    if (threadIdx.x == 0) {
        s_lohi[0] = 0x7fffffff;
        s_lohi[1] = -1;
    }
    __syncthreads();
    {
        int lo = active ? L : 0x7fffffff, hi = active ? R : -1;
        for (int o = 32; o > 0; o >>= 1) {
            lo = min(lo, __shfl_xor(lo, o));
            hi = max(hi, __shfl_xor(hi, o));
        }
        if ((threadIdx.x & 63) == 0) {
            atomicMin(&s_lohi[0], lo);
            atomicMax(&s_lohi[1], hi);
        }
    }
    __syncthreads();
    const int cmin = s_lohi[0], cmax = s_lohi[1];
    if (cmax < cmin) return; // nothing to do in this block

    // stage the own-view rows y-r..y+r, columns x0-r .. x0+NCC_TX-1+r
    {
        const int rowBytes = W * 3;
        const int b0 = (x0 - r) * 3;
        const int nb = (NCC_TX + 2 * r) * 3;
        for (int j = 0; j < ws; j++) {
            const uint8_t *src = d.img_own + (size_t)(y - r + j) * rowBytes;
            for (int i = threadIdx.x; i < nb; i += NCC_TX) {
                const int b = b0 + i;
                sA[j * strideA + i] = (b >= 0 && b < rowBytes) ? src[b] : 0;
            }
        }
    }
    int Sa = 0, Saa = 0;
    long long va = 0;
    if (active) {
        Sa = d.S1_own[pix];
        Saa = d.S2_own[pix];
        va = (long long)n * Saa - (long long)Sa * Sa;
    }
    int best = -1;
    double bestv = -1.0;
    const uint8_t *mq = d.mask_oth + (size_t)y * W;
    const int aoff = threadIdx.x * 3;

    for (int lo = cmin; lo <= cmax; lo += NCC_CH) {
        const int hi = min(lo + NCC_CH - 1, cmax);
        __syncthreads(); // previous pass done with sB (and sA staged on first pass)
        {
            const int rowBytes = W * 3;
            const int b0 = (lo - r) * 3;
            const int nb = (hi - lo + 1 + 2 * r) * 3;
            for (int j = 0; j < ws; j++) {
                const uint8_t *src = d.img_oth + (size_t)(y - r + j) * rowBytes;
                for (int i = threadIdx.x; i < nb; i += NCC_TX) {
                    const int b = b0 + i;
                    sB[j * strideB + i] = (b >= 0 && b < rowBytes) ? src[b] : 0;
                }
            }
        }
        __syncthreads();
        if (active) {
            const int c0 = max(L, lo), c1 = min(R, hi);
            for (int c = c0; c <= c1; c++) {
                if (mq[c] != 255) continue; // .cpp:209
                const int boff = (c - lo) * 3;
                int Sab = 0;
                for (int j = 0; j < ws; j++) {
                    const uint8_t *pa = sA + j * strideA + aoff;
                    const uint8_t *pb = sB + j * strideB + boff;
                    for (int i = 0; i < ws * 3; i++) Sab += (int)pa[i] * (int)pb[i];
                }
                const int Sb = d.S1_oth[(size_t)y * W + c];
                const int Sbb = d.S2_oth[(size_t)y * W + c];
                const long long vb = (long long)n * Sbb - (long long)Sb * Sb;
                const long long num = (long long)n * Sab - (long long)Sa * Sb;
                double score = 0.0;
                if (va > 0 && vb > 0) score = (double)num / sqrt((double)va * (double)vb);
                if (score > bestv) { // .cpp:213
                    bestv = score;
                    best = c;
                }
            }
        }
    }
    if (active && best != -1) d.d16_out[pix] = (int16_t)(best - x); // .cpp:219-222 / 301-302 / 563-564
}

void launch_ncc_argmax(const StageArgs &a, int mode, hipStream_t st) {
    int rows = 0, cols = 0;
    for (int v = 0; v < a.ndir; v++) {
        rows = max(rows, a.d[v].own.YR - a.d[v].own.YL + 1);
        cols = max(cols, a.d[v].own.XR - a.d[v].own.XL + 1);
    }
    if (rows <= 0 || cols <= 0) return;
    const int ws = 2 * a.r + 1;
    const int strideA = (((NCC_TX + 2 * a.r) * 3 + 3) & ~3) + 4;
    const int strideB = (((NCC_CH + 2 * a.r) * 3 + 3) & ~3) + 4;
    const size_t lds = 16 + (size_t)ws * strideA + (size_t)ws * strideB;
    dim3 grid((cols + NCC_TX - 1) / NCC_TX, rows, a.ndir);
    hipLaunchKernelGGL(k_ncc_argmax, grid, dim3(NCC_TX), lds, st, a, mode, strideA, strideB);
}
