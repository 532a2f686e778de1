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
// in ascending column order.  That integer form is the FILTER: its value is within a few ulp of the true
// score, the reference's within ~n ulp of it (fixed-order fp64 sums), so whenever the winner is not ahead of
// every other candidate (and of the initial -1) by NCC_TIE_TOL the reference's last-bit rounding may pick
// another column.  Such pixels are appended to a tie list and re-evaluated by k_ncc_exact, a
// reference-order fp64 restatement of WindowToVec + arma::dot (CManageData.cpp:81-90, op_dot_meat.hpp:20-55),
// which overwrites their result: the chosen column is the reference's on few-level, saturated and periodic
// textures too (tests/test_gpu_ncc_ties.py).
#include "rsm_dev.h"

#include <stdlib.h>

// ---------------------------------------------------------------- next valid parent column
// nv[row][t] = smallest i > t with parent[row][i] != NOMATCH, or INT_MAX. One wave per parent row.
__global__ void k_next_valid(const double *__restrict__ parent0, const double *__restrict__ parent1, int Wp, int Hp,
                             int32_t *__restrict__ nv0, int32_t *__restrict__ nv1) {
    const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= Hp) return;
    const double *p = (blockIdx.y ? parent1 : parent0) + (size_t)row * Wp;
    int32_t *o = (blockIdx.y ? nv1 : nv0) + (size_t)row * Wp;
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
    hipLaunchKernelGGL(k_next_valid, dim3((Hp + 3) / 4, 1), dim3(256), 0, st, parent, parent, Wp, Hp, nv, nv);
}
void launch_next_valid2(const double *parent0, const double *parent1, int Wp, int Hp, int32_t *nv0, int32_t *nv1, hipStream_t st) {
    hipLaunchKernelGGL(k_next_valid, dim3((Hp + 3) / 4, 2), dim3(256), 0, st, parent0, parent1, Wp, Hp, nv0, nv1);
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
#define NCC_CH 384  // candidate columns staged per pass
// |integer-form score - reference fp64 score| < 1e-12 for every radius <= 15 (n <= 2883 terms of relative 1.1e-16);
// two candidates closer than this in integer form are decided by k_ncc_exact
#define NCC_TIE_TOL 1e-11

// Running best of the strict '>' scan (.cpp:213) over integer-form scores.  `exact`: the score is the reference's to
// the last bit (a zero-variance window scores exactly 0 there too: u = 0 everywhere, CManageData.cpp:86-89, and the
// initial -1 is a constant); a (near) tie between two such scores needs no second opinion.
struct NccBest {
    double v;
    int c;
    bool exact;
    bool tie;
};
__device__ __forceinline__ void ncc_offer(NccBest &b, double sc, int c, bool exact) {
    if (fabs(sc - b.v) <= NCC_TIE_TOL && !(exact && b.exact)) b.tie = true;
    if (sc > b.v) { // .cpp:213
        b.v = sc;
        b.c = c;
        b.exact = exact;
    }
}
// Filter score (n Sab - Sa Sb) / sqrt(va vb) of one candidate, va = n Saa - Sa^2 > 0 of the own window given as a
// double.  Every integer here is below 2^53, so the fp64 products and differences are exact; 1/sqrt is the
// hardware estimate plus one Newton step (relative error ~2e-16) -- four times cheaper than a correctly rounded
// sqrt and divide, and the filter only has to be within NCC_TIE_TOL / 2 of the reference's score.
// Zero variance on either side -> exactly 0, as in the reference; *exact tells the caller so.
__device__ __forceinline__ double ncc_filter_score(double nd, uint32_t Sab, double Sa, double va, int Sb, int Sbb, bool *exact) {
    const double sb = (double)Sb;
    const double vb = nd * (double)Sbb - sb * sb;
    const double num = nd * (double)Sab - Sa * sb;
    const double p = va * vb;
    const bool nz = va > 0.0 && vb > 0.0;
    double r = __builtin_amdgcn_rsq(nz ? p : 1.0);
    const double e = __builtin_fma(-(p * r), r, 1.0);
    r = __builtin_fma(0.5 * r, e, r);
    *exact = !nz;
    return nz ? num * r : 0.0;
}
// appends the pixels of this wave whose scan saw a (near) tie to the tie list of k_ncc_exact
__device__ __forceinline__ void ncc_tie_append(const StageArgs &a, int cnt_slot, bool tie, uint32_t entry) {
    const unsigned long long mm = __ballot(tie);
    if (mm) {
        const int lane = threadIdx.x & 63, leader = __builtin_ctzll(mm);
        int base = 0;
        if (lane == leader) base = atomicAdd(a.tie_cnt + cnt_slot, __popcll(mm));
        base = __shfl(base, leader);
        if (tie) a.tie_list[base + __popcll(mm & ((1ull << lane) - 1ull))] = entry;
    }
}

// Generic-radius fallback (byte-wise LDS reads); radii 1..7 use k_ncc_dot4 below.
__global__ __launch_bounds__(NCC_TX) void k_ncc_bytes(StageArgs a, int mode, int strideA, int strideB) {
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
    double Sa = 0.0, va = 0.0;
    if (active) {
        Sa = (double)d.S1_own[pix];
        va = (double)n * (double)d.S2_own[pix] - Sa * Sa;
    }
    NccBest bb{-1.0, -1, true, false};
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
                bool ex;
                const double sc = ncc_filter_score((double)n, (uint32_t)Sab, Sa, va, d.S1_oth[(size_t)y * W + c], d.S2_oth[(size_t)y * W + c], &ex);
                ncc_offer(bb, sc, c, ex);
            }
        }
    }
    if (active && bb.c != -1) d.d16_out[pix] = (int16_t)(bb.c - x); // .cpp:219-222 / 301-302 / 563-564
    ncc_tie_append(a, mode == 2, active && bb.tie, (uint32_t)pix | ((uint32_t)blockIdx.z << 31));
}

// ---------------------------------------------------------------- NCC interval argmax, v_dot4_u32_u8
// Both views are read from their BGRX copies (one dword per pixel, X = 0, k_pyramid.hip): a window row is
// 2R+1 aligned dwords and Sab is a chain of v_dot4_u32_u8 with no byte re-alignment.  Candidates are
// evaluated in register-blocked groups of NCC_G consecutive columns: per window row 2R+1 own dwords and
// NCC_G+2R other-view dwords feed NCC_G*(2R+1) dot4s.
//  k_ncc_dot4 : one row x 256 pixels per workgroup, rows of both views staged in LDS, one pixel per lane.
//               Pixels whose interval is longer than NCC_WIDE go to a worklist instead.
//  k_ncc_wide : one workgroup per worklist pixel, its candidates spread over the 256 lanes.  (Whole rows
//               whose parent row is NOMATCH search the entire other margin, .cpp:260-283: a handful of
//               rows carrying as many evaluations as the rest of the level.)
#define NCC_G 5
#define NCC_WIDE 160 // the largest interval the band kernel takes (its LDS staging is sized for it); StageArgs::ncc_wide may lower the threshold
#define RG_SLOTS 512 // row-kernel workgroups per direction in grid.y: every workgroup loops over the listed rows slot by slot (slot, slot + gridDim.y, ...),
                     // so no listed row is ever left out; k_ncc_wide skips every row that k_rg_rows listed (>= RG_MIN wide pixels)
#define RG_MIN 48 // wide pixels of a (direction, row) from which the row is matched by k_ncc_rowgemm instead of k_ncc_wide
#define RG_SLIDE_MIN 1024 // wide pixels of a row from which k_ncc_slide takes it (when its widest interval allows): C2's rows below an empty parent row at
                          // the lower levels (128...1024 pixels x as many candidates) run faster in the GEMM (initial match 1.43 against 1.83 ms)
#define RG_MID_MIN 512 // pixels of a row with an interval beyond the band kernel's crossover (ncc_mid) from which ALL of them join the row
                       // kernel: a short row (C2's lowest level: 128 pixels x 128 candidates) is latency-bound there (initial match 1.40 -> 1.88 ms)

// offers the candidate group c0..c0+G-1 (their G accumulated Sab) to the running best, ascending columns
template <int R>
__device__ __forceinline__ void ncc_score_group(const uint32_t (&acc)[NCC_G], int c0, int cend, const uint8_t *mk,
                                                const int32_t *s1, const int32_t *s2, double Sa, double va, NccBest &b) {
    constexpr int n = (2 * R + 1) * (2 * R + 1) * 3;
#pragma unroll
    for (int g = 0; g < NCC_G; g++) {
        const int c = c0 + g;
        if (c > cend || mk[g] != 255) continue; // .cpp:209
        bool ex;
        const double sc = ncc_filter_score((double)n, acc[g], Sa, va, s1[g], s2[g], &ex);
        ncc_offer(b, sc, c, ex);
    }
}

template <int R>
__global__ __launch_bounds__(NCC_TX) void k_ncc_dot4(StageArgs a, int mode) {
    constexpr int WS = 2 * R + 1, G = NCC_G, NB = G + WS - 1;
    constexpr int SA = NCC_TX + 2 * R;         // dwords per staged own-view row
    constexpr int SB = NCC_CH + 2 * R + G + 3; // dwords per staged other-view row
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    int *s_ctl = (int *)smem; // [0]=min L, [1]=max R over the narrow pixels of the block
    uint32_t *sA = (uint32_t *)(smem + 16);
    uint32_t *sB = sA + WS * SA;
    int32_t *sS1 = (int32_t *)(sB + WS * SB);
    int32_t *sS2 = sS1 + NCC_CH + G;
    uint8_t *sM = (uint8_t *)(sS2 + NCC_CH + G);
    const DirArgs &d = a.d[blockIdx.z];
    const int W = a.W;
    constexpr int n = WS * WS * 3;
    const int y = d.own.YL + blockIdx.y;
    const int x0 = d.own.XL + blockIdx.x * NCC_TX;
    if (y > d.own.YR || x0 > d.own.XR) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int x = x0 + tid;
    const size_t pix = (size_t)y * W + x;

    bool active = (x <= d.own.XR) && (d.mask_own[pix] == 255);
    if (mode == 2 && active) active = (d.d16_in[pix] == NOMATCH); // Rematch: .cpp:538
    int L = 0x7fffffff, Rr = -1;
    if (active) {
        if (mode == 0) {
            L = d.oth.XL; // .cpp:207
            Rr = d.oth.XR;
        } else {
            L = d.BL[pix];
            Rr = d.BR[pix];
        }
        L = max(L, R); // windows that would leave the image are skipped (UB in the reference)
        Rr = min(Rr, W - 1 - R);
        if (L > Rr) active = false;
    }
    // a row that holds RG_MIN or more pixels with an interval longer than ncc_mid goes to a row kernel (k_ncc_rowstat counted them):
    // there every such pixel leaves this kernel, elsewhere only those beyond its own limit
    const int wide_from = (a.opt_no_rowgemm != 1 && a.wrow[NCC_WROW_MID(a.H) + blockIdx.z * a.H + y] >= RG_MID_MIN) ? a.ncc_mid : NCC_WIDE;
    const bool wide = active && (Rr - L + 1 > wide_from);
    // append of the wide pixels to the worklist of k_ncc_wide / the row kernels, ONE pair of atomics per workgroup
    // (per wave, the single list counter -- ~88 same-address atomics per microsecond -- was 2.2 ms of a 12.5 MP frame
    // of wide pixels)
    __shared__ int s_wn[NCC_TX / 64], s_wbase;
    const unsigned long long wmask = __ballot(wide);
    const bool narrow = active && !wide;
    if (lane == 0) s_wn[tid >> 6] = __popcll(wmask);
    if (tid == 0) {
        s_ctl[0] = 0x7fffffff;
        s_ctl[1] = -1;
    }
    __syncthreads();
    {
        int lo = narrow ? L : 0x7fffffff, hi = narrow ? Rr : -1;
        for (int o = 32; o > 0; o >>= 1) {
            lo = min(lo, __shfl_xor(lo, o));
            hi = max(hi, __shfl_xor(hi, o));
        }
        if (lane == 0) {
            atomicMin(&s_ctl[0], lo);
            atomicMax(&s_ctl[1], hi);
        }
        if (tid == 0) {
            int tot = 0;
            for (int w = 0; w < NCC_TX / 64; w++) tot += s_wn[w];
            if (tot) {
                s_wbase = atomicAdd(a.ncc_cnt, tot);
                atomicAdd(a.wrow + blockIdx.z * a.H + y, tot);
            }
        }
    }
    __syncthreads();
    if (wide) {
        int base = s_wbase;
        for (int w = 0; w < (tid >> 6); w++) base += s_wn[w];
        a.rf_list[base + __popcll(wmask & ((1ull << lane) - 1ull))] = (uint32_t)pix | ((uint32_t)blockIdx.z << 31);
    }
    const int cmin = s_ctl[0], cmax = s_ctl[1];
    if (cmax < cmin) return;

    // ---- stage own-view rows: sA[j][i] <-> image column x0 - R + i of row y - R + j
    for (int i = tid; i < SA; i += NCC_TX) {
        const int col = x0 - R + i;
        uint32_t v[WS];
#pragma unroll
        for (int j = 0; j < WS; j++) v[j] = (col >= 0 && col < W) ? d.img4_own[(size_t)(y - R + j) * W + col] : 0u;
#pragma unroll
        for (int j = 0; j < WS; j++) sA[j * SA + i] = v[j];
    }
    double Sa = 0.0, va = 0.0;
    if (narrow) {
        Sa = (double)d.S1_own[pix];
        va = (double)n * (double)d.S2_own[pix] - Sa * Sa;
    }
    NccBest bb{-1.0, -1, true, false};
    for (int lo = cmin; lo <= cmax; lo += NCC_CH) {
        const int hi = min(lo + NCC_CH - 1, cmax);
        __syncthreads();
        // sB[j][i] <-> image column lo - R + i; mask / window sums of candidate columns lo .. hi (+G slack)
        {
            const int ndw = hi - lo + 1 + 2 * R + G;
            for (int i = tid; i < ndw; i += NCC_TX) {
                const int col = lo - R + i;
                uint32_t v[WS];
#pragma unroll
                for (int j = 0; j < WS; j++) v[j] = (col >= 0 && col < W) ? d.img4_oth[(size_t)(y - R + j) * W + col] : 0u;
#pragma unroll
                for (int j = 0; j < WS; j++) sB[j * SB + i] = v[j];
            }
            for (int i = tid; i <= hi - lo + G; i += NCC_TX) {
                const int col = lo + i;
                const bool ok = col <= hi; // columns past hi are group padding: never scored
                const size_t o = (size_t)y * W + (ok ? col : lo);
                sM[i] = ok ? d.mask_oth[o] : (uint8_t)0;
                sS1[i] = d.S1_oth[o];
                sS2[i] = d.S2_oth[o];
            }
        }
        __syncthreads();
        if (narrow) {
            const int c1 = min(Rr, hi);
            for (int c0 = max(L, lo); c0 <= c1; c0 += G) {
                uint32_t acc[G];
#pragma unroll
                for (int g = 0; g < G; g++) acc[g] = 0u;
                const int bo = c0 - lo;
#pragma unroll 1 // one window row per iteration keeps the live set at ~2R+1+NB+G dwords (occupancy)
                for (int j = 0; j < WS; j++) {
                    uint32_t av[WS], bw[NB];
#pragma unroll
                    for (int m = 0; m < WS; m++) av[m] = sA[j * SA + tid + m];
#pragma unroll
                    for (int t = 0; t < NB; t++) bw[t] = sB[j * SB + bo + t];
#pragma unroll
                    for (int m = 0; m < WS; m++)
#pragma unroll
                        for (int g = 0; g < G; g++) acc[g] = __builtin_amdgcn_udot4(av[m], bw[g + m], acc[g], false);
                }
                ncc_score_group<R>(acc, c0, c1, sM + bo, sS1 + bo, sS2 + bo, Sa, va, bb);
            }
        }
    }
    if (narrow && bb.c != -1) d.d16_out[pix] = (int16_t)(bb.c - x); // .cpp:219-222 / 301-302 / 563-564
    ncc_tie_append(a, 0, narrow && bb.tie, (uint32_t)pix | ((uint32_t)blockIdx.z << 31));
}

// One workgroup per wide pixel (worklist of k_ncc_dot4), candidates spread over the 256 lanes, NCC_G consecutive
// ones per lane and pass.  The other view's rows of a pass (256 * NCC_G candidates + window) are staged in LDS with
// coalesced loads first: read straight from global memory every lane would fetch 15 dwords per window row.
template <int R>
__global__ __launch_bounds__(NCC_TX) void k_ncc_wide(StageArgs a, int mode) {
    constexpr int WS = 2 * R + 1, G = NCC_G, NB = G + WS - 1;
    constexpr int n = WS * WS * 3;
    constexpr int CHW = NCC_TX * G;  // candidates per pass
    constexpr int SBW = CHW + 2 * R; // dwords per staged row
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint32_t *sB = (uint32_t *)smem; // [WS][SBW]
    __shared__ uint32_t sAw[WS * WS];
    __shared__ double s_v[NCC_TX / 64];
    __shared__ int s_c[NCC_TX / 64];
    __shared__ int s_x[NCC_TX / 64]; // bit 0: exact, bit 1: tie
    __shared__ uint32_t s_items[NCC_TX];
    __shared__ int s_nitems;
    const int count = *a.ncc_cnt;
    const int W = a.W;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    // Up to 256 list entries at a time: every thread looks at one and keeps it unless its row belongs to the row kernel (a
    // frame of wide pixels is 12.5 M entries to skip: one entry per workgroup and iteration -- two dependent loads -- was
    // 11 ms).  The chunk shrinks with the list, so that a short list of kept pixels (each costs its workgroup a whole
    // candidate scan) still spreads over the grid: with fixed 256-entry chunks a 20 k-entry list kept 80 workgroups busy with
    // up to 256 scans each (C3's elliptic masks: 4.4 ms instead of 0.3).
    const int chunk = min(NCC_TX, max(1, (count + (int)gridDim.x - 1) / (int)gridDim.x));
    for (int base = blockIdx.x * chunk; base < count; base += gridDim.x * chunk) { // uniform
    __syncthreads();
    if (tid == 0) s_nitems = 0;
    __syncthreads();
    if (tid < chunk && base + tid < count) {
        const uint32_t e0 = a.rf_list[base + tid];
        const int y0 = (int)((e0 & 0x7fffffffu) / W);
        if (a.opt_no_rowgemm == 1 || a.wrow[(e0 >> 31) * a.H + y0] < RG_MIN) s_items[atomicAdd(&s_nitems, 1)] = e0;
    }
    __syncthreads();
    const int nitems = s_nitems;
    for (int it = 0; it < nitems; it++) { // uniform
        const uint32_t ent = s_items[it];
        const DirArgs &d = a.d[ent >> 31];
        const size_t pix = ent & 0x7fffffffu;
        const int y = (int)(pix / W), x = (int)(pix % W);
        int L, Rr;
        if (mode == 0) {
            L = d.oth.XL;
            Rr = d.oth.XR;
        } else {
            L = d.BL[pix];
            Rr = d.BR[pix];
        }
        L = max(L, R);
        Rr = min(Rr, W - 1 - R);
        __syncthreads(); // previous item done with sAw / s_v / s_c / sB
        if (tid < WS * WS) sAw[tid] = d.img4_own[(size_t)(y - R + tid / WS) * W + x - R + tid % WS];
        const double Sa = (double)d.S1_own[pix];
        const double va = (double)n * (double)d.S2_own[pix] - Sa * Sa;
        NccBest bb{-1.0, 0x7fffffff, true, false};
        for (int lo = L; lo <= Rr; lo += CHW) { // uniform
            const int hi = min(lo + CHW - 1, Rr);
            const int ndw = hi - lo + 1 + 2 * R + (G - 1); // columns lo - R .. hi + R (+ group padding)
            __syncthreads();
            for (int i = tid; i < ndw; i += NCC_TX) {
                const int col = min(lo - R + i, W - 1); // clamped columns belong to candidates > Rr only
                uint32_t v[WS];
#pragma unroll
                for (int j = 0; j < WS; j++) v[j] = d.img4_oth[(size_t)(y - R + j) * W + col];
#pragma unroll
                for (int j = 0; j < WS; j++) sB[j * (SBW + G) + i] = v[j];
            }
            __syncthreads();
            const int c0 = lo + tid * G;
            if (c0 <= hi) {
                uint32_t acc[G];
#pragma unroll
                for (int g = 0; g < G; g++) acc[g] = 0u;
                const int bo = c0 - lo;
#pragma unroll 1
                for (int j = 0; j < WS; j++) {
                    uint32_t av[WS], bw[NB];
#pragma unroll
                    for (int m = 0; m < WS; m++) av[m] = sAw[j * WS + m];
#pragma unroll
                    for (int t = 0; t < NB; t++) bw[t] = sB[j * (SBW + G) + bo + t];
#pragma unroll
                    for (int m = 0; m < WS; m++)
#pragma unroll
                        for (int g = 0; g < G; g++) acc[g] = __builtin_amdgcn_udot4(av[m], bw[g + m], acc[g], false);
                }
                uint8_t mk[G];
                int32_t s1[G], s2[G];
#pragma unroll
                for (int g = 0; g < G; g++) {
                    const size_t o = (size_t)y * W + min(c0 + g, Rr);
                    mk[g] = d.mask_oth[o];
                    s1[g] = d.S1_oth[o];
                    s2[g] = d.S2_oth[o];
                }
                ncc_score_group<R>(acc, c0, Rr, mk, s1, s2, Sa, va, bb);
            }
        }
        // block argmax; equal scores -> the smaller column (the reference scans ascending with '>'); two lanes'
        // bests within NCC_TIE_TOL of each other are a tie like two candidates of one lane
        double bv = bb.v;
        int bc = bb.c;
        bool bx = bb.exact, tie = bb.tie;
        for (int o = 32; o > 0; o >>= 1) {
            const double ov = __shfl_xor(bv, o);
            const int oc = __shfl_xor(bc, o);
            const bool ox = __shfl_xor((int)bx, o) != 0;
            if (bc != 0x7fffffff && oc != 0x7fffffff && fabs(ov - bv) <= NCC_TIE_TOL && !(ox && bx)) tie = true;
            if (ov > bv || (ov == bv && oc < bc)) {
                bv = ov;
                bc = oc;
                bx = ox;
            }
        }
        tie = __ballot(tie) != 0ull;
        if (lane == 0) {
            s_v[wid] = bv;
            s_c[wid] = bc;
            s_x[wid] = (int)bx | ((int)tie << 1);
        }
        __syncthreads();
        if (tid == 0) {
            for (int w = 1; w < NCC_TX / 64; w++) {
                const bool ox = s_x[w] & 1;
                if (s_x[w] & 2) tie = true;
                if (bc != 0x7fffffff && s_c[w] != 0x7fffffff && fabs(s_v[w] - bv) <= NCC_TIE_TOL && !(ox && bx)) tie = true;
                if (s_v[w] > bv || (s_v[w] == bv && s_c[w] < bc)) {
                    bv = s_v[w];
                    bc = s_c[w];
                    bx = ox;
                }
            }
            if (bv > -1.0) d.d16_out[pix] = (int16_t)(bc - x);
            if (tie) a.tie_list[atomicAdd(a.tie_cnt, 1)] = ent;
        }
    }
    }
}

// Rematch (mode 2): the per-row lists of launch_set_boundary hold the ~1 % of the pixels that are evaluated at all,
// a few candidates each.  One wave per row list, one lane per pixel, windows straight from the BGRX images (L1 / L2 resident rows): staging
// rows in LDS per 256-pixel workgroup, as k_ncc_dot4 does, would be paid by nearly every workgroup for one or two
// active lanes.
template <int R>
__global__ __launch_bounds__(256) void k_ncc_sparse(StageArgs a) {
    constexpr int WS = 2 * R + 1, G = NCC_G, NB = G + WS - 1;
    constexpr int n = WS * WS * 3;
    const int W = a.W, H = a.H;
    const int lane = threadIdx.x & 63;
    // one wave per (direction, margin row) list written by k_setb_horiz (which sets the count of every margin row)
    for (int idx = blockIdx.x * 4 + ((int)threadIdx.x >> 6); idx < a.ndir * H; idx += gridDim.x * 4) {
        const int dir = idx / H, yy = a.d[dir].own.YL + idx % H;
        if (yy > a.d[dir].own.YR) continue; // wave-uniform
        const int row = dir * H + yy;
    for (int item = lane, count = a.ncc_cnt[16 + row]; item < count; item += 64) {
        const DirArgs &d = a.d[dir];
        const size_t pix = a.rf_list[(size_t)row * W + item];
        const int y = (int)(pix / W), x = (int)(pix % W);
        const int L = max((int)d.BL[pix], R), Rr = min((int)d.BR[pix], W - 1 - R);
        const double Sa = (double)d.S1_own[pix];
        const double va = (double)n * (double)d.S2_own[pix] - Sa * Sa;
        NccBest bb{-1.0, -1, true, false};
        for (int c0 = L; c0 <= Rr; c0 += G) {
            uint32_t acc[G];
#pragma unroll
            for (int g = 0; g < G; g++) acc[g] = 0u;
#pragma unroll 1
            for (int j = 0; j < WS; j++) {
                const uint32_t *arow = d.img4_own + (size_t)(y - R + j) * W + x - R;
                const uint32_t *brow = d.img4_oth + (size_t)(y - R + j) * W;
                uint32_t av[WS], bw[NB];
#pragma unroll
                for (int m = 0; m < WS; m++) av[m] = arow[m];
#pragma unroll
                for (int t = 0; t < NB; t++) bw[t] = brow[min(c0 - R + t, W - 1)]; // clamped columns belong to c > Rr only
#pragma unroll
                for (int m = 0; m < WS; m++)
#pragma unroll
                    for (int g = 0; g < G; g++) acc[g] = __builtin_amdgcn_udot4(av[m], bw[g + m], acc[g], false);
            }
            uint8_t mk[G];
            int32_t s1[G], s2[G];
#pragma unroll
            for (int g = 0; g < G; g++) {
                const size_t o = (size_t)y * W + min(c0 + g, Rr);
                mk[g] = d.mask_oth[o];
                s1[g] = d.S1_oth[o];
                s2[g] = d.S2_oth[o];
            }
            ncc_score_group<R>(acc, c0, Rr, mk, s1, s2, Sa, va, bb);
        }
        if (bb.c != -1) d.d16_out[pix] = (int16_t)(bb.c - x); // .cpp:563-564
        if (bb.tie) a.tie_list[atomicAdd(a.tie_cnt + 1, 1)] = (uint32_t)pix | ((uint32_t)dir << 31); // rare: no aggregation
    }
    }
}


// ---------------------------------------------------------------- reference-order re-evaluation of tie pixels
// One wave per listed pixel redoes the reference's candidate scan (.cpp:202-222 / 268-302 / 535-564) the reference's
// way: vecL = (a - mean a) / ||a - mean a|| element by element, per candidate mean / norm / dot in fp64 with
// Armadillo's two-accumulator order (oracle/stereo_oracle.c: orc_window_to_vec, orc_arma_dot), gather order byte
// column outer, window row inner (CManageData.cpp:84-86).  The candidates are spread over the lanes (each lane scans
// its own ascending subsequence with '>'), the wave keeps the largest score and, among equal scores, the smallest
// column -- the first maximum, which is what the sequential strict-'>' scan returns.
// (a - mean a) / norm takes at most 256 values per pixel: a 256-entry table in LDS replaces the n divisions.
#define NCX_MAXR 15
#define NCX_MAXN (3 * (2 * NCX_MAXR + 1) * (2 * NCX_MAXR + 1))
__global__ __launch_bounds__(256) void k_ncc_exact(StageArgs a, int mode) {
    __shared__ double sT[4][256];
    __shared__ uint8_t sWin[4][NCX_MAXN + 13];
    const int count = a.tie_cnt[mode == 2];
    const int W = a.W, r = a.r, w = 2 * r + 1, n = 3 * w * w;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    for (int base = blockIdx.x * 4; base < count; base += gridDim.x * 4) { // block-uniform
        const int item = base + wid;
        const bool on = item < count; // wave-uniform
        const uint32_t ent = a.tie_list[on ? item : base];
        const DirArgs &d = a.d[ent >> 31];
        const size_t pix = ent & 0x7fffffffu;
        const int y = (int)(pix / W), x = (int)(pix % W);
        int L, Rr;
        if (mode == 0) {
            L = d.oth.XL;
            Rr = d.oth.XR;
        } else {
            L = d.BL[pix];
            Rr = d.BR[pix];
        }
        L = max(L, r); // windows that would leave the image are skipped (UB in the reference), as in the scan kernels
        Rr = min(Rr, W - 1 - r);
        __syncthreads(); // previous round done with sT / sWin
        // own window in the reference's vector order: k = j * w + i <-> byte column j of window row i
        for (int k = lane; k < n; k += 64) {
            const int j = k / w, i = k - j * w;
            sWin[wid][k] = d.img_own[((size_t)(y - r + i) * W + (x - r)) * 3 + j];
        }
        __syncthreads();
        const double meanL = (double)d.S1_own[pix] / (double)n; // accumulate(u) / n: the byte sum is exact in fp64
        double acc1 = 0.0, acc2 = 0.0;
        for (int k = 0; k < n; k++) { // every lane runs the same chain (fn_norm.hpp:99-130)
            const double u = (double)sWin[wid][k] - meanL;
            if (k & 1) acc2 += u * u;
            else acc1 += u * u;
        }
        double normL = sqrt(acc1 + acc2);
        if (normL == 0) normL = 1; // CManageData.cpp:89
#pragma unroll
        for (int t = 0; t < 4; t++) sT[wid][lane * 4 + t] = ((double)(lane * 4 + t) - meanL) / normL; // vecL /= normL, .cpp:203
        __syncthreads();
        double bv = -1.0; // .cpp:205
        int bc = 0x7fffffff;
        if (on) {
            for (int c = L + lane; c <= Rr; c += 64) {
                if (d.mask_oth[(size_t)y * W + c] != 255) continue; // .cpp:209
                const double meanR = (double)d.S1_oth[(size_t)y * W + c] / (double)n;
                double m1 = 0.0, m2 = 0.0, d1 = 0.0, d2 = 0.0;
                const uint8_t *b0 = d.img_oth + ((size_t)(y - r) * W + (c - r)) * 3;
                int k = 0;
                for (int j = 0; j < 3 * w; j++)
                    for (int i = 0; i < w; i++, k++) {
                        const double ur = (double)b0[(size_t)i * W * 3 + j] - meanR;
                        const double vl = sT[wid][sWin[wid][k]];
                        if (k & 1) {
                            m2 += ur * ur;
                            d2 += vl * ur;
                        } else {
                            m1 += ur * ur;
                            d1 += vl * ur;
                        }
                    }
                double normR = sqrt(m1 + m2);
                if (normR == 0) normR = 1;
                const double sc = (d1 + d2) / normR; // .cpp:211
                if (sc > bv) {                       // .cpp:213
                    bv = sc;
                    bc = c;
                }
            }
        }
        for (int o = 32; o > 0; o >>= 1) {
            const double ov = __shfl_xor(bv, o);
            const int oc = __shfl_xor(bc, o);
            if (ov > bv || (ov == bv && oc < bc)) {
                bv = ov;
                bc = oc;
            }
        }
        // No candidate beats -1 strictly: the reference leaves d[x] as it was (.cpp:219, :301, :563) -- and "as it was" is
        // NOMATCH in all three callers: both initial matchers start from a NOMATCH-filled map (.cpp:174, :235) and Rematch
        // only scans pixels whose value IS NOMATCH (.cpp:537).  The filter kernel may already have stored its own choice
        // here, so the value is restored explicitly.
        if (on && lane == 0) d.d16_out[pix] = (bc != 0x7fffffff) ? (int16_t)(bc - x) : (int16_t)NOMATCH;
    }
}


// ---------------------------------------------------------------- rows of wide pixels as an int8 GEMM on the matrix cores
// A row whose parent row holds no disparity searches the WHOLE other margin for every pixel (.cpp:260-283; at the lowest
// level of a 256-disparity rig every pixel does, .cpp:207): for such a row Sab(x, c) = <window(x), window(c)> over all
// pixels x and all candidates c is a dense [pixels x candidates x n] product -- the one place on this path where the
// patch correlation IS a tile GEMM (north_star).  v_mfma_i32_16x16x64_i8 takes signed bytes: both views are staged as
// b - 128 (BGRX dwords ^ 0x00808080, X = 0), then
//     Sab = acc + 128 (Sa + Sb) - 16384 n        (exact int32; |acc| <= n 2^14)
// K runs over the window as 16-byte pieces (4 pixels of one window row; the pixels past the window edge are zeroed in
// the A operand): lane l feeds piece 4 ks + (l >> 4) of pixel / candidate (l & 15) -- the layout of the instruction
// (tests/micro/mfma_i8_probe.hip).  A wave owns 16 pixels x 64 candidates per step (1 x 4 tiles, 4 MFMAs per 20 LDS
// dwords); the epilogue turns the 16 accumulators of a lane into filter scores scaled by the pixel's own sqrt(va)
// (constant per pixel, so the argmax and -- with the tolerance scaled alike -- the tie test are unchanged) and keeps the
// running best of 4 pixels per lane; candidates ascend per lane, lanes are merged at the end (largest score, then
// smallest column).  Near ties go to k_ncc_exact like everywhere else.
#define RG_T 1                 // 16-pixel tiles per wave
#define RG_PX (4 * 16 * RG_T)   // pixels per workgroup: 4 waves x RG_T tiles
#define RG_CC 512 // candidates per staged chunk (the staging latency is exposed once per chunk)
typedef int rg_v4i __attribute__((ext_vector_type(4)));
struct __attribute__((packed, aligned(4))) RgQuad {
    uint32_t a, b, c, d;
};
// Per (direction, row), BEFORE the matchers run: how many pixels search an interval longer than ncc_mid, and the widest interval.
// A row with RG_MIN or more of them is matched by a row kernel (every such pixel of it: the row kernels run at 260-330 G
// pixel-candidate pairs per second from ~65 candidates on, the 5-candidate band kernel at ~190), and which one follows from the
// widest interval (k_rg_rows).  One wave per row, 64 pixels per round.
__global__ __launch_bounds__(256) void k_ncc_rowstat(StageArgs a, int mode) {
    const DirArgs &d = a.d[blockIdx.z];
    const int y = d.own.YL + (int)blockIdx.x * 4 + (int)(threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (y > d.own.YR) return;
    const int W = a.W, R = a.r;
    int cnt = 0, wmax = 0;
    for (int xb = d.own.XL; xb <= d.own.XR; xb += 64) { // uniform
        const int x = xb + lane;
        int width = 0;
        if (x <= d.own.XR) {
            const size_t pix = (size_t)y * W + x;
            if (d.mask_own[pix] == 255) {
                int L = mode == 0 ? d.oth.XL : (int)d.BL[pix], Rr = mode == 0 ? d.oth.XR : (int)d.BR[pix];
                L = max(L, R);
                Rr = min(Rr, W - 1 - R);
                width = max(Rr - L + 1, 0);
            }
        }
        cnt += __popcll(__ballot(width > a.ncc_mid));
        wmax = max(wmax, width);
    }
    for (int o = 32; o > 0; o >>= 1) wmax = max(wmax, __shfl_xor(wmax, o));
    if (lane == 0) {
        a.wrow[NCC_WROW_MID(a.H) + blockIdx.z * a.H + y] = cnt;
        a.wrow[NCC_WROW_MAX(a.H) + blockIdx.z * a.H + y] = wmax;
    }
}
// the rows of a direction with at least RG_MIN wide pixels, in order, as two lists ([0] = count, then the rows): list 0 for
// k_ncc_rowgemm (rows whose widest interval exceeds ncc_slide_max: the int8 GEMM wins from ~500 candidates on), list 1 for
// k_ncc_slide.  `force`: 2 = every row to the GEMM, 3 = every row to the sliding sums (A/B, tests).
__global__ __launch_bounds__(64) void k_rg_rows(StageArgs a, int force) {
    const DirArgs &d = a.d[blockIdx.x];
    int32_t *list0 = a.wrow + NCC_WROW_LIST(a.H, 0) + blockIdx.x * (a.H + 1), *list1 = a.wrow + NCC_WROW_LIST(a.H, 1) + blockIdx.x * (a.H + 1);
    const int lane = threadIdx.x;
    int seen0 = 0, seen1 = 0;
    for (int yb = d.own.YL; yb <= d.own.YR; yb += 64) { // uniform
        const int yy = yb + lane;
        const bool q = yy <= d.own.YR && a.wrow[blockIdx.x * a.H + yy] >= RG_MIN;
        // the sliding sums pay on long rows (a workgroup = four 128-column tiles) of moderate intervals; short rows are latency-bound there
        const bool slide = force == 3 || (force != 2 && q && a.wrow[NCC_WROW_MAX(a.H) + blockIdx.x * a.H + yy] <= a.ncc_slide_max &&
                                          a.wrow[blockIdx.x * a.H + yy] >= RG_SLIDE_MIN);
        const unsigned long long m0 = __ballot(q && !slide), m1 = __ballot(q && slide);
        if (q && !slide) list0[1 + seen0 + __popcll(m0 & ((1ull << lane) - 1ull))] = yy;
        if (q && slide) list1[1 + seen1 + __popcll(m1 & ((1ull << lane) - 1ull))] = yy;
        seen0 += __popcll(m0);
        seen1 += __popcll(m1);
    }
    if (lane == 0) {
        list0[0] = seen0;
        list1[0] = seen1;
    }
}
template <int R>
__global__ __launch_bounds__(256) void k_ncc_rowgemm(StageArgs a, int mode) {
    constexpr int WS = 2 * R + 1, n = 3 * WS * WS, PQ = (WS + 3) / 4, NP = WS * PQ, KS = (NP + 3) / 4;
    constexpr int SA = RG_PX + 4 * PQ, SB = RG_CC + 4 * PQ;
    __shared__ uint32_t sA[WS * SA], sB[WS * SB];
    __shared__ double sRvb[RG_CC], sPSa[RG_PX], sPsv[RG_PX]; // candidates: 1 / sqrt(vb); pixels: Sa, sqrt(va)
    __shared__ int sSb[RG_CC], sPL[RG_PX], sPR[RG_PX];       // candidates: Sb; pixels: their interval (empty: not wide)
    __shared__ uint8_t sOk[RG_CC];
    __shared__ int s_lohi[2];
    const DirArgs &d = a.d[blockIdx.z];
    const int W = a.W;
    const int x0 = d.own.XL + blockIdx.x * RG_PX;
    if (x0 > d.own.XR) return;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, lr = lane & 15, lg = lane >> 4;
    // blockIdx.y = slot: this workgroup takes the slot-th row of the direction with at least RG_MIN wide pixels (the few
    // wide pixels of other rows stay with k_ncc_wide), from the list k_rg_rows made.
    const int32_t *rowlist = a.wrow + NCC_WROW_LIST(a.H, 0) + blockIdx.z * (a.H + 1); // [0] = number of rows, then the rows
    const int nrows = rowlist[0];
    for (int slot = blockIdx.y; slot < nrows; slot += gridDim.y) { // uniform
        __syncthreads(); // the previous row is done with the shared arrays
        const int y = rowlist[1 + slot];
        const int wide_from = a.wrow[NCC_WROW_MID(a.H) + blockIdx.z * a.H + y] >= RG_MID_MIN ? a.ncc_mid : NCC_WIDE; // as k_ncc_dot4 chose for this row
        if (tid == 0) {
            s_lohi[0] = 0x7fffffff;
            s_lohi[1] = -1;
        }
        __syncthreads();
        if (tid < RG_PX) { // this workgroup's pixels
            const int x = x0 + tid;
            const size_t pix = (size_t)y * W + min(x, W - 1);
            bool active = (x <= d.own.XR) && (d.mask_own[pix] == 255);
            int L = 0x7fffffff, Rr = -1;
            if (active) {
                if (mode == 0) {
                    L = d.oth.XL; // .cpp:207
                    Rr = d.oth.XR;
                } else {
                    L = d.BL[pix];
                    Rr = d.BR[pix];
                }
                L = max(L, R);
                Rr = min(Rr, W - 1 - R);
                if (L > Rr) active = false;
            }
            const bool wide = active && (Rr - L + 1 > wide_from); // exactly k_ncc_dot4's test
            const double Sa = wide ? (double)d.S1_own[pix] : 0.0;
            const double va = wide ? (double)n * (double)d.S2_own[pix] - Sa * Sa : 0.0;
            sPSa[tid] = Sa;
            sPsv[tid] = va > 0.0 ? sqrt(va) : -1.0; // -1: zero variance -- every score is exactly 0, the scale is 1
            sPL[tid] = wide ? L : 0x7fffffff;
            sPR[tid] = wide ? Rr : -1;
            int lo_ = wide ? L : 0x7fffffff, hi_ = wide ? Rr : -1;
            for (int o = 32; o > 0; o >>= 1) {
                lo_ = min(lo_, __shfl_xor(lo_, o));
                hi_ = max(hi_, __shfl_xor(hi_, o));
            }
            if (lane == 0) {
                atomicMin(&s_lohi[0], lo_);
                atomicMax(&s_lohi[1], hi_);
            }
        }
        // own-view rows as signed bytes: sA[j][i] <-> column x0 - R + i of row y - R + j
        for (int i = tid; i < SA; i += 256) {
            const int col = x0 - R + i;
#pragma unroll
            for (int j = 0; j < WS; j++) sA[j * SA + i] = ((col >= 0 && col < W) ? d.img4_own[(size_t)(y - R + j) * W + col] : 0u) ^ 0x00808080u;
        }
        __syncthreads();
        const int cmin = s_lohi[0], cmax = s_lohi[1];
        // running best of this lane's 4 RG_T pixels e = 4 t + r: tile t, accumulator register r -> pixel wv * 16 RG_T + t * 16 + 4 lg + r
        double bestv[4 * RG_T];
        int bestc[4 * RG_T];
        unsigned exact = 0xffu, tie = 0u; // the initial -1 is exact
#pragma unroll
        for (int e = 0; e < 4 * RG_T; e++) {
            const double sv = sPsv[wv * (16 * RG_T) + (e >> 2) * 16 + lg * 4 + (e & 3)];
            bestv[e] = sv > 0.0 ? -sv : -1.0; // .cpp:205, scaled
            bestc[e] = 0x7fffffff;
        }
        for (int lo = cmin; lo <= cmax; lo += RG_CC) { // uniform (no trip when the chunk holds no wide pixel)
            __syncthreads();
            // only the 64-candidate blocks the chunk really has are staged (a 257-candidate row fills half a chunk)
            const int ncs = min(RG_CC, (cmax - lo + 64) & ~63);
            for (int i = tid; i < ncs + 4 * PQ; i += 256) {
                const int col = lo - R + i;
#pragma unroll
                for (int j = 0; j < WS; j++) sB[j * SB + i] = ((col >= 0 && col < W) ? d.img4_oth[(size_t)(y - R + j) * W + col] : 0u) ^ 0x00808080u;
            }
            for (int ci = tid; ci < ncs; ci += 256) {
                const int c = lo + ci;
                const bool ok = c <= cmax && c >= R && c <= W - 1 - R && d.mask_oth[(size_t)y * W + c] == 255; // .cpp:209
                const size_t o = (size_t)y * W + (ok ? c : min(max(lo, 0), W - 1));
                const int Sb = d.S1_oth[o];
                const double vb = (double)n * (double)d.S2_oth[o] - (double)Sb * (double)Sb;
                double r = __builtin_amdgcn_rsq(vb > 0.0 ? vb : 1.0);
                r = __builtin_fma(0.5 * r, __builtin_fma(-(vb * r), r, 1.0), r); // one Newton step: ~2e-16 relative
                sOk[ci] = ok;
                sSb[ci] = Sb;
                sRvb[ci] = vb > 0.0 ? r : 0.0; // zero variance: score exactly 0
            }
            __syncthreads();
#pragma unroll 1
            for (int cb = 0; cb < RG_CC / 64 && lo + cb * 64 <= cmax; cb++) { // uniform
                rg_v4i acc[RG_T][4];
#pragma unroll
                for (int t = 0; t < RG_T; t++)
#pragma unroll
                    for (int u = 0; u < 4; u++) acc[t][u] = rg_v4i{0, 0, 0, 0};
#pragma unroll 2 // two k-steps of fragments in flight (fully unrolled, the hoisted gathers of all KS steps cost hundreds of registers)
                for (int ks = 0; ks < KS; ks++) {
                    const int p = 4 * ks + lg, j = min(p / PQ, WS - 1), q = p % PQ;
                    rg_v4i fa[RG_T], fb[4];
                    // a K piece = four consecutive dwords of a staged row: one 16-byte load (two ds_read2_b32: the start is
                    // only dword-aligned), the own-view piece masked where it runs past the window (or past the last piece)
                    const bool live = p < NP;
#pragma unroll
                    for (int t = 0; t < RG_T; t++) {
                        const int xl = wv * (16 * RG_T) + t * 16 + lr;
                        const RgQuad v = *(const RgQuad *)&sA[j * SA + xl + 4 * q];
                        fa[t][0] = live ? (int)v.a : 0;
                        fa[t][1] = (live && 4 * q + 1 < WS) ? (int)v.b : 0;
                        fa[t][2] = (live && 4 * q + 2 < WS) ? (int)v.c : 0;
                        fa[t][3] = (live && 4 * q + 3 < WS) ? (int)v.d : 0;
                    }
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        const int cl = cb * 64 + u * 16 + lr;
                        const RgQuad v = *(const RgQuad *)&sB[j * SB + cl + 4 * q];
                        fb[u] = rg_v4i{(int)v.a, (int)v.b, (int)v.c, (int)v.d}; // (a dead piece is zeroed on the own-view side)
                    }
#pragma unroll
                    for (int t = 0; t < RG_T; t++)
#pragma unroll
                        for (int u = 0; u < 4; u++) acc[t][u] = __builtin_amdgcn_mfma_i32_16x16x64_i8(fa[t], fb[u], acc[t][u], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0); // keep the later steps' gathers from being hoisted up here (registers)
                }
                // epilogue: D[4 lg + r][lr] of tile (t, u) = pixel e = 4 t + r of this lane x candidate cb * 64 + u * 16 + lr
#pragma unroll
                for (int e = 0; e < 4 * RG_T; e++) {
                    const int xl = wv * (16 * RG_T) + (e >> 2) * 16 + lg * 4 + (e & 3);
                    const int Lp = sPL[xl], Rp = sPR[xl];
                    const double Sa = sPSa[xl], sv = sPsv[xl];
                    const double tol = NCC_TIE_TOL * (sv > 0.0 ? sv : 1.0);
                    const int Sai = (int)Sa;
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        const int cl = cb * 64 + u * 16 + lr, c = lo + cl;
                        if (!sOk[cl] || c < Lp || c > Rp) continue;
                        const int Sb = sSb[cl];
                        const double rvb = sRvb[cl];
                        const int Sab = acc[e >> 2][u][e & 3] + 128 * (Sai + Sb) - 16384 * n;
                        const double sc = ((double)n * (double)Sab - Sa * (double)Sb) * rvb; // score * sqrt(va)
                        const bool exc = rvb == 0.0 || !(sv > 0.0); // a zero-variance window on either side: exactly 0
                        if (fabs(sc - bestv[e]) <= tol && !(exc && ((exact >> e) & 1u))) tie |= 1u << e;
                        if (sc > bestv[e]) { // .cpp:213
                            bestv[e] = sc;
                            bestc[e] = c;
                            exact = (exact & ~(1u << e)) | ((unsigned)exc << e);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0); // one pixel's constants live at a time
                }
            }
        }
        // merge the 16 lanes that share the pixels (same lg): largest score, then smallest column
#pragma unroll
        for (int e = 0; e < 4 * RG_T; e++) {
            const int xl = wv * (16 * RG_T) + (e >> 2) * 16 + lg * 4 + (e & 3), x = x0 + xl;
            const double sv = sPsv[xl];
            const double tol = NCC_TIE_TOL * (sv > 0.0 ? sv : 1.0);
            double bv = bestv[e];
            int bc = bestc[e];
            bool bx = (exact >> e) & 1u, tt = (tie >> e) & 1u;
            for (int o = 1; o < 16; o <<= 1) {
                const double ov = __shfl_xor(bv, o);
                const int oc = __shfl_xor(bc, o);
                const bool ox = __shfl_xor((int)bx, o) != 0;
                if (bc != 0x7fffffff && oc != 0x7fffffff && fabs(ov - bv) <= tol && !(ox && bx)) tt = true;
                if (ov > bv || (ov == bv && oc < bc)) {
                    bv = ov;
                    bc = oc;
                    bx = ox;
                }
            }
            for (int o = 1; o < 16; o <<= 1) tt |= __shfl_xor((int)tt, o) != 0;
            if (lr == 0 && sPR[xl] >= 0) { // a wide pixel
                const size_t pix = (size_t)y * W + x;
                if (bc != 0x7fffffff) d.d16_out[pix] = (int16_t)(bc - x); // .cpp:219-222 / 301-302
                if (tt) a.tie_list[atomicAdd(a.tie_cnt, 1)] = (uint32_t)pix | ((uint32_t)blockIdx.z << 31);
            }
        }
    }
}

// ---------------------------------------------------------------- rows of wide pixels by sliding window sums
// The same rows k_ncc_rowgemm takes, without the matrix cores: for a fixed disparity D = c - x the products of two
// window columns do not depend on the pixel,
//     Sab(x, x + D) = sum_{i = -R..R} V_D[x + i],      V_D[u] = sum_j <A[j][u], B[j][u + D]>   (one v_dot4_u32_u8 per row j)
// so a pixel-candidate pair costs 2R + 1 dot4s and a (2R + 1)-term box sum instead of (2R + 1)^2 dot4s (15 x 15: 15 + ~9
// integer operations against 225 dot4s, or against 4 MFMAs + 20 LDS reads per 16 x 64 tile in k_ncc_rowgemm, which the
// operand gathers bound).  Wide pixels of a row all scan (nearly) the same candidate range, so the D planes are shared.
// Mapping: a wave owns SL_COLS = 128 columns (2 per lane: the own-view window columns stay in 2 (2R + 1) registers for
// the whole tile) and walks D upwards two planes at a time; per row j it reads ONE aligned ds_read_b64 of the other view
// (the next two dwords of its sliding 4-dword window) for four dot4s; V goes through a small LDS line from which every
// lane takes the 2R + 2 consecutive values of its two box sums; the per-candidate constants (Sb, 1 / sqrt(vb), validity)
// slide the same way.  The 4 waves of a workgroup take the quarters of the tile's D range (their own staged chunks of
// the other view's rows, no workgroup barrier in the loop) and merge their bests at the end; the inner 128 - 2R columns
// of a tile are its pixels.  Scores, running best, tie flags: exactly k_ncc_rowgemm's epilogue (scaled filter score).
#define SL_COLS 128
#define SL_DC 64 // disparity planes per staged chunk (even)
template <int R>
__global__ __launch_bounds__(256) void k_ncc_slide(StageArgs a, int mode) {
    constexpr int WS = 2 * R + 1, n = 3 * WS * WS;
    constexpr int NBC = SL_COLS + SL_DC + 4;         // other-view columns / candidates staged per chunk
    constexpr int NV = SL_COLS + 2 * R + 2;          // V line
    constexpr int PXT = SL_COLS - 2 * R;             // pixels of a tile
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    // per wave: sB[WS][NBC] u32 | sRvb[NBC] f64 | sSb[NBC] i32 | sV[2][NV] i32
    constexpr size_t WAVE_BYTES = ((size_t)WS * NBC * 4 + (size_t)NBC * 8 + (size_t)NBC * 4 + (size_t)2 * NV * 4 + 15) & ~(size_t)15;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    uint8_t *wbase = smem + wv * WAVE_BYTES;
    double *sRvb = (double *)wbase;                              // -1: not a candidate, 0: zero variance (score exactly 0)
    uint32_t *sB = (uint32_t *)(wbase + (size_t)NBC * 8);
    int *sSb = (int *)(sB + WS * NBC);
    int *sV = sSb + NBC;
    // merge area behind the four wave regions
    double *mV = (double *)(smem + 4 * WAVE_BYTES); // [4][SL_COLS]
    int *mC = (int *)(mV + 4 * SL_COLS);            // [4][SL_COLS]
    int *mF = mC + 4 * SL_COLS;                     // [4][SL_COLS]: bit 0 exact, bit 1 tie
    const DirArgs &d = a.d[blockIdx.z];
    const int W = a.W;
    const int32_t *rowlist = a.wrow + NCC_WROW_LIST(a.H, 1) + blockIdx.z * (a.H + 1); // [0] = number of rows, then the rows (k_rg_rows)
    const int nrows = rowlist[0];
    __shared__ int s_planes[4];
    for (int q = lane; q < 2 * NV; q += 64) sV[q] = 0;
    for (int slot = blockIdx.y; slot < nrows; slot += gridDim.y) { // uniform
        const int y = rowlist[1 + slot];
        const int wide_from = a.wrow[NCC_WROW_MID(a.H) + blockIdx.z * a.H + y] >= RG_MID_MIN ? a.ncc_mid : NCC_WIDE; // as k_ncc_dot4 chose for this row
        // A workgroup owns FOUR consecutive tiles.  How its 4 waves share them depends on the disparity range the tiles
        // span (device data: a pre-pass measures it): up to ~2 chunks of planes per tile every wave takes a tile of its own
        // and walks all its planes (setup and merge amortised over 4 times the work: 257 candidates are 65 planes per
        // wave otherwise); longer ranges are split 2- or 4-way over the waves, tile after tile.
        {
            const int ut = d.own.XL - R + (4 * blockIdx.x + wv) * PXT;
            int dlo = 0x7fffffff, dhi = -0x7fffffff;
#pragma unroll
            for (int e = 0; e < 2; e++) {
                const int t = 2 * lane + e, x = ut + t;
                const bool inside = t >= R && t < SL_COLS - R && x <= d.own.XR;
                const size_t pix = (size_t)y * W + min(max(x, 0), W - 1);
                if (!(inside && d.mask_own[pix] == 255)) continue;
                int L = mode == 0 ? d.oth.XL : (int)d.BL[pix], Rr = mode == 0 ? d.oth.XR : (int)d.BR[pix];
                L = max(L, R);
                Rr = min(Rr, W - 1 - R);
                if (Rr - L + 1 > wide_from) {
                    dlo = min(dlo, L - x);
                    dhi = max(dhi, Rr - x);
                }
            }
            for (int o = 32; o > 0; o >>= 1) {
                dlo = min(dlo, __shfl_xor(dlo, o));
                dhi = max(dhi, __shfl_xor(dhi, o));
            }
            __syncthreads(); // the previous row is done with s_planes and the merge area
            if (lane == 0) s_planes[wv] = dhi >= dlo ? dhi - dlo + 1 : 0;
            __syncthreads();
        }
        const int pmax = max(max(s_planes[0], s_planes[1]), max(s_planes[2], s_planes[3]));
        const int nsplit = pmax > 12 * SL_DC ? 4 : (pmax > 6 * SL_DC ? 2 : 1); // uniform over the workgroup
        const int part = wv % nsplit;
        for (int g = 0; g < nsplit; g++) { // uniform
        const int tile = 4 * blockIdx.x + g * (4 / nsplit) + wv / nsplit;
        const int u0 = d.own.XL - R + tile * PXT; // image column of tile column 0
        // ---- this lane's two pixels (tile columns 2 lane, 2 lane + 1)
        int Lp[2], Rp[2];
        double Sa[2], sv[2], bestv[2];
        int bestc[2];
        unsigned exact = 3u, tie = 0u;
        int dlo = 0x7fffffff, dhi = -0x7fffffff;
#pragma unroll
        for (int e = 0; e < 2; e++) {
            const int t = 2 * lane + e, x = u0 + t;
            const bool inside = t >= R && t < SL_COLS - R && x <= d.own.XR;
            const size_t pix = (size_t)y * W + min(max(x, 0), W - 1);
            bool active = inside && d.mask_own[pix] == 255;
            int L = 0x7fffffff, Rr = -1;
            if (active) {
                if (mode == 0) {
                    L = d.oth.XL; // .cpp:207
                    Rr = d.oth.XR;
                } else {
                    L = d.BL[pix];
                    Rr = d.BR[pix];
                }
                L = max(L, R);
                Rr = min(Rr, W - 1 - R);
                if (L > Rr) active = false;
            }
            const bool wide = active && (Rr - L + 1 > wide_from); // exactly k_ncc_dot4's test
            const double s1 = wide ? (double)d.S1_own[pix] : 0.0;
            const double va = wide ? (double)n * (double)d.S2_own[pix] - s1 * s1 : 0.0;
            Sa[e] = s1;
            sv[e] = va > 0.0 ? sqrt(va) : -1.0; // -1: zero variance -- every score is exactly 0, the scale is 1
            Lp[e] = wide ? L : 0x7fffffff;
            Rp[e] = wide ? Rr : -1;
            bestv[e] = sv[e] > 0.0 ? -sv[e] : -1.0; // .cpp:205, scaled
            bestc[e] = 0x7fffffff;
            if (wide) {
                dlo = min(dlo, L - x);
                dhi = max(dhi, Rr - x);
            }
        }
        for (int o = 32; o > 0; o >>= 1) {
            dlo = min(dlo, __shfl_xor(dlo, o));
            dhi = max(dhi, __shfl_xor(dhi, o));
        }
        if (dhi >= dlo) { // uniform over the waves that share the tile
            dlo &= ~1; // even: the sliding reads stay 8-byte aligned
            int span = (dhi - dlo + 1 + nsplit - 1) / nsplit;
            span = (span + 1) & ~1;
            const int D0 = dlo + part * span, D1 = min(dhi, D0 + span - 1); // this wave's planes
            // own-view window columns of this lane's two tile columns
            uint32_t a0[WS], a1[WS];
#pragma unroll
            for (int j = 0; j < WS; j++) {
                const int c0 = u0 + 2 * lane, c1 = c0 + 1;
                const uint32_t *row = d.img4_own + (size_t)(y - R + j) * W;
                a0[j] = (c0 >= 0 && c0 < W) ? row[c0] : 0u;
                a1[j] = (c1 >= 0 && c1 < W) ? row[c1] : 0u;
            }
            // the wave's planes in equal chunks of at most SL_DC (a last chunk of a few planes would pay a whole staging)
            const int planes = D1 - D0 + 1;
            const int nch = planes > 0 ? (planes + SL_DC - 1) / SL_DC : 0;
            const int per = nch ? (((planes + nch - 1) / nch) + 1) & ~1 : 0; // even
            for (int Dc = D0; Dc <= D1; Dc += per) { // uniform
                // ---- stage the chunk: other-view columns u0 + Dc + i, i < NBC, and their candidate constants.  Every
                // load of the chunk is issued before the first LDS write (one memory round trip, not one per 64 columns).
                const int cb = u0 + Dc;
                constexpr int NI = (NBC + 63) / 64;
                __builtin_amdgcn_wave_barrier();
                {
                    uint32_t v[NI][WS];
                    int sb_[NI], s2_[NI];
                    uint8_t mk_[NI];
#pragma unroll
                    for (int ii = 0; ii < NI; ii++) {
                        const int i = lane + 64 * ii, col = cb + i;
                        const bool in = i < NBC && col >= 0 && col < W;
                        const size_t o = (size_t)y * W + (in ? col : 0);
#pragma unroll
                        for (int j = 0; j < WS; j++) v[ii][j] = in ? d.img4_oth[(size_t)(y - R + j) * W + col] : 0u;
                        sb_[ii] = d.S1_oth[o];
                        s2_[ii] = d.S2_oth[o];
                        mk_[ii] = d.mask_oth[o];
                    }
#pragma unroll
                    for (int ii = 0; ii < NI; ii++) {
                        const int i = lane + 64 * ii, col = cb + i;
                        if (i >= NBC) continue;
                        const bool in = col >= 0 && col < W;
#pragma unroll
                        for (int j = 0; j < WS; j++) sB[j * NBC + i] = v[ii][j];
                        const bool ok = in && col >= R && col <= W - 1 - R && mk_[ii] == 255; // .cpp:209
                        const int Sb = sb_[ii];
                        const double vb = (double)n * (double)s2_[ii] - (double)Sb * (double)Sb;
                        double r = __builtin_amdgcn_rsq(vb > 0.0 ? vb : 1.0);
                        r = __builtin_fma(0.5 * r, __builtin_fma(-(vb * r), r, 1.0), r); // one Newton step: ~2e-16 relative
                        sSb[i] = Sb;
                        sRvb[i] = !ok ? -1.0 : (vb > 0.0 ? r : 0.0);
                    }
                }
                __builtin_amdgcn_wave_barrier();
                // sliding state at plane Dc: other-view dwords at tile columns 2 lane, 2 lane + 1 (shifted by the plane),
                // candidate constants of c = x_a + D, x_a + D + 1
                uint32_t b0[WS], b1[WS];
#pragma unroll
                for (int j = 0; j < WS; j++) {
                    const uint2 v = *(const uint2 *)&sB[j * NBC + 2 * lane];
                    b0[j] = v.x;
                    b1[j] = v.y;
                }
                int2 sb01 = *(const int2 *)&sSb[2 * lane];
                double2 rv01 = *(const double2 *)&sRvb[2 * lane];
                const int kmax = min(per, D1 - Dc + 1); // planes of this chunk (may be odd: the surplus plane is masked by Rp)
#pragma unroll 1
                for (int k = 0; k < kmax; k += 2) {
                    const int D = Dc + k;
                    uint32_t v00 = 0u, v10 = 0u, v01 = 0u, v11 = 0u; // V[tile column][plane]
                    const int io = 2 * lane + k + 2;
#pragma unroll
                    for (int j = 0; j < WS; j++) {
                        const uint2 nx = *(const uint2 *)&sB[j * NBC + io];
                        v00 = __builtin_amdgcn_udot4(a0[j], b0[j], v00, false);
                        v10 = __builtin_amdgcn_udot4(a1[j], b1[j], v10, false);
                        v01 = __builtin_amdgcn_udot4(a0[j], b1[j], v01, false);
                        v11 = __builtin_amdgcn_udot4(a1[j], nx.x, v11, false);
                        b0[j] = nx.x;
                        b1[j] = nx.y;
                    }
                    const int2 sb23 = *(const int2 *)&sSb[io];
                    const double2 rv23 = *(const double2 *)&sRvb[io];
                    // ---- box sums over the tile columns: V line index = tile column + R
                    sV[2 * lane + R] = (int)v00;
                    sV[2 * lane + 1 + R] = (int)v10;
                    sV[NV + 2 * lane + R] = (int)v01;
                    sV[NV + 2 * lane + 1 + R] = (int)v11;
                    __builtin_amdgcn_wave_barrier();
                    int S[2][2]; // [plane][pixel]
#pragma unroll
                    for (int pl = 0; pl < 2; pl++) {
                        int w[2 * R + 2];
#pragma unroll
                        for (int m = 0; m < R + 1; m++) {
                            const int2 v = *(const int2 *)&sV[pl * NV + 2 * lane + 2 * m];
                            w[2 * m] = v.x;
                            w[2 * m + 1] = v.y;
                        }
                        int s0 = 0;
#pragma unroll
                        for (int m = 0; m < 2 * R + 1; m++) s0 += w[m];
                        S[pl][0] = s0;
                        S[pl][1] = s0 - w[0] + w[2 * R + 1];
                    }
                    __builtin_amdgcn_wave_barrier();
                    // ---- scores: pixel e, plane pl -> candidate c = x_a + e + D + pl, constants index e + pl of (01 | 23)
                    const int sbv[3] = {sb01.x, sb01.y, sb23.x};
                    const double rvv[3] = {rv01.x, rv01.y, rv23.x};
#pragma unroll
                    for (int e = 0; e < 2; e++) {
                        const double tol = NCC_TIE_TOL * (sv[e] > 0.0 ? sv[e] : 1.0);
                        const int Sai = (int)Sa[e];
#pragma unroll
                        for (int pl = 0; pl < 2; pl++) { // ascending candidates
                            const int c = u0 + 2 * lane + e + D + pl;
                            const double rvb = rvv[e + pl];
                            if (rvb < 0.0 || c < Lp[e] || c > Rp[e]) continue;
                            const int Sb = sbv[e + pl];
                            (void)Sai;
                            const double sc = ((double)n * (double)S[pl][e] - Sa[e] * (double)Sb) * rvb; // score * sqrt(va)
                            const bool exc = rvb == 0.0 || !(sv[e] > 0.0); // a zero-variance window on either side: exactly 0
                            if (fabs(sc - bestv[e]) <= tol && !(exc && ((exact >> e) & 1u))) tie |= 1u << e;
                            if (sc > bestv[e]) { // .cpp:213
                                bestv[e] = sc;
                                bestc[e] = c;
                                exact = (exact & ~(1u << e)) | ((unsigned)exc << e);
                            }
                        }
                    }
                    sb01 = sb23;
                    rv01 = rv23;
                }
            }
        }
        // ---- merge the waves that shared the tile (ascending plane ranges = ascending candidates): largest score, then
        // smallest column
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 2; e++) {
            mV[wv * SL_COLS + 2 * lane + e] = bestv[e];
            mC[wv * SL_COLS + 2 * lane + e] = bestc[e];
            mF[wv * SL_COLS + 2 * lane + e] = (int)(((exact >> e) & 1u) | (((tie >> e) & 1u) << 1));
        }
        __syncthreads();
        if (part == 0) {
#pragma unroll
            for (int e = 0; e < 2; e++) {
                if (Rp[e] < 0) continue; // not a wide pixel of this tile
                const int t = 2 * lane + e, x = u0 + t;
                const double tol = NCC_TIE_TOL * (sv[e] > 0.0 ? sv[e] : 1.0);
                double bv = bestv[e];
                int bc = bestc[e];
                bool bx = (exact >> e) & 1u, tt = (tie >> e) & 1u;
                for (int w2 = wv + 1; w2 < wv + nsplit; w2++) {
                    const double ov = mV[w2 * SL_COLS + t];
                    const int oc = mC[w2 * SL_COLS + t], of = mF[w2 * SL_COLS + t];
                    const bool ox = of & 1;
                    tt |= (of >> 1) & 1;
                    if (bc != 0x7fffffff && oc != 0x7fffffff && fabs(ov - bv) <= tol && !(ox && bx)) tt = true;
                    if (oc != 0x7fffffff && (ov > bv || (ov == bv && oc < bc))) {
                        bv = ov;
                        bc = oc;
                        bx = ox;
                    }
                }
                const size_t pix = (size_t)y * W + x;
                if (bc != 0x7fffffff) d.d16_out[pix] = (int16_t)(bc - x); // .cpp:219-222 / 301-302
                if (tt) a.tie_list[atomicAdd(a.tie_cnt, 1)] = (uint32_t)pix | ((uint32_t)blockIdx.z << 31);
            }
        }
        } // tile groups
    }
}
template <int R>
static size_t slide_lds_bytes() {
    constexpr int WS = 2 * R + 1, NBC = SL_COLS + SL_DC + 4, NV = SL_COLS + 2 * R + 2;
    const size_t wave = ((size_t)WS * NBC * 4 + (size_t)NBC * 8 + (size_t)NBC * 4 + (size_t)2 * NV * 4 + 15) & ~(size_t)15;
    return 4 * wave + (size_t)4 * SL_COLS * (8 + 4 + 4);
}

template <int R>
static void launch_dot4(const StageArgs &a_in, int mode, dim3 grid, hipStream_t st) {
    StageArgs a = a_in;
    if (a.ncc_mid <= 0) a.ncc_mid = R >= 5 ? 64 : (R >= 3 ? 96 : NCC_WIDE); // 11x11 / 65 candidates: 261 against 177 G pairs/s; 5x5 / 65: 372 against 444
    constexpr int WS = 2 * R + 1, SA = NCC_TX + 2 * R, SB = NCC_CH + 2 * R + NCC_G + 3;
    const size_t lds = 16 + (size_t)WS * (SA + SB) * 4 + (size_t)(NCC_CH + NCC_G) * 9 + 16;
    if (mode == 2) { // the worklist was filled by launch_set_boundary
        hipLaunchKernelGGL(k_ncc_sparse<R>, dim3(2048), dim3(256), 0, st, a);
        if (!a.opt_no_exact) hipLaunchKernelGGL(k_ncc_exact, dim3(128), dim3(256), 0, st, a, mode);
        return;
    }
    // per row: pixels with long intervals and the widest interval (decides, before the matchers run, which rows the row kernels take)
    if (a.opt_no_rowgemm != 1) hipLaunchKernelGGL(k_ncc_rowstat, dim3((grid.y + 3) / 4, 1, grid.z), dim3(256), 0, st, a, mode);
    // *a.ncc_cnt (wide-pixel count) is zero on entry: the caller hands every launch a fresh counter
    hipLaunchKernelGGL(k_ncc_dot4<R>, grid, dim3(NCC_TX), lds, st, a, mode);
    const size_t ldsw = (size_t)WS * (NCC_TX * NCC_G + 2 * R + NCC_G) * 4;
    if (ldsw > 65536) // radii 6 and 7 stage more than the default 64 KB of dynamic LDS per workgroup
        (void)hipFuncSetAttribute((const void *)k_ncc_wide<R>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsw);
    hipLaunchKernelGGL(k_ncc_wide<R>, dim3(8192), dim3(NCC_TX), ldsw, st, a, mode);
    // Rows of wide pixels: two formulations of the same scores (profiles/r04_ncc_micro.log, 12.5 MP frame, every pixel wide):
    // sliding window sums 11x11 / 129 candidates 4.9 ms (330 G pairs/s), 15x15 / 257 9.4 ms, 15x15 / 1025 33.7 ms; int8 row
    // GEMM on the matrix cores 6.2 / 10.1 / 28.8 ms -- the GEMM wins from ~500 candidates on (C2's rows below an empty parent row
    // search 2048), the sliding sums below.  k_rg_rows sorts the rows by their widest interval (opt_no_rowgemm: 0 = that choice,
    // 1 = no row kernel, 2 = every row to the GEMM, 3 = every row to the sliding sums).
    const int kind = a.opt_no_rowgemm;
    if (kind != 1) {
        hipLaunchKernelGGL(k_rg_rows, dim3(a.ndir), dim3(64), 0, st, a, kind);
        // grid.y = row SLOTS: rows with many wide pixels are few (all of them only at a wide lowest level)
        if (kind != 3)
            hipLaunchKernelGGL(k_ncc_rowgemm<R>, dim3((grid.x * NCC_TX + RG_PX - 1) / RG_PX, min((int)grid.y, RG_SLOTS), grid.z), dim3(256), 0, st, a, mode);
        if (kind != 2) {
            const size_t lds_s = slide_lds_bytes<R>();
            if (lds_s > 65536) // the attribute is per DEVICE: set on every launch, as for k_ncc_wide (contexts may live on several GPUs)
                (void)hipFuncSetAttribute((const void *)k_ncc_slide<R>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_s);
            const int tiles = (grid.x * NCC_TX + (SL_COLS - 2 * R) - 1) / (SL_COLS - 2 * R);
            hipLaunchKernelGGL(k_ncc_slide<R>, dim3((tiles + 3) / 4, min((int)grid.y, RG_SLOTS), grid.z), dim3(256), lds_s, st, a, mode);
        }
    }
    if (!a.opt_no_exact) hipLaunchKernelGGL(k_ncc_exact, dim3(128), dim3(256), 0, st, a, mode);
}

void launch_ncc_argmax(const StageArgs &a, int mode, hipStream_t st) {
    int rows = 0, cols = 0;
    for (int v = 0; v < a.ndir; v++) {
        rows = max(rows, a.d[v].own.YR - a.d[v].own.YL + 1);
        cols = max(cols, a.d[v].own.XR - a.d[v].own.XL + 1);
    }
    if (rows <= 0 || cols <= 0) return;
    dim3 grid((cols + NCC_TX - 1) / NCC_TX, rows, a.ndir);
    if (!a.opt_ncc_bytes) {
        switch (a.r) {
        case 1: return launch_dot4<1>(a, mode, grid, st);
        case 2: return launch_dot4<2>(a, mode, grid, st);
        case 3: return launch_dot4<3>(a, mode, grid, st);
        case 4: return launch_dot4<4>(a, mode, grid, st);
        case 5: return launch_dot4<5>(a, mode, grid, st);
        case 6: return launch_dot4<6>(a, mode, grid, st);
        case 7: return launch_dot4<7>(a, mode, grid, st);
        default: break;
        }
    }
    const int ws = 2 * a.r + 1;
    const int strideA = (((NCC_TX + 2 * a.r) * 3 + 3) & ~3) + 4;
    const int strideB = (((NCC_CH + 2 * a.r) * 3 + 3) & ~3) + 4;
    const size_t lds = 16 + (size_t)ws * strideA + (size_t)ws * strideB;
    if (lds > 65536) // radius 15
        (void)hipFuncSetAttribute((const void *)k_ncc_bytes, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(k_ncc_bytes, grid, dim3(NCC_TX), lds, st, a, mode, strideA, strideB);
    if (!a.opt_no_exact) hipLaunchKernelGGL(k_ncc_exact, dim3(128), dim3(256), 0, st, a, mode);
}
