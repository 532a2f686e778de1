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

#include <stdlib.h>

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

// ---------------------------------------------------------------- NCC interval argmax, v_dot4_u32_u8
// Same decomposition (one row x 256 pixels per workgroup, both views' rows staged in LDS), but:
//  * the pixel's own (2R+1) x 3(2R+1)-byte window lives in registers as dwords aligned to the window
//    start (v_alignbyte_b32 once per pixel), bytes past the window masked to 0;
//  * per candidate and window row the other view's bytes come from LDS as NA+1 aligned dwords,
//    re-aligned with v_alignbyte_b32 and accumulated 4 MACs at a time with v_dot4_u32_u8;
//  * mask / S1 / S2 of the candidate columns are staged in LDS next to the rows.
template <int R>
__global__ __launch_bounds__(NCC_TX) void k_ncc_dot4(StageArgs a, int mode, int strideA, int strideB) {
    constexpr int WS = 2 * R + 1, WB = 3 * WS, NA = (WB + 3) / 4, LASTV = WB - 4 * (NA - 1);
    constexpr uint32_t LASTMASK = (LASTV == 4) ? 0xffffffffu : ((1u << (8 * LASTV)) - 1u);
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    int *s_lohi = (int *)smem;
    uint32_t *sA = (uint32_t *)(smem + 16);
    uint32_t *sB = sA + (size_t)WS * (strideA >> 2);
    int32_t *sS1 = (int32_t *)(sB + (size_t)WS * (strideB >> 2));
    int32_t *sS2 = sS1 + NCC_CH;
    uint8_t *sM = (uint8_t *)(sS2 + NCC_CH);
    const DirArgs &d = a.d[blockIdx.z];
    const int W = a.W;
    constexpr int n = WS * WS * 3;
    const int y = d.own.YL + blockIdx.y;
    const int x0 = d.own.XL + blockIdx.x * NCC_TX;
    if (y > d.own.YR || x0 > d.own.XR) return;
    const int x = x0 + threadIdx.x;
    const size_t pix = (size_t)y * W + x;

    bool active = (x <= d.own.XR) && (d.mask_own[pix] == 255);
    if (mode == 2 && active) active = (d.d16_in[pix] == NOMATCH);
    int L = 0x7fffffff, Rr = -1;
    if (active) {
        if (mode == 0) {
            L = d.oth.XL;
            Rr = d.oth.XR;
        } else {
            L = d.BL[pix];
            Rr = d.BR[pix];
        }
        L = max(L, R);
        Rr = min(Rr, W - 1 - R);
        if (L > Rr) active = false;
    }
    if (threadIdx.x == 0) {
        s_lohi[0] = 0x7fffffff;
        s_lohi[1] = -1;
    }
    __syncthreads();
    {
        int lo = active ? L : 0x7fffffff, hi = active ? Rr : -1;
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
    if (cmax < cmin) return;

    const int rowBytes = W * 3;
    // ---- stage own-view rows: LDS byte i of row j <-> image byte column baseA + i
    const int baseA = (3 * (x0 - R)) & ~3;
    {
        const int nd = strideA >> 2;
        for (int j = 0; j < WS; j++) {
            const uint8_t *src = d.img_own + (size_t)(y - R + j) * rowBytes;
            for (int i = threadIdx.x; i < nd; i += NCC_TX) {
                const int b = baseA + 4 * i;
                uint32_t v = 0;
                if (b + 3 < rowBytes) __builtin_memcpy(&v, src + b, 4);
                else
                    for (int k = 0; k < 4; k++)
                        if (b + k < rowBytes) v |= (uint32_t)src[b + k] << (8 * k);
                sA[j * nd + i] = v;
            }
        }
    }
    __syncthreads();
    // ---- own window -> registers
    uint32_t Areg[WS][NA];
    int Sa = 0, Saa = 0;
    long long va = 0;
    if (active) {
        const int oa = 3 * (x - R) - baseA;
        const int wa = oa >> 2, sh = oa & 3;
        const int nd = strideA >> 2;
#pragma unroll
        for (int j = 0; j < WS; j++) {
            uint32_t raw[NA + 1];
#pragma unroll
            for (int m = 0; m <= NA; m++) raw[m] = sA[j * nd + wa + m];
#pragma unroll
            for (int m = 0; m < NA; m++) Areg[j][m] = __builtin_amdgcn_alignbyte(raw[m + 1], raw[m], sh);
            Areg[j][NA - 1] &= LASTMASK;
        }
        Sa = d.S1_own[pix];
        Saa = d.S2_own[pix];
        va = (long long)n * Saa - (long long)Sa * Sa;
    } else {
#pragma unroll
        for (int j = 0; j < WS; j++)
#pragma unroll
            for (int m = 0; m < NA; m++) Areg[j][m] = 0;
    }
    int best = -1;
    double bestv = -1.0;

    for (int lo = cmin; lo <= cmax; lo += NCC_CH) {
        const int hi = min(lo + NCC_CH - 1, cmax);
        const int baseB = (3 * (lo - R)) & ~3;
        const int ndB = strideB >> 2;
        __syncthreads();
        {
            const int need = ((3 * (hi + R) + 3 - baseB) + 3 + 8) >> 2; // dwords incl. slack for the NA+1-th read
            const int ndw = min(need, ndB);
            for (int j = 0; j < WS; j++) {
                const uint8_t *src = d.img_oth + (size_t)(y - R + j) * rowBytes;
                for (int i = threadIdx.x; i < ndw; i += NCC_TX) {
                    const int b = baseB + 4 * i;
                    uint32_t v = 0;
                    if (b + 3 < rowBytes) __builtin_memcpy(&v, src + b, 4);
                    else
                        for (int k = 0; k < 4; k++)
                            if (b + k < rowBytes) v |= (uint32_t)src[b + k] << (8 * k);
                    sB[j * ndB + i] = v;
                }
            }
            for (int i = threadIdx.x; i <= hi - lo; i += NCC_TX) {
                const size_t o = (size_t)y * W + lo + i;
                sM[i] = d.mask_oth[o];
                sS1[i] = d.S1_oth[o];
                sS2[i] = d.S2_oth[o];
            }
        }
        __syncthreads();
        if (active) {
            const int c0 = max(L, lo), c1 = min(Rr, hi);
            for (int c = c0; c <= c1; c++) {
                if (sM[c - lo] != 255) continue;
                const int ob = 3 * (c - R) - baseB;
                const int wb = ob >> 2, shb = ob & 3;
                uint32_t Sab = 0;
#pragma unroll
                for (int j = 0; j < WS; j++) {
                    uint32_t raw[NA + 1];
#pragma unroll
                    for (int m = 0; m <= NA; m++) raw[m] = sB[j * ndB + wb + m];
#pragma unroll
                    for (int m = 0; m < NA; m++)
                        Sab = __builtin_amdgcn_udot4(Areg[j][m], __builtin_amdgcn_alignbyte(raw[m + 1], raw[m], shb), Sab, false);
                }
                const int Sb = sS1[c - lo];
                const int Sbb = sS2[c - lo];
                const long long vb = (long long)n * Sbb - (long long)Sb * Sb;
                const long long num = (long long)n * (long long)Sab - (long long)Sa * Sb;
                double score = 0.0;
                if (va > 0 && vb > 0) score = (double)num / sqrt((double)va * (double)vb);
                if (score > bestv) {
                    bestv = score;
                    best = c;
                }
            }
        }
    }
    if (active && best != -1) d.d16_out[pix] = (int16_t)(best - x);
}

template <int R>
static void launch_dot4(const StageArgs &a, int mode, dim3 grid, hipStream_t st) {
    constexpr int WS = 2 * R + 1;
    const int strideA = (((NCC_TX + 2 * R) * 3 + 3 + 12) + 3) & ~3;
    const int strideB = (((NCC_CH + 2 * R) * 3 + 3 + 12) + 3) & ~3;
    const size_t lds = 16 + (size_t)WS * strideA + (size_t)WS * strideB + (size_t)NCC_CH * 9 + 16;
    hipLaunchKernelGGL(k_ncc_dot4<R>, grid, dim3(NCC_TX), lds, st, a, mode, strideA, strideB);
}

void launch_ncc_argmax(const StageArgs &a, int mode, hipStream_t st) {
    int rows = 0, cols = 0;
    for (int v = 0; v < a.ndir; v++) {
        rows = max(rows, a.d[v].own.YR - a.d[v].own.YL + 1);
        cols = max(cols, a.d[v].own.XR - a.d[v].own.XL + 1);
    }
    if (rows <= 0 || cols <= 0) return;
    dim3 grid((cols + NCC_TX - 1) / NCC_TX, rows, a.ndir);
    static const bool force_bytes = getenv("RSM_NCC_BYTES") != nullptr; // A/B switch for validation
    if (!force_bytes) {
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
    hipLaunchKernelGGL(k_ncc_bytes, grid, dim3(NCC_TX), lds, st, a, mode, strideA, strideB);
}
