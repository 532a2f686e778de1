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
// reference's WindowToVec + arma::dot (see the comment in the kernel).
// Right-window reads have no bounds check in the reference (.cpp:628): emulated on the flat
// row-major buffer, bytes outside the whole image read 0 (same rule as oracle/stereo_oracle.c).
#include "rsm_dev.h"

#include <limits.h>

// exp(-n), n = 0..127, as glibc rounds them: in sweep 1 every state is an integer, so the smoothness weights are
// exp(-k^2) and must equal the CPU libm's to keep the first (ill-conditioned) sweeps bit-identical.
__constant__ double kExpNegInt[128] = {0x1.0000000000000p+0, 0x1.78b56362cef38p-2, 0x1.152aaa3bf81ccp-3, 0x1.97db0ccceb0afp-5, 0x1.2c155b8213cf4p-6, 0x1.b993fe00d5376p-8, 0x1.44e51f113d4d6p-9, 0x1.de16b9c24a98fp-11, 0x1.5fc21041027adp-12, 0x1.02cf22526545ap-13, 0x1.7cd79b5647c9bp-15, 0x1.18354238f6764p-16, 0x1.9c54c3b43bc8bp-18, 0x1.2f6053b981d98p-19, 0x1.be6c6fdb01612p-21, 0x1.4875ca227ec38p-22, 0x1.e355bbaee85cbp-24, 0x1.639e3175a689dp-25, 0x1.05a628c699fa1p-26, 0x1.81056ff2c5772p-28, 0x1.1b48655f37267p-29, 0x1.a0db0d0ddb3ecp-31, 0x1.32b48bf117da2p-32, 0x1.c3527e433fab1p-34, 0x1.4c1078fe9228ap-35, 0x1.e8a37a45fc32ep-37, 0x1.67852a7007e42p-38, 0x1.0885298767e9ap-39, 0x1.853f01d6d53bap-41, 0x1.1e642baeb84a0p-42, 0x1.a56e0c2ac7f75p-44, 0x1.36121e24d3bbap-45, 0x1.c8464f7616468p-47, 0x1.4fb547c775da8p-48, 0x1.ee001eed62aa0p-50, 0x1.6b7719a59f0e0p-51, 0x1.0b6c3afdde064p-52, 0x1.898471fca6055p-54, 0x1.2188ad6ae3303p-55, 0x1.aa0de4bf35b38p-57, 0x1.39792499b1a24p-58, 0x1.cd480a1b74820p-60, 0x1.536452ee2f75cp-61, 0x1.f36bd37f42f3ep-63, 0x1.6f741de1748ecp-64, 0x1.0e5b73d1ff53dp-65, 0x1.8dd5e1bb09d7ep-67, 0x1.24b6031b49bdap-68, 0x1.aebabae3a41b5p-70, 0x1.3ce9b9de78f85p-71, 0x1.d257d547e083fp-73, 0x1.571db733a9d61p-74, 0x1.f8e6c24b5592ep-76, 0x1.737c5645114b5p-77, 0x1.1152eaeb73c08p-78, 0x1.923372c67a074p-80, 0x1.27ec458c65e3cp-81, 0x1.b374b315f87c1p-83, 0x1.4063f8cc8bb98p-84, 0x1.d775d87da854dp-86, 0x1.5ae191a99585ap-87, 0x1.fe7116182e9ccp-89, 0x1.778fe2497184cp-90, 0x1.1452b7723aed2p-91, 0x1.969d47321e4ccp-93, 0x1.2b2b8dd05b318p-94, 0x1.b83bf23a9a9ebp-96, 0x1.43e7fc88b8056p-97, 0x1.dca23bae16424p-99, 0x1.5eafffb34ba31p-100, 0x1.02057d1245cebp-101, 0x1.7baee1bffa80bp-103, 0x1.175af0cf60ec5p-104, 0x1.9b138170d6bfep-106, 0x1.2e73f53fba844p-107, 0x1.bd109d9d94bdap-109, 0x1.4775e0840bfddp-110, 0x1.e1dd273aa8a4ap-112, 0x1.62891f06b3450p-113, 0x1.04da4d1452919p-114, 0x1.7fd974d372e45p-116, 0x1.1a6baeadb4fd1p-117, 0x1.9f96445648b9fp-119, 0x1.31c5957a47de2p-120, 0x1.c1f2daf3b6a46p-122, 0x1.4b0dc07cabf98p-123, 0x1.e726c3f64d0fep-125, 0x1.666d0dad2961dp-126, 0x1.07b7112bc1ffep-127, 0x1.840fbc08fdc8ap-129, 0x1.1d8508fa8246ap-130, 0x1.a425b317eeacdp-132, 0x1.35208867c2683p-133, 0x1.c6e2d05bbc000p-135, 0x1.4eafb87eab0f2p-136, 0x1.ec7f3b269efa8p-138, 0x1.6a5bea046b42ep-139, 0x1.0a9bdfb02d240p-140, 0x1.8851d84118908p-142, 0x1.20a717e64a9bdp-143, 0x1.a8c1f14e2af5dp-145, 0x1.3884e838aea68p-146, 0x1.cbe0a45f75eb1p-148, 0x1.525be4e4e601dp-149, 0x1.f1e6b68529e33p-151, 0x1.6e55d2bf838a7p-152, 0x1.0d88cf37f00ddp-153, 0x1.8c9feab89b876p-155, 0x1.23d1f3e5834a0p-156, 0x1.ad6b22f55db42p-158, 0x1.3bf2cf6722e46p-159, 0x1.d0ec7df4f7bd4p-161, 0x1.56126259e093cp-162, 0x1.f75d6040aeff6p-164, 0x1.725ae6e7b9d35p-165, 0x1.107df698da211p-166, 0x1.90fa1509bd50dp-168, 0x1.2705b5b153fb8p-169, 0x1.b2216c6efdac1p-171, 0x1.3f6a58b795de3p-172, 0x1.d606847fc727ap-174, 0x1.59d34dd8a5473p-175, 0x1.fce362fe6e7d0p-177, 0x1.766b45dd84f18p-178, 0x1.137b6ce8e052cp-179, 0x1.9560792d19314p-181, 0x1.2a42764857b19p-182, 0x1.b6e4f282b43f4p-184};

__device__ __forceinline__ double exp_neg(double t) { // exp(-t), t >= 0
    if (t < 128.0 && t == (double)(int)t) return kExpNegInt[(int)t];
    return exp(-t);
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
        d.rf_key[i] = RF_NOKEY;
        d.rf_key[i + a.rf_stride] = RF_NOKEY;
    }
}

void launch_refine_init(const StageArgs &a, hipStream_t st) {
    const size_t n = (size_t)a.W * a.H;
    size_t blocks = (n + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(k_refine_init, dim3((unsigned)blocks, 1, a.ndir), dim3(256), 0, st, a);
}

// The update of .cpp:652-672 given the data term (pwp, delta = pdp - dCenter).
__device__ __forceinline__ double refine_update(int mode, double dC, double dE, double dW, double dN, double dS,
                                                double pwp, double delta, double ws) {
    // pwp == 0 only happens for index 1 (.cpp:642-643: pdp = 0)
    const double pdp = (pwp == 0) ? 0.0 : dC + delta;
    if (mode == 1) return (pdp * pwp + ws * (dE + dW) / 2) / (pwp + ws); // .cpp:658
    if (mode == 2) return (pdp * pwp + ws * (dN + dS) / 2) / (pwp + ws); // .cpp:661
    const double ex = fabs(dE - dC) - fabs(dW - dC);
    const double ey = fabs(dS - dC) - fabs(dN - dC);
    const double wx = exp_neg(ex * ex); // .cpp:665-666
    const double wy = exp_neg(ey * ey);
    double ds;
    if (wx + wy == 0) ds = (dE + dW + dS + dN) / 4;
    else ds = (wx * (dE + dW) + wy * (dN + dS)) / (2 * (wx + wy));
    return (pdp * pwp + ws * ds) / (pwp + ws); // .cpp:671
}

__device__ __forceinline__ double byte_f64(uint32_t v, int b) { return (double)(float)((v >> (8 * b)) & 0xffu); }
__device__ __forceinline__ int sum4(uint32_t v, int acc) { return (int)__builtin_amdgcn_sad_u8(v, 0u, (uint32_t)acc); }

// Data term (pwp, delta = pdp - dCenter) of pixel (x, y) for iMatch = key: .cpp:624-650.
// Bit-faithful fp64 restatement of CManageData::WindowToVec (CManageData.cpp:81-90) + arma::dot on the
// 27-element windows, same gather order (byte column outer, row inner) and the same two-accumulator sums as
// Armadillo (op_dot_meat.hpp:20-55, fn_norm.hpp:99-130): the sweep is ill-conditioned at int(d - 1.5)
// boundaries, so xi must match the reference to the last bit.
// Reads the BGRX copies (one aligned dword per pixel, X = 0): 24 loads, the window bytes stay packed in
// registers.  The reference's unchecked right-window reads (.cpp:628) are emulated on the flat buffer: the flat
// byte index test fi in [0, 3WH) of the BGR image is the flat pixel index test in [0, WH) here.
__device__ __forceinline__ void refine_data_term_packed(const uint32_t *__restrict__ A, const uint32_t *__restrict__ B,
                                                        int W, int H, int x, int y, int key, double &pwp, double &delta) {
    const long long npx = (long long)W * H;
    uint32_t aP[3][3], bP[3][5];
#pragma unroll
    for (int j = 0; j < 3; j++) {
        const uint32_t *pa = A + (size_t)(y - 1 + j) * W + (x - 1);
        const long long fb = (long long)(y - 1 + j) * W + key;
#pragma unroll
        for (int p = 0; p < 3; p++) aP[j][p] = pa[p];
#pragma unroll
        for (int q = 0; q < 5; q++) {
            const long long fi = fb + q;
            bP[j][q] = (fi >= 0 && fi < npx) ? B[fi] : 0u;
        }
    }
    int SL = 0;
#pragma unroll
    for (int j = 0; j < 3; j++)
#pragma unroll
        for (int p = 0; p < 3; p++) SL = sum4(aP[j][p], SL);
    const double meanL = (double)SL / 27.0;
    double n1 = 0.0, n2 = 0.0;
#pragma unroll
    for (int k = 0; k < 27; k++) {
        const double u = byte_f64(aP[k % 3][(k / 3) / 3], (k / 3) % 3) - meanL;
        if (k & 1) n2 += u * u;
        else n1 += u * u;
    }
    double normL = sqrt(n1 + n2);
    if (normL == 0) normL = 1;
    // the three shifts c = 0, 1, 2 as a rolled loop over a sliding register window (the body reads bP[.][0..2],
    // then the window moves one pixel): keeps the routine at ~50 registers.  The empty asm stops the compiler
    // from hoisting the 27 left-window differences (54 registers) out of the loop; they are recomputed instead.
    double x0 = 0.0, x1 = 0.0, x2 = 0.0;
#pragma unroll 1
    for (int c = 0; c < 3; c++) {
#pragma unroll
        for (int j = 0; j < 3; j++)
#pragma unroll
            for (int p = 0; p < 3; p++) asm volatile("" : "+v"(aP[j][p]));
        int SR = 0;
#pragma unroll
        for (int j = 0; j < 3; j++)
#pragma unroll
            for (int p = 0; p < 3; p++) SR = sum4(bP[j][p], SR);
        const double meanR = (double)SR / 27.0;
        double m1 = 0.0, m2 = 0.0, d1 = 0.0, d2 = 0.0;
#pragma unroll
        for (int k = 0; k < 27; k++) {
            const double ur = byte_f64(bP[k % 3][(k / 3) / 3], (k / 3) % 3) - meanR;
            const double ul = byte_f64(aP[k % 3][(k / 3) / 3], (k / 3) % 3) - meanL;
            if (k & 1) {
                m2 += ur * ur;
                d2 += ul * ur;
            } else {
                m1 += ur * ur;
                d1 += ul * ur;
            }
        }
        double normR = sqrt(m1 + m2);
        if (normR == 0) normR = 1;
        x0 = x1;
        x1 = x2;
        x2 = (1 - (d1 + d2) / (normL * normR)) / 2; // .cpp:629
#pragma unroll
        for (int j = 0; j < 3; j++)
#pragma unroll
            for (int q = 0; q < 4; q++) bP[j][q] = bP[j][q + 1];
    }
    int index = x0 >= x1; // .cpp:631-632
    if ((index ? x1 : x0) > x2) index = 2;
    if (index == 0) {
        pwp = x1 - x0;
        delta = -0.5;
    } else if (index == 2) {
        pwp = x1 - x2;
        delta = 0.5;
    } else {
        pwp = 0.5 * (x0 + x2) - x1;
        delta = (pwp == 0) ? 0.0 : 0.5 * (x0 - x2) / (x0 + x2 - 2 * x1);
    }
}

// One quad (4 adjacent lanes) computes refine_data_term_packed for one (x, y, key): every lane restates the left
// window's mean and norm, lane q >= 1 the right window of shift c = q - 1 (lane 0 shadows c = 0), lane 0
// combines.  Each value is produced by the same operation sequence as in refine_data_term_packed.
__device__ __forceinline__ void refine_data_term_quad(const uint32_t *__restrict__ A, const uint32_t *__restrict__ B,
                                                      int W, int H, int x, int y, int key, int q, double &pwp,
                                                      double &delta) {
    const long long npx = (long long)W * H;
    const int c = q > 0 ? q - 1 : 0;
    uint32_t aP[3][3], bP[3][3];
#pragma unroll
    for (int j = 0; j < 3; j++) {
        const uint32_t *pa = A + (size_t)(y - 1 + j) * W + (x - 1);
        const long long fb = (long long)(y - 1 + j) * W + key + c;
#pragma unroll
        for (int p = 0; p < 3; p++) {
            aP[j][p] = pa[p];
            const long long fi = fb + p;
            bP[j][p] = (fi >= 0 && fi < npx) ? B[fi] : 0u;
        }
    }
    int SL = 0, SR = 0;
#pragma unroll
    for (int j = 0; j < 3; j++)
#pragma unroll
        for (int p = 0; p < 3; p++) {
            SL = sum4(aP[j][p], SL);
            SR = sum4(bP[j][p], SR);
        }
    const double meanL = (double)SL / 27.0, meanR = (double)SR / 27.0;
    double n1 = 0.0, n2 = 0.0, m1 = 0.0, m2 = 0.0, d1 = 0.0, d2 = 0.0;
#pragma unroll
    for (int k = 0; k < 27; k++) {
        const double ul = byte_f64(aP[k % 3][(k / 3) / 3], (k / 3) % 3) - meanL;
        const double ur = byte_f64(bP[k % 3][(k / 3) / 3], (k / 3) % 3) - meanR;
        if (k & 1) {
            n2 += ul * ul;
            m2 += ur * ur;
            d2 += ul * ur;
        } else {
            n1 += ul * ul;
            m1 += ur * ur;
            d1 += ul * ur;
        }
    }
    double normL = sqrt(n1 + n2), normR = sqrt(m1 + m2);
    if (normL == 0) normL = 1;
    if (normR == 0) normR = 1;
    const double xi = (1 - (d1 + d2) / (normL * normR)) / 2; // .cpp:629
    const int qb = (int)(threadIdx.x & 63) & ~3; // the quad's first lane
    const double x0 = __shfl(xi, qb + 1), x1 = __shfl(xi, qb + 2), x2 = __shfl(xi, qb + 3);
    int index = x0 >= x1; // .cpp:631-632
    if ((index ? x1 : x0) > x2) index = 2;
    if (index == 0) {
        pwp = x1 - x0;
        delta = -0.5;
    } else if (index == 2) {
        pwp = x1 - x2;
        delta = 0.5;
    } else {
        pwp = 0.5 * (x0 + x2) - x1;
        delta = (pwp == 0) ? 0.0 : 0.5 * (x0 - x2) / (x0 + x2 - 2 * x1);
    }
}

// First sweep of a level: every cache entry is empty, so instead of a worklist the kernel walks the whole
// interior (and also does the mode 0 copy-through).
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_refine_first(StageArgs a) {
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
    double pwp, delta;
    refine_data_term_packed(d.img4_own, d.img4_oth, W, H, x, y, key, pwp, delta);
    const size_t cpix = pix + (size_t)((key - x) & 1) * a.rf_stride;
    d.rf_key[cpix] = (int16_t)(key - x);
    d.rf_pwp[cpix] = pwp;
    d.rf_delta[cpix] = delta;
    d.f64_b[pix] = refine_update(mode, dC, dE, dW, dN, dS, pwp, delta, a.ws);
}

// One Jacobi sweep f64_a -> f64_b: RF_PPT vertically adjacent pixels per thread, every load issued up front.
// The data term (pwp, delta) of a pixel is cached, two entries per pixel indexed by the parity of iMatch - x: the
// iteration settles into flipping between two ADJACENT iMatch values, so both stay resident (C2's top level: 40 % of the
// pixels miss in sweep 2, 2 % in sweep 10, 0.03 % in sweep 40, a few hundred pixels per sweep at the end).  A pixel whose entry belongs to
// another iMatch is a miss.  Misses are served inside the workgroup: compacted through an LDS list and computed
// four lanes per entry (a lane each when the list is long); computing them in place would run the data-term
// routine in every second wave for one or two lanes.
// TOP only gives the top level's launches (the dominant kernel) their own name in rocprof traces.
#define RFU_CAP (256 * RF_PPT) // every pixel of the workgroup may miss
template <int TOP>
__global__ __launch_bounds__(256) void k_refine_sweep(StageArgs a) {
    __shared__ uint32_t s_list[RFU_CAP]; // slot of the owner (thread * RF_PPT + i) | (iMatch - x) << 16
    __shared__ double s_res[RFU_CAP][2];
    __shared__ int s_n;
    const DirArgs &d = a.d[blockIdx.z];
    const int W = a.W, H = a.H;
    const int x = d.own.XL + 1 + blockIdx.x * 256 + (int)threadIdx.x;
    const int y0 = d.own.YL + 1 + blockIdx.y * RF_PPT;
    const int ylast = d.own.YR - 1;
    if (y0 > ylast) return; // uniform
    const bool colok = x <= d.own.XR - 1;
    const int xs = colok ? x : d.own.XL + 1; // out-of-range lanes shadow a valid column (no stores)
    const double *__restrict__ in = d.f64_a;
    double *__restrict__ out = d.f64_b;
    if (threadIdx.x == 0) s_n = 0;
    double col[RF_PPT + 2], dE[RF_PPT], dW[RF_PPT];
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
    }
    int rel[RF_PPT], crel[RF_PPT];
    double pwp[RF_PPT], delta[RF_PPT];
#pragma unroll
    for (int i = 0; i < RF_PPT; i++) {
        const int yy = min(y0 + i, ylast);
        rel[i] = (int)(col[i + 1] - 1.5); // .cpp:625 (iMatch - x)
        const size_t cpix = (size_t)yy * W + xs + (size_t)(rel[i] & 1) * a.rf_stride;
        crel[i] = d.rf_key[cpix];
        pwp[i] = d.rf_pwp[cpix];
        delta[i] = d.rf_delta[cpix];
    }
    __syncthreads(); // s_n = 0
    const int lane = threadIdx.x & 63;
    unsigned live = 0, miss = 0;
    int mode[RF_PPT];
#pragma unroll
    for (int i = 0; i < RF_PPT; i++) {
        const double dC = col[i + 1], dN = col[i], dS = col[i + 2];
        const bool lv = colok && (y0 + i <= ylast) && dC != (double)NOMATCH; // .cpp:613
        mode[i] = (int)(dE[i] != (double)NOMATCH && dW[i] != (double)NOMATCH) +
                  (int)(dS != (double)NOMATCH && dN != (double)NOMATCH) * 2; // .cpp:620
        const bool ms = lv && mode[i] != 0 && crel[i] != rel[i];
        const unsigned long long mm = __ballot(ms);
        if (mm) {
            const int leader = __builtin_ctzll(mm);
            int base = 0;
            if (lane == leader) base = atomicAdd(&s_n, __popcll(mm));
            base = __shfl(base, leader);
            if (ms) s_list[base + __popcll(mm & ((1ull << lane) - 1ull))] = (threadIdx.x * RF_PPT + i) | ((uint32_t)(rel[i] & 0xffff) << 16);
        }
        live |= (unsigned)lv << i;
        miss |= (unsigned)ms << i;
    }
    __syncthreads();
    const int n = s_n;
    if (n) { // uniform
        for (int done = 0; done < n;) { // uniform
            if (n - done > 64) { // a lane per entry: one round serves up to 256 with 1/3 of the quads' instructions
                const int e = done + (int)threadIdx.x;
                if (e < n) {
                    const uint32_t en = s_list[e];
                    const int slot = en & 0xffff, r = (int)(int16_t)(en >> 16);
                    const int ex = d.own.XL + 1 + blockIdx.x * 256 + slot / RF_PPT, ey = y0 + slot % RF_PPT;
                    double p, q;
                    refine_data_term_packed(d.img4_own, d.img4_oth, W, H, ex, ey, r + ex, p, q);
                    s_res[slot][0] = p;
                    s_res[slot][1] = q;
                }
                done += 256;
            } else { // four lanes per entry: a third of the latency, which is what a handful of misses costs
                const int e = done + ((int)threadIdx.x >> 2);
                if (done + (((int)threadIdx.x >> 6) << 4) < n) { // wave-uniform: this wave's 16 quads reach the list
                    const bool ok = e < n;
                    const uint32_t en = s_list[ok ? e : done];
                    const int slot = en & 0xffff, r = (int)(int16_t)(en >> 16);
                    const int ex = d.own.XL + 1 + blockIdx.x * 256 + slot / RF_PPT, ey = y0 + slot % RF_PPT;
                    double p, q;
                    refine_data_term_quad(d.img4_own, d.img4_oth, W, H, ex, ey, r + ex, threadIdx.x & 3, p, q);
                    if (ok && (threadIdx.x & 3) == 0) {
                        s_res[slot][0] = p;
                        s_res[slot][1] = q;
                    }
                }
                done += 64;
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < RF_PPT; i++) {
        if (!((live >> i) & 1u)) continue;
        const size_t pix = (size_t)(y0 + i) * W + x;
        if ((miss >> i) & 1u) {
            pwp[i] = s_res[threadIdx.x * RF_PPT + i][0];
            delta[i] = s_res[threadIdx.x * RF_PPT + i][1];
            const size_t cpix = pix + (size_t)(rel[i] & 1) * a.rf_stride;
            d.rf_key[cpix] = (int16_t)rel[i];
            d.rf_pwp[cpix] = pwp[i];
            d.rf_delta[cpix] = delta[i];
        }
        out[pix] = (mode[i] == 0) ? col[i + 1] /* .cpp:655 */
                                  : refine_update(mode[i], col[i + 1], dE[i], dW[i], col[i], col[i + 2], pwp[i], delta[i], a.ws);
    }
}

// One sweep f64_a -> f64_b (a.flag2 = sweep index, a.flag = top level); ev0 / ev1 (optional) bracket the launch.
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
