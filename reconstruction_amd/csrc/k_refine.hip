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
// NCC itself is exact integer: ncc = (27*Sab - Sa*Sb) / sqrt((27*Saa - Sa^2)(27*Sbb - Sb^2)),
// zero variance -> 0 (the reference's normu == 0 ? 1 path, CManageData.cpp:88-89).
// Right-window reads have no bounds check in the reference (.cpp:628): emulated on the flat
// row-major buffer, bytes outside the whole image read 0 (same rule as oracle/stereo_oracle.c).
#include "rsm_dev.h"

#include <limits.h>

__global__ void k_refine_init(StageArgs a) {
    const DirArgs &d = a.d[blockIdx.z];
    const size_t n = (size_t)a.W * a.H;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        const double v = (double)d.d16_in[i]; // convertTo CV_64F, .cpp:585
        d.f64_a[i] = v;
        d.f64_b[i] = v; // copyTo, .cpp:587
        d.rf_key[i] = INT_MIN;
    }
}

void launch_refine_init(const StageArgs &a, hipStream_t st) {
    const size_t n = (size_t)a.W * a.H;
    size_t blocks = (n + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(k_refine_init, dim3((unsigned)blocks, 1, a.ndir), dim3(256), 0, st, a);
}

// TOP only gives the top level's launches (the dominant kernel) their own name in rocprof traces.
template <int TOP>
__global__ __launch_bounds__(256) void k_refine_sweep(StageArgs a) {
    const DirArgs &d = a.d[blockIdx.z];
    const int x = d.own.XL + 1 + blockIdx.x * blockDim.x + threadIdx.x;
    const int y = d.own.YL + 1 + blockIdx.y;
    if (x > d.own.XR - 1 || y > d.own.YR - 1) return;
    const int W = a.W, H = a.H;
    const double *in = d.f64_a;
    double *out = d.f64_b;
    const size_t pix = (size_t)y * W + x;
    const double dC = in[pix];
    if (dC == (double)NOMATCH) return; // .cpp:613
    const double dE = in[pix + 1], dW = in[pix - 1], dN = in[pix - W], dS = in[pix + W];
    const int mode = (int)(dE != (double)NOMATCH && dW != (double)NOMATCH) +
                     (int)(dS != (double)NOMATCH && dN != (double)NOMATCH) * 2; // .cpp:620
    if (mode == 0) {
        out[pix] = dC; // .cpp:655
        return;
    }
    const int key = (int)(dC - 1.5) + x; // .cpp:625
    double pwp, delta;
    if (d.rf_key[pix] == key) {
        pwp = d.rf_pwp[pix];
        delta = d.rf_delta[pix];
    } else {
        const uint8_t *A = d.img_own, *B = d.img_oth;
        const long long total = (long long)W * H * 3;
        const int rowB = W * 3;
        int Sa = 0, Saa = 0;
        int Sb[3] = {0, 0, 0}, Sbb[3] = {0, 0, 0}, Sab[3] = {0, 0, 0};
        for (int j = 0; j < 3; j++) {
            const uint8_t *pa = A + (size_t)(y - 1 + j) * rowB + (size_t)(x - 1) * 3;
            const long long bbase = (long long)(y - 1 + j) * rowB + (long long)key * 3;
            int bb[15];
#pragma unroll
            for (int i = 0; i < 15; i++) {
                const long long fi = bbase + i;
                bb[i] = (fi >= 0 && fi < total) ? (int)B[fi] : 0;
            }
#pragma unroll
            for (int i = 0; i < 9; i++) {
                const int av = pa[i];
                Sa += av;
                Saa += av * av;
#pragma unroll
                for (int c = 0; c < 3; c++) Sab[c] += av * bb[i + 3 * c];
            }
#pragma unroll
            for (int c = 0; c < 3; c++)
#pragma unroll
                for (int i = 0; i < 9; i++) {
                    const int bv = bb[i + 3 * c];
                    Sb[c] += bv;
                    Sbb[c] += bv * bv;
                }
        }
        const int va = 27 * Saa - Sa * Sa;
        double xi[3];
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const int vb = 27 * Sbb[c] - Sb[c] * Sb[c];
            const int num = 27 * Sab[c] - Sa * Sb[c];
            double ncc = 0.0;
            if (va > 0 && vb > 0) ncc = (double)num / sqrt((double)va * (double)vb);
            xi[c] = (1 - ncc) / 2; // .cpp:629
        }
        int index = xi[0] >= xi[1]; // .cpp:631-632
        if (xi[index] > xi[2]) index = 2;
        if (index == 0) {
            pwp = xi[1] - xi[0];
            delta = -0.5;
        } else if (index == 2) {
            pwp = xi[1] - xi[2];
            delta = 0.5;
        } else {
            pwp = 0.5 * (xi[0] + xi[2]) - xi[1];
            delta = (pwp == 0) ? 0.0 : 0.5 * (xi[0] - xi[2]) / (xi[0] + xi[2] - 2 * xi[1]);
        }
        d.rf_key[pix] = key;
        d.rf_pwp[pix] = pwp;
        d.rf_delta[pix] = delta;
    }
    // pwp == 0 only happens for index 1 (.cpp:642-643: pdp = 0)
    const double pdp = (pwp == 0) ? 0.0 : dC + delta;
    const double ws = a.ws;
    double res;
    if (mode == 1) {
        res = (pdp * pwp + ws * (dE + dW) / 2) / (pwp + ws); // .cpp:658
    } else if (mode == 2) {
        res = (pdp * pwp + ws * (dN + dS) / 2) / (pwp + ws); // .cpp:661
    } else {
        const double ex = fabs(dE - dC) - fabs(dW - dC);
        const double ey = fabs(dS - dC) - fabs(dN - dC);
        const double wx = exp(-(ex * ex)); // .cpp:665-666
        const double wy = exp(-(ey * ey));
        double ds;
        if (wx + wy == 0) ds = (dE + dW + dS + dN) / 4;
        else ds = (wx * (dE + dW) + wy * (dN + dS)) / (2 * (wx + wy));
        res = (pdp * pwp + ws * ds) / (pwp + ws); // .cpp:671
    }
    out[pix] = res;
}

void launch_refine_sweep(const StageArgs &a, hipStream_t st) {
    int rows = 0, cols = 0;
    for (int v = 0; v < a.ndir; v++) {
        rows = max(rows, a.d[v].own.YR - a.d[v].own.YL - 1);
        cols = max(cols, a.d[v].own.XR - a.d[v].own.XL - 1);
    }
    if (rows <= 0 || cols <= 0) return;
    const dim3 grid((cols + 255) / 256, rows, a.ndir);
    if (a.flag) hipLaunchKernelGGL(k_refine_sweep<1>, grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL(k_refine_sweep<0>, grid, dim3(256), 0, st, a);
}
