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
#ifndef RF_TU
#define RF_TU 0 // which part of this file a translation unit compiles: 0 everything but the two time-skewed kernels, 1 k_refine_skew1, 2 k_refine_skew
#endif

#include <limits.h>
#include <type_traits>

// exp(-t), t >= 0, for the smoothness weights (.cpp:665-666).  The reference calls its C runtime's exp, whose last bit is not
// specified, and the sweep amplifies last-bit differences chaotically (tests/test_oracle_exp_control.py).  So the weights come
// from ONE fully specified evaluation, restated identically in the CPU oracle (oracle/stereo_oracle.c: orc_exp_neg) -- since
// round 5 the exp of a real C runtime: glibc 2.35's table-driven exp (sysdeps/ieee754/dbl-64/e_exp.c, EXP_TABLE_BITS 7,
// EXP_POLY_ORDER 5; <= 0.509 ulp) with exactly the operations of its FMA build (__exp_fma):
//     kd = fma(x, 128/ln2, 0x1.8p52);  ki = bits(kd);  kd -= 0x1.8p52           x = -t = (128 e + j) ln2/128 + r
//     r  = fma(kd, -ln2lo/128, fma(kd, -ln2hi/128, x))                           |r| <= ln2/256
//     tmp = fma(r2*r2, fma(r, C5, C4), fma(fma(r, C3, C2), r2, r + T[j]))        r2 = r*r
//     exp = fma(s, tmp, s),  s = 2^e H[j]  (table word + (ki << 45): one integer add on the high dword)
// and glibc's specialcase() for |x| in [512, 1024) (where s alone may underflow); 0 beyond.  Every operation is a correctly
// rounded IEEE-754 operation (fma included; fp64 denormals are on), so the bits are those of the oracle and of the host libm's
// exp on every glibc >= 2.28 FMA host (0 of 8.7 M arguments differ).  12 fp64 + 3 integer vector instructions and ONE 16-byte
// LDS read per call on an 8-deep dependent chain (rounds 3-4: a degree-13 Taylor Horner chain, 19 instructions 19 deep, whose
// last bit differed from glibc's in 5.9 % of the arguments).  The 2 KB table {bits(T[j]), bits(H[j]) - (j << 45)} lives in
// LDS: every refine kernel stages it first (exp_tab_stage), the gather then costs one LDS round trip beside the polynomial.
#include "exp_table.h"
#define EXP_INVLN2N 0x1.71547652b82fep+7
#define EXP_SHIFT 0x1.8p52
#define EXP_NEGLN2HIN (-0x1.62e42fefa0000p-8)
#define EXP_NEGLN2LON (-0x1.cf79abc9e3b3ap-47)
#define EXP_C2 0x1.ffffffffffdbdp-2
#define EXP_C3 0x1.555555555543cp-3
#define EXP_C4 0x1.55555cf172b91p-5
#define EXP_C5 0x1.1111167a4d017p-7
typedef const double2 *ExpTab; // the table in LDS: 128 x {tail, scale word}

// every thread of the workgroup, before anything returns; the caller's barrier (__syncthreads) makes it visible
__device__ __forceinline__ void exp_tab_stage(double2 *s_tab) {
    unsigned long long *w = (unsigned long long *)s_tab;
    for (int i = threadIdx.x; i < 256; i += blockDim.x) w[i] = RSM_EXP_TAB[i];
}

// the part before the final scaling: tmp ~ exp(r) - 1 + tail, s = 2^(ki/128)'s table word (may be below the normal range for t >= 512)
__device__ __forceinline__ void exp_core(double x, ExpTab tab, double &tmp, double &s) {
    const double kd0 = __builtin_fma(x, EXP_INVLN2N, EXP_SHIFT);
    const uint32_t ki = (uint32_t)__double2loint(kd0); // ki mod 2^32: two's complement in the low mantissa bits
    const double kd = kd0 - EXP_SHIFT;
    double r = __builtin_fma(kd, EXP_NEGLN2HIN, x);
    r = __builtin_fma(kd, EXP_NEGLN2LON, r);
    const double2 e = tab[ki & 127u];
    const double r2 = r * r;
    const double pa = __builtin_fma(r, EXP_C3, EXP_C2), lo = r + e.x, pb = __builtin_fma(r, EXP_C5, EXP_C4);
    tmp = __builtin_fma(pa, r2, lo);
    const double r4 = r2 * r2;
    tmp = __builtin_fma(r4, pb, tmp);
    s = __hiloint2double(__double2hiint(e.y) + (int)(ki << 13), __double2loint(e.y)); // + (ki << 45)
}

// t < 512 (callers guarantee it with a wave-uniform test): the result and the scale are normal
__device__ __forceinline__ double exp_neg_small(double t, ExpTab tab) {
    double tmp, s;
    exp_core(-t, tab, tmp, s);
    return __builtin_fma(s, tmp, s);
}
// Two arguments at once, both table reads issued FIRST: they need only the first fma of each chain, and their LDS round
// trip (the one latency of the routine that is not arithmetic) then runs beside the two reductions and polynomials
// instead of in front of `r + tail` (the scheduler otherwise sinks a read below the reduction it does not depend on).
__device__ __forceinline__ void exp_neg2_small(double t1, double t2, double &w1, double &w2, ExpTab tab) {
    const double ka = __builtin_fma(-t1, EXP_INVLN2N, EXP_SHIFT), kb = __builtin_fma(-t2, EXP_INVLN2N, EXP_SHIFT);
    const uint32_t kia = (uint32_t)__double2loint(ka), kib = (uint32_t)__double2loint(kb);
    const double2 ea = tab[kia & 127u], eb = tab[kib & 127u];
#ifndef RF_EXP_NO_SCHED_BARRIER
    __builtin_amdgcn_sched_barrier(0);
#endif
    const double kda = ka - EXP_SHIFT, kdb = kb - EXP_SHIFT;
    double ra = __builtin_fma(kda, EXP_NEGLN2HIN, -t1), rb = __builtin_fma(kdb, EXP_NEGLN2HIN, -t2);
    ra = __builtin_fma(kda, EXP_NEGLN2LON, ra);
    rb = __builtin_fma(kdb, EXP_NEGLN2LON, rb);
    const double ra2 = ra * ra, rb2 = rb * rb;
    const double paa = __builtin_fma(ra, EXP_C3, EXP_C2), pab = __builtin_fma(rb, EXP_C3, EXP_C2);
    const double pba = __builtin_fma(ra, EXP_C5, EXP_C4), pbb = __builtin_fma(rb, EXP_C5, EXP_C4);
    const double ra4 = ra2 * ra2, rb4 = rb2 * rb2;
    const double sa = __hiloint2double(__double2hiint(ea.y) + (int)(kia << 13), __double2loint(ea.y));
    const double sb = __hiloint2double(__double2hiint(eb.y) + (int)(kib << 13), __double2loint(eb.y));
    double ta = __builtin_fma(paa, ra2, ra + ea.x), tb = __builtin_fma(pab, rb2, rb + eb.x);
    ta = __builtin_fma(ra4, pba, ta);
    tb = __builtin_fma(rb4, pbb, tb);
    w1 = __builtin_fma(sa, ta, sa);
    w2 = __builtin_fma(sb, tb, sb);
}

// any t >= 0
__device__ __forceinline__ double exp_neg(double t, ExpTab tab) {
    double tmp, s;
    exp_core(-t, tab, tmp, s);
    double v = __builtin_fma(s, tmp, s);
    if (__builtin_expect(!(t < 512.0), 0)) { // e_exp.c: specialcase(), k < 0 (rare: |ex| or |ey| > 22.6 px)
        const double s2 = __hiloint2double(__double2hiint(s) + 0x3fe00000, __double2loint(s)); // 2^1022 s
        const double st = s2 * tmp; // (a separate multiply and add there, as glibc's build has them)
        double y = s2 + st;
        if (y < 1.0) { // the result is subnormal: re-round y as 1 + y would be, so that the final scaling rounds once
            double lo = s2 - y + st;
            const double hi = 1.0 + y;
            lo = 1.0 - hi + y + lo;
            y = (hi + lo) - 1.0;
            if (y == 0.0) y = 0.0;
        }
        v = 0x1p-1022 * y;
        if (t >= 1024.0) v = 0.0; // e_exp.c: __math_uflow
    }
    return v;
}
__device__ __forceinline__ void exp_neg2(double t1, double t2, double &w1, double &w2, ExpTab tab) {
    w1 = exp_neg(t1, tab);
    w2 = exp_neg(t2, tab);
}

#if RF_TU == 0 // (the translation units k_refine_skew.hip / k_refine_skew1.hip include this file for the device functions and ONE kernel each)
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

#endif // RF_TU == 0
// The update of .cpp:652-672 given the data term (pwp, delta = pdp - dCenter).
__device__ __forceinline__ double refine_update(int mode, double dC, double dE, double dW, double dN, double dS,
                                                double pwp, double delta, double ws, ExpTab tab) {
    // pwp == 0 only happens for index 1 (.cpp:642-643: pdp = 0)
    const double pdp = (pwp == 0) ? 0.0 : dC + delta;
    if (mode == 1) return (pdp * pwp + ws * (dE + dW) / 2) / (pwp + ws); // .cpp:658
    if (mode == 2) return (pdp * pwp + ws * (dN + dS) / 2) / (pwp + ws); // .cpp:661
    const double ex = fabs(dE - dC) - fabs(dW - dC);
    const double ey = fabs(dS - dC) - fabs(dN - dC);
    double wx, wy;
    exp_neg2(ex * ex, ey * ey, wx, wy, tab); // .cpp:665-666
    double ds;
    if (wx + wy == 0) ds = (dE + dW + dS + dN) / 4;
    else ds = (wx * (dE + dW) + wy * (dN + dS)) / (2 * (wx + wy));
    return (pdp * pwp + ws * ds) / (pwp + ws); // .cpp:671
}

// fp64 division a / b as the hardware sequence the compiler emits for it (v_div_scale x2, v_rcp, two Newton steps, quotient,
// residual, v_div_fmas, v_div_fixup) WITHOUT the operand scaling and the fix-up: 8 instead of 11 instructions, identical bits
// whenever v_div_scale leaves both operands unscaled and v_div_fixup passes the quotient through -- a and b finite and non-zero,
// b normal, |exponent(a) - exponent(b)| < 768, biased exponent(a) > 53, a / b normal (ISA: V_DIV_SCALE_F64).  The caller's guard:
// 2^-300 < |a| < 2^300 and 2^-300 < b < 2^300 (tests/test_gpu_golden.py holds it to the IEEE quotient on 4 M operand pairs).
__device__ __forceinline__ double div_unscaled(double a, double b) {
    double y = __builtin_amdgcn_rcp(b);
    double e = __builtin_fma(-b, y, 1.0);
    y = __builtin_fma(y, e, y);
    e = __builtin_fma(-b, y, 1.0);
    y = __builtin_fma(y, e, y);
    const double q = a * y;
    const double rem = __builtin_fma(-b, q, a);
    return __builtin_fma(rem, y, q);
}

#if RF_TU == 0 // (the translation units k_refine_skew.hip / k_refine_skew1.hip include this file for the device functions and ONE kernel each)
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

#endif // RF_TU == 0
// Lane masks straight from the compare (one v_cmp into a scalar pair), combined with scalar logic; rf_sel turns a mask back into
// a select / branch condition at no cost.  A ballot of a COMBINED bool costs two vector instructions (v_cndmask 0 / 1 + v_cmp).
#define RF_FNE(x, y) __builtin_amdgcn_fcmp((x), (y), 14) // unordered or not equal: C's !=
#define RF_FGT(x, y) __builtin_amdgcn_fcmp((x), (y), 2)  // ordered and greater: C's >
#define RF_FUGE(x, y) __builtin_amdgcn_fcmp((x), (y), 11) // unordered or greater-equal: C's !(x < y)
#define RF_IEQ(x, y) __builtin_amdgcn_sicmp((x), (y), 32)
#define RF_IGE(x, y) __builtin_amdgcn_sicmp((x), (y), 39) // signed >=
#define rf_sel(m) __builtin_amdgcn_inverse_ballot_w64(m)

// refine_update3 with the predicates as lane masks and both divisions unscaled (k_refine_skew, round 4): m_lv = the lanes whose
// result is kept.  The row takes the general sequence when any kept lane is outside the unscaled division's guard -- a weight
// below e^-200 (the reference's exp underflow, .cpp:667-668, is inside that case), a numerator that is zero or tiny (a zero sum of
// neighbours; pwp == 0 with ws * ds == 0: (dC + delta) * 0 = +-0 otherwise adds like .cpp:642-643's pdp = 0) -- and whenever ws is
// outside [2^-200, 2^200] (wsok: pwp is in [0, 1], so pwp + ws is in range with it).
__device__ __forceinline__ double refine_update3m(double dC, double dE, double dW, double dN, double dS, double pwp, double delta, double ws,
                                                  unsigned long long m_lv, bool wsok, ExpTab tab) {
    const double ex = fabs(dE - dC) - fabs(dW - dC);
    const double ey = fabs(dS - dC) - fabs(dN - dC);
    const double tx = ex * ex, ty = ey * ey;
    if (wsok) { // kernel-uniform
        double wx, wy;
        exp_neg2_small(tx, ty, wx, wy, tab); // .cpp:665-666; the same bits as exp_neg for t < 512 (guard below: <= 200)
        const double a1 = wx * (dE + dW) + wy * (dN + dS);
        const double ds = div_unscaled(a1, 2 * (wx + wy)); // the denominator is in [2^-287, 4]
        const double a2 = (dC + delta) * pwp + ws * ds;
        const double u = div_unscaled(a2, pwp + ws);        // .cpp:671
        const unsigned long long m_bad = RF_FGT(fmax(tx, ty), 200.0) | ~(RF_FGT(fabs(a1), 0x1p-300) & RF_FGT(fabs(a2), 0x1p-300));
        if (!(m_lv & m_bad)) return u; // wave-uniform; the usual case
    }
    const double pdp = (pwp == 0) ? 0.0 : dC + delta;
    double wx, wy;
    exp_neg2(tx, ty, wx, wy, tab); // .cpp:665-666
    const double sw = wx + wy;
    double ds = (wx * (dE + dW) + wy * (dN + dS)) / (2 * sw);
    ds = (sw == 0) ? (dE + dW + dS + dN) / 4 : ds; // .cpp:667-668
    return (pdp * pwp + ws * ds) / (pwp + ws); // .cpp:671
}

// refine_update3 with its one wave-uniform test taken from lane masks (no ballot of a combined bool) and nothing else changed:
// the general divisions, the test early in the chain (k_refine_skew variant 8).
__device__ __forceinline__ double refine_update3e(double dC, double dE, double dW, double dN, double dS, double pwp, double delta, double ws,
                                                  unsigned long long m_lv, ExpTab tab) {
    const double ex = fabs(dE - dC) - fabs(dW - dC);
    const double ey = fabs(dS - dC) - fabs(dN - dC);
    const double tx = ex * ex, ty = ey * ey;
    if (!(m_lv & (RF_FUGE(fmax(tx, ty), 512.0) | ~RF_FNE(pwp, 0.0)))) { // wave-uniform; the usual case
        double wx, wy;
        exp_neg2_small(tx, ty, wx, wy, tab); // .cpp:665-666; both weights > exp(-512) > 0
        const double ds = (wx * (dE + dW) + wy * (dN + dS)) / (2 * (wx + wy));
        return ((dC + delta) * pwp + ws * ds) / (pwp + ws); // .cpp:671
    }
    const double pdp = (pwp == 0) ? 0.0 : dC + delta;
    double wx, wy;
    exp_neg2(tx, ty, wx, wy, tab); // .cpp:665-666
    const double sw = wx + wy;
    double ds = (wx * (dE + dW) + wy * (dN + dS)) / (2 * sw);
    ds = (sw == 0) ? (dE + dW + dS + dN) / 4 : ds; // .cpp:667-668
    return (pdp * pwp + ws * ds) / (pwp + ws); // .cpp:671
}

// refine_update for mode 3 without a divergent branch (k_refine_skew's straight-line path, entered by all lanes): the same
// operations on the same operands.  The three special cases of the general form -- a weight beyond exp's underflow threshold
// (|ex| or |ey| > 27), both weights zero (.cpp:667-668), pwp == 0 (.cpp:642-643) -- are tested ONCE for the whole row (`lv`: the
// lanes whose result is kept) and handled by the general sequence when any kept lane needs it; otherwise their compares and
// selects (10 of ~135 vector instructions) are not executed at all.
__device__ __forceinline__ double refine_update3(double dC, double dE, double dW, double dN, double dS, double pwp, double delta, double ws, bool lv, ExpTab tab) {
    const double ex = fabs(dE - dC) - fabs(dW - dC);
    const double ey = fabs(dS - dC) - fabs(dN - dC);
    const double tx = ex * ex, ty = ey * ey;
    if (!__ballot(lv && (!(fmax(tx, ty) < 512.0) || pwp == 0))) { // wave-uniform; the usual case
        double wx, wy;
        exp_neg2_small(tx, ty, wx, wy, tab); // .cpp:665-666; both weights > exp(-512) > 0
        const double ds = (wx * (dE + dW) + wy * (dN + dS)) / (2 * (wx + wy));
        return ((dC + delta) * pwp + ws * ds) / (pwp + ws); // .cpp:671
    }
    const double pdp = (pwp == 0) ? 0.0 : dC + delta;
    double wx, wy;
    exp_neg2(tx, ty, wx, wy, tab); // .cpp:665-666
    const double sw = wx + wy;
    double ds = (wx * (dE + dW) + wy * (dN + dS)) / (2 * sw);
    if (__ballot(sw == 0)) // both weights underflowed somewhere in the row (|ex|, |ey| > 27): .cpp:667-668, wave-uniform and rare
        ds = (sw == 0) ? (dE + dW + dS + dN) / 4 : ds;
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
// (pwp, delta) from the three matching costs xi at iMatch + {0, 1, 2}: .cpp:631-650
__device__ __forceinline__ void refine_entry(double x0, double x1, double x2, double &pwp, double &delta) {
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

// xs (optional): the three matching costs themselves.
__device__ __forceinline__ void refine_data_term_packed(const uint32_t *__restrict__ A, const uint32_t *__restrict__ B,
                                                        int W, int H, int x, int y, int key, double &pwp, double &delta,
                                                        double *xs = nullptr) {
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
    refine_entry(x0, x1, x2, pwp, delta);
    if (xs) {
        xs[0] = x0;
        xs[1] = x1;
        xs[2] = x2;
    }
}

// k_refine_first's form of the data term: every pixel of the level computes one, the kernel is bound by its vector
// instructions (85 % VALU-busy), and it has 128 registers to spend -- so the 27 left-window differences (u - meanL) and the left
// norm are computed ONCE and kept for the three shifts and for the extra matching cost of the second cache way, where the
// 50-register routine of the sweep kernels' miss path recomputes them per shift.  Each value comes out of the same operation
// sequence (same conversions, same two-accumulator sums in the same order) as in refine_data_term_packed.
struct RfLeft {
    double ul[27]; // byte - meanL in the reference's vector order (byte column outer, row inner)
    double normL;
};
__device__ __forceinline__ void refine_left(const uint32_t *__restrict__ A, int W, int x, int y, RfLeft &L) {
    uint32_t aP[3][3];
#pragma unroll
    for (int j = 0; j < 3; j++) {
        const uint32_t *pa = A + (size_t)(y - 1 + j) * W + (x - 1);
#pragma unroll
        for (int p = 0; p < 3; p++) aP[j][p] = pa[p];
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
        L.ul[k] = u;
        if (k & 1) n2 += u * u;
        else n1 += u * u;
    }
    double normL = sqrt(n1 + n2);
    if (normL == 0) normL = 1;
    L.normL = normL;
}
// the matching cost xi = (1 - ncc) / 2 against the right window whose left edge is column `col` (.cpp:626-629)
__device__ __forceinline__ double refine_cost_left(const RfLeft &L, const uint32_t *__restrict__ B, int W, int H, int y, int col) {
    const long long npx = (long long)W * H;
    uint32_t bP[3][3];
#pragma unroll
    for (int j = 0; j < 3; j++) {
        const long long fb = (long long)(y - 1 + j) * W + col;
#pragma unroll
        for (int p = 0; p < 3; p++) {
            const long long fi = fb + p;
            bP[j][p] = (fi >= 0 && fi < npx) ? B[fi] : 0u;
        }
    }
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
        if (k & 1) {
            m2 += ur * ur;
            d2 += L.ul[k] * ur;
        } else {
            m1 += ur * ur;
            d1 += L.ul[k] * ur;
        }
    }
    double normR = sqrt(m1 + m2);
    if (normR == 0) normR = 1;
    return (1 - (d1 + d2) / (L.normL * normR)) / 2; // .cpp:629
}

// One quad (4 adjacent lanes) computes refine_data_term_packed for one (x, y, key): every lane restates the left
// window's mean and norm, lane q >= 1 the right window of shift c = q - 1 (lane 0 shadows c = 0), lane 0
// combines.  Each value is produced by the same operation sequence as in refine_data_term_packed.
__device__ __forceinline__ void refine_data_term_quad(const uint32_t *__restrict__ A, const uint32_t *__restrict__ B,
                                                      int W, int H, int x, int y, int key, int q, double &pwp,
                                                      double &delta, double *xi_own = nullptr) {
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
    if (xi_own) *xi_own = xi; // (test entry: this lane's own matching cost)
    const int qb = (int)(threadIdx.x & 63) & ~3; // the quad's first lane
    const double x0 = __shfl(xi, qb + 1), x1 = __shfl(xi, qb + 2), x2 = __shfl(xi, qb + 3);
    refine_entry(x0, x1, x2, pwp, delta);
}

#if RF_TU == 0 // (the translation units k_refine_skew.hip / k_refine_skew1.hip include this file for the device functions and ONE kernel each)
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
    const size_t cpix = pix + (size_t)((key - x) & 1) * a.rf_stride;
    d.rf_key[cpix] = (int16_t)(key - x);
    d.rf_pwp[cpix] = pwp;
    d.rf_delta[cpix] = delta;
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
        const size_t cpix2 = pix + (size_t)(rel2 & 1) * a.rf_stride;
        d.rf_key[cpix2] = (int16_t)rel2;
        d.rf_pwp[cpix2] = p2;
        d.rf_delta[cpix2] = q2;
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
// DEFER = 1: the sweep only LISTS its misses (miss_list, sized so that no shard can overflow) and leaves those pixels to
// k_refine_fixup; without the service code the kernel needs half the registers.
template <int TOP, int DEFER>
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
    const int y0 = max(d.own.YL + 1, a.row_lo) + blockIdx.y * RF_PPT; // rows [row_lo, row_hi) of the interior
    const int ylast = min(d.own.YR - 1, a.row_hi - 1);
    if (y0 > ylast) return; // uniform
    const bool colok = x <= d.own.XR - 1;
    const int xs = colok ? x : d.own.XL + 1; // out-of-range lanes shadow a valid column (no stores)
    const double *__restrict__ in = d.f64_a;
    double *__restrict__ out = d.f64_b;
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
        const size_t cpix = (size_t)yy * W + xs + (size_t)(rel[i] & 1) * a.rf_stride;
        crel[i] = d.rf_key[cpix];
        pwp[i] = d.rf_pwp[cpix];
        delta[i] = d.rf_delta[cpix];
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
    unsigned deferred = 0; // pixels whose update k_refine_fixup will write
    if (DEFER) {
        if (n) { // wave-uniform: one append per wave; consecutive workgroups go to consecutive shards, and a shard holds every
                 // pixel of its workgroups (rsm_api.hip: miss_cap), so nothing is ever dropped
            const unsigned shard = (blockIdx.x + blockIdx.y * gridDim.x + blockIdx.z * gridDim.x * gridDim.y) & (RF_UPD_SHARDS - 1);
            int base = 0;
            if (lane == 0) base = atomicAdd(a.upd_cnt + (a.flag3 & 1) * RF_UPD_SHARDS + shard, n);
            base = __shfl(base, 0);
#pragma unroll
            for (int i = 0; i < RF_PPT; i++)
                if (((miss >> i) & 1u) && base + pos[i] < a.miss_cap)
                    a.miss_list[(size_t)shard * a.miss_cap + base + pos[i]] = RfMiss{(uint32_t)((size_t)(y0 + i) * W + x) | ((uint32_t)blockIdx.z << 31), rel[i]};
            deferred = miss;
        }
    } else
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
        if (!((live >> i) & 1u) || ((deferred >> i) & 1u)) continue;
        const size_t pix = (size_t)(y0 + i) * W + x;
        if ((miss >> i) & 1u) {
            pwp[i] = s_res[wid][pos[i]][0];
            delta[i] = s_res[wid][pos[i]][1];
            const size_t cpix = pix + (size_t)(rel[i] & 1) * a.rf_stride;
            d.rf_key[cpix] = (int16_t)rel[i];
            d.rf_pwp[cpix] = pwp[i];
            d.rf_delta[cpix] = delta[i];
        }
        out[pix] = (mode[i] == 0) ? col[i + 1] /* .cpp:655 */
                                  : refine_update(mode[i], col[i + 1], dE[i], dW[i], col[i], col[i + 2], pwp[i], delta[i], a.ws, s_exp);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Two Jacobi sweeps per launch (option refine_multi_from; OFF by default -- measured slower, see below).
// A sweep moves 52 B per pixel through the fabric (state in and out, and the lines of both cache ways).  Here a
// workgroup keeps a tile of the state in LDS across two sweeps: it loads the 20 x 66 input values around its
// 16 x 62 output pixels, computes sweep t on 18 x 64 (one ring of redundant work: 1.16x), sweep t+1 on its own
// 16 x 62 from the LDS copy, and writes only that -- per two sweeps the state crosses the fabric once and the
// second sweep's cache entries mostly come from lines the first one pulled into L2.  Rows are processed in a rolled
// loop (a wave per row, values from LDS, the entry from global memory), so the data-term routine exists once.
// Cache misses are served inside the wave (list in LDS, four lanes per entry).  The ring pixels belong to
// neighbouring workgroups, which may miss on them too: a new entry is therefore never written to the cache during
// the launch (a reader could see a torn key / pwp / delta triple) but appended to an update list that
// k_refine_apply scatters afterwards.  Dropping a record (list full, or a second miss on the same way of the same
// pixel in this launch) is always safe: the cache keeps an older, still consistent entry.
// Values are those of two single sweeps, bit for bit (tests/test_gpu_parity.py).
// MEASURED (C2 top level, round 2): 232 us per launch = 116 us per sweep against 107 us for a settled single sweep.
// PMC: 93.7 M VALU wave-instructions per launch, VALU busy 156 us of chip time -- the halo, the tile bookkeeping and
// the per-row loop cost more issue slots than the halved traffic returns; variants that stage both cache ways in
// LDS up front (64 KB per workgroup, 2 workgroups per CU) or update five independent pixels per lane (248 VGPRs)
// ran at 283-310 us.  The single sweep is VALU-active 60 of its 107 us (36.8 M quad-cycles): the stage sits within
// 1.8x of its fp64 issue floor and at the fabric's rate for its traffic, so halving the traffic alone cannot halve it.
#define RM_TW 64
#define RM_TR 16
template <int TOP>
__global__ __launch_bounds__(256) void k_refine_multi(StageArgs a) {
    __shared__ double tin[RM_TR + 4][RM_TW + 2]; // rows Y0-2 .. Y0+17, columns X0-1 .. X0+64 of the input
    __shared__ double ts1[RM_TR + 2][RM_TW];     // rows Y0-1 .. Y0+16, columns X0 .. X0+63 after the first sweep
    __shared__ uint8_t tflag[RM_TR + 2][RM_TW];  // way + 1 of a record the first sweep emitted for the pixel
    __shared__ uint32_t s_list[4][64];
    __shared__ double s_res[4][16][2];
    __shared__ double2 s_exp[128]; // the specified exp's table (exp_tab_stage; the barrier below covers it)
    exp_tab_stage(s_exp);
    const DirArgs &d = a.d[blockIdx.z];
    const int W = a.W, H = a.H;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int X0 = d.own.XL + blockIdx.x * (RM_TW - 2); // lane l <-> column X0 + l; lanes 1..62 are this workgroup's
    const int Y0 = d.own.YL + 1 + blockIdx.y * RM_TR;   // own rows Y0 .. Y0+15
    const int xlo = d.own.XL + 1, xhi = d.own.XR - 1, ylo = d.own.YL + 1, yhi = d.own.YR - 1; // the interior
    if (Y0 > yhi || X0 + 1 > xhi) return; // uniform (no barrier has been passed yet)
    const double *__restrict__ in = d.f64_a;
    double *__restrict__ out = d.f64_b;
    for (int e = threadIdx.x; e < (RM_TR + 4) * (RM_TW + 2); e += 256) {
        const int i = e / (RM_TW + 2), j = e - i * (RM_TW + 2);
        const int yy = min(max(Y0 - 2 + i, 0), H - 1), xx = min(max(X0 - 1 + j, 0), W - 1);
        tin[i][j] = in[(size_t)yy * W + xx];
    }
    __syncthreads();
    const int x = X0 + lane;
    const int xc = min(x, W - 1);
    const bool xin = x >= xlo && x <= xhi;
    const bool xown = xin && lane >= 1 && lane <= RM_TW - 2;
    const unsigned shard = (blockIdx.x + blockIdx.y * gridDim.x + blockIdx.z * 7u) & (RF_UPD_SHARDS - 1);
    int32_t *cnt = a.upd_cnt + (a.flag3 & 1) * RF_UPD_SHARDS + shard;
#pragma unroll 1
    for (int p = 0; p < 2; p++) {
        const int nrows = p ? RM_TR : RM_TR + 2;
#pragma unroll 1
        for (int r = wid; r < nrows; r += 4) {
            const int y = p ? Y0 + r : Y0 - 1 + r;
            double dC, dN, dS, dE, dW;
            if (p == 0) {
                const int i = r + 1, j = lane + 1;
                dC = tin[i][j];
                dN = tin[i - 1][j];
                dS = tin[i + 1][j];
                dE = tin[i][j + 1];
                dW = tin[i][j - 1];
            } else {
                const int i = r + 1;
                dC = ts1[i][lane];
                dN = ts1[i - 1][lane];
                dS = ts1[i + 1][lane];
                dE = ts1[i][min(lane + 1, RM_TW - 1)];
                dW = ts1[i][max(lane - 1, 0)];
            }
            const bool yin = y >= ylo && y <= yhi;
            const bool live = yin && (p ? xown : xin) && dC != (double)NOMATCH; // .cpp:613
            const int mode = (int)(dE != (double)NOMATCH && dW != (double)NOMATCH) +
                             (int)(dS != (double)NOMATCH && dN != (double)NOMATCH) * 2; // .cpp:620
            const int rel = (int)(dC - 1.5); // .cpp:625 (iMatch - x)
            const size_t pix = (size_t)min(max(y, 0), H - 1) * W + xc;
            const size_t cpix = pix + (size_t)(rel & 1) * a.rf_stride;
            const int crel = d.rf_key[cpix];
            double pwp = d.rf_pwp[cpix], delta = d.rf_delta[cpix];
            const bool miss = live && mode != 0 && crel != rel;
            const unsigned long long mm = __ballot(miss);
            if (mm) { // wave-uniform
                const int n = __popcll(mm), pos = __popcll(mm & ((1ull << lane) - 1ull));
                if (miss) s_list[wid][pos] = (uint32_t)lane | ((uint32_t)(rel & 0xffff) << 16);
                __builtin_amdgcn_wave_barrier();
                for (int done = 0; done < n; done += 16) { // four lanes per entry, 16 entries per round
                    const int e = done + (lane >> 2);
                    const bool ok = e < n;
                    const uint32_t en = s_list[wid][ok ? e : done];
                    const int ex = X0 + (int)(en & 0xffff), er = (int)(int16_t)(en >> 16);
                    double pq, qq;
                    refine_data_term_quad(d.img4_own, d.img4_oth, W, H, ex, y, er + ex, lane & 3, pq, qq);
                    if (ok && (lane & 3) == 0) {
                        s_res[wid][e - done][0] = pq;
                        s_res[wid][e - done][1] = qq;
                    }
                    __builtin_amdgcn_wave_barrier();
                    if (miss && pos >= done && pos < done + 16) {
                        pwp = s_res[wid][pos - done][0];
                        delta = s_res[wid][pos - done][1];
                    }
                    __builtin_amdgcn_wave_barrier();
                }
                // the new entries of this workgroup's own pixels go to the update list
                const bool own = miss && xown && y >= Y0 && y < Y0 + RM_TR;
                const bool emit = own && (p == 0 || tflag[r + 1][lane] != (uint8_t)((rel & 1) + 1));
                const unsigned long long em = __ballot(emit);
                if (em) {
                    const int leader = __builtin_ctzll(em);
                    int base = 0;
                    if (lane == leader) base = atomicAdd(cnt, __popcll(em));
                    base = __shfl(base, leader) + __popcll(em & ((1ull << lane) - 1ull));
                    if (emit && base < a.upd_cap) {
                        RfUpd u;
                        u.pix = (uint32_t)pix | ((uint32_t)blockIdx.z << 31);
                        u.rel = rel;
                        u.pwp = pwp;
                        u.delta = delta;
                        a.upd_list[(size_t)shard * a.upd_cap + base] = u;
                    }
                }
                if (p == 0) tflag[r][lane] = emit ? (uint8_t)((rel & 1) + 1) : (uint8_t)0;
            } else if (p == 0) {
                tflag[r][lane] = 0;
            }
            const double val = !live ? dC : (mode == 0 ? dC /* .cpp:655 */ : refine_update(mode, dC, dE, dW, dN, dS, pwp, delta, a.ws, s_exp));
            if (p == 0) ts1[r][lane] = val;
            else if (live) out[pix] = val;
        }
        __syncthreads();
    }
}

// Deferred miss service (a.defer sweeps): a lane per listed pixel -- every lane of the launch computes a data term, where
// the in-sweep service runs the ~1000-instruction routine in nearly every wave for a handful of lanes (sweeps ~4..40 of a
// level: 1-10 % of the pixels miss, scattered over all waves).  Writes the cache entry and the pixel's update exactly as
// the sweep would have, and clears the other counter set for the next deferring sweep.
__global__ __launch_bounds__(256) void k_refine_fixup(StageArgs a) {
    __shared__ double2 s_exp[128]; // the specified exp's table (exp_tab_stage)
    exp_tab_stage(s_exp);
    __syncthreads();
    const int32_t *cnt = a.upd_cnt + (a.flag3 & 1) * RF_UPD_SHARDS;
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x < RF_UPD_SHARDS) a.upd_cnt[((a.flag3 + 1) & 1) * RF_UPD_SHARDS + threadIdx.x] = 0;
    const int W = a.W, H = a.H;
    for (int sh = blockIdx.y; sh < RF_UPD_SHARDS; sh += gridDim.y) {
        const int n = min(cnt[sh], a.miss_cap);
        for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
            const RfMiss u = a.miss_list[(size_t)sh * a.miss_cap + i];
            const DirArgs &d = a.d[u.pix >> 31];
            const size_t pix = (size_t)(u.pix & 0x7fffffffu);
            const int y = (int)(pix / W), x = (int)(pix - (size_t)y * W);
            const double *__restrict__ in = d.f64_a;
            const double dC = in[pix], dE = in[pix + 1], dW = in[pix - 1], dN = in[pix - W], dS = in[pix + W];
            const int mode = (int)(dE != (double)NOMATCH && dW != (double)NOMATCH) +
                             (int)(dS != (double)NOMATCH && dN != (double)NOMATCH) * 2; // .cpp:620 (never 0 for a listed pixel)
            double pwp, delta;
            refine_data_term_packed(d.img4_own, d.img4_oth, W, H, x, y, u.rel + x, pwp, delta);
            const size_t cpix = pix + (size_t)(u.rel & 1) * a.rf_stride;
            d.rf_key[cpix] = (int16_t)u.rel;
            d.rf_pwp[cpix] = pwp;
            d.rf_delta[cpix] = delta;
            d.f64_b[pix] = refine_update(mode, dC, dE, dW, dN, dS, pwp, delta, a.ws, s_exp);
        }
    }
}
void launch_refine_fixup(const StageArgs &a, hipStream_t st) {
    hipLaunchKernelGGL(k_refine_fixup, dim3(64, RF_UPD_SHARDS), dim3(256), 0, st, a);
}

// scatters the update list of the k_refine_multi launch a.flag3 into the cache and clears the other counter set
__global__ __launch_bounds__(256) void k_refine_apply(StageArgs a) {
    const int32_t *cnt = a.upd_cnt + (a.flag3 & 1) * RF_UPD_SHARDS;
    if (blockIdx.x == 0 && threadIdx.x < RF_UPD_SHARDS) a.upd_cnt[((a.flag3 + 1) & 1) * RF_UPD_SHARDS + threadIdx.x] = 0;
    for (int sh = blockIdx.y; sh < RF_UPD_SHARDS; sh += gridDim.y) {
        const int n = min(cnt[sh], a.upd_cap);
        for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
            const RfUpd u = a.upd_list[(size_t)sh * a.upd_cap + i];
            const DirArgs &d = a.d[u.pix >> 31];
            const size_t cpix = (size_t)(u.pix & 0x7fffffffu) + (size_t)(u.rel & 1) * a.rf_stride;
            d.rf_key[cpix] = (int16_t)u.rel;
            d.rf_pwp[cpix] = u.pwp;
            d.rf_delta[cpix] = u.delta;
        }
    }
}

#endif // RF_TU == 0
// k_refine_skew's miss path, entered by the whole wave when any of its lanes misses.  A handful of misses per row is the
// usual case once the iteration has settled: they are listed in LDS and computed four lanes per entry, 16 entries per
// round (a third of the instructions of a lane computing its own).  The new entries are
// installed in the LDS copy of the cache row and -- for pixels the workgroup owns, once per (pixel, way) and launch --
// appended to the update list.
__device__ __forceinline__ void skew_miss(const StageArgs &a, const DirArgs &d, int W, int H, int x, int x_lane0, int r, int rel, int way, int lane, bool miss,
                                          bool owned, int32_t *cnt, unsigned shard, double2 (*ent_rows)[64], uint32_t *key_row,
                                          unsigned long long *emit_row, uint32_t kk, uint8_t *mlist, unsigned dir) {
    const unsigned long long mm = __ballot(miss);
    const int n = __popcll(mm);
    const int rank = __popcll(mm & ((1ull << lane) - 1ull));
    if (miss) mlist[rank] = (uint8_t)lane;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    double2 pd = make_double2(0.0, 0.0);
#pragma unroll 1
    for (int done = 0; done < n; done += 16) { // 16 entries per round, four lanes each
        const int j = done + (lane >> 2);
        const int ml = mlist[j < n ? j : done]; // the lane this quad works for
        const int mrel = __shfl(rel, ml);
        const int mx = x_lane0 + ml;
        double p, q;
        refine_data_term_quad(d.img4_own, d.img4_oth, W, H, mx, r, mrel + mx, lane & 3, p, q);
        // lane ml picks up the result of the quad that served it (every lane of a quad holds it)
        const double mp = __shfl(p, (rank & 15) << 2), mq = __shfl(q, (rank & 15) << 2);
        if (miss && rank >= done && rank < done + 16) {
            pd.x = mp;
            pd.y = mq;
        }
    }
    if (miss) {
        ent_rows[way][lane] = pd;
        key_row[lane] = way ? ((kk & 0xffffu) | ((uint32_t)(rel & 0xffff) << 16)) : ((kk & 0xffff0000u) | (uint32_t)(rel & 0xffff));
    }
    const unsigned long long fl0 = emit_row[0], fl1 = emit_row[1]; // (one wave at a time works on a row slot)
    const bool emit = miss && owned && !(((way ? fl1 : fl0) >> lane) & 1ull);
    const unsigned long long em = __ballot(emit);
    if (em) {
        const int leader = __builtin_ctzll(em);
        int base = 0;
        if (lane == leader) base = atomicAdd(cnt, __popcll(em));
        base = __shfl(base, leader) + __popcll(em & ((1ull << lane) - 1ull));
        if (emit && base < a.upd_cap) {
            RfUpd u;
            u.pix = (uint32_t)((size_t)r * W + x) | ((uint32_t)dir << 31);
            u.rel = rel;
            u.pwp = pd.x;
            u.delta = pd.y;
            a.upd_list[(size_t)shard * a.upd_cap + base] = u;
        }
        const bool listed = emit && base < a.upd_cap;
        const unsigned long long n0 = __ballot(listed && !way), n1 = __ballot(listed && way);
        if (lane == 0) {
            emit_row[0] = fl0 | n0;
            emit_row[1] = fl1 | n1;
        }
    }
}

#if RF_TU == 2 // k_refine_skew.hip: the time-skewed kernel as a translation unit of its own (its scheduling strategy: Makefile)
// ---------------------------------------------------------------------------------------------------------------
// T Jacobi sweeps per launch, time-skewed down the rows (option refine_skew_from).
// A workgroup of T waves owns a strip of 64 columns and a chunk of rows and streams down it: in step s it stages row
// s + 1 of the input (the state and BOTH cache ways, 44 B per pixel, every load a full row segment) while wave t - 1
// advances sweep t on row s - 2t + 1, so a row's values cross the fabric once per T sweeps instead of once per sweep,
// and the cache ways of a pixel are fetched once however the pixel alternates between them.  The levels are two rows
// apart, so within a step no wave needs what another one writes: one workgroup barrier per step, none inside.  The
// state lives in T rings of 4 rows in LDS (E / W neighbours come from the ring), the cache entries of the 2T + 1 rows
// in flight beside them; no tile load phase.  Redundant work: T columns either side of a strip (64 / (64 - 2T)) and
// T (T - 1) row updates per chunk.  The region that sweep t can compute shrinks by one pixel per sweep from the staged
// region (a trapezoid in space-time); pixels outside it copy through and are never consumed by a valid update.  Rows
// and columns on the margin border (never updated by the reference, .cpp:592,608) copy through at every level.
// Cache misses (rare once the iteration has settled, which is when this kernel is used) are served by the lane
// itself; the new entry lives in LDS for the rest of the launch and goes to the update list that k_refine_apply
// scatters afterwards (another strip may be staging the same pixel's entries at that moment: an in-place write could
// be read torn).  At most one record per (pixel, way) and launch, so k_refine_apply never sees two writers of a slot.
// Values are those of T single sweeps, bit for bit (tests/test_gpu_parity.py).
// V (option refine_skew_variant, T = 4; the default is 28 = 4 + 8 + 16).
//   V & 4: a row without a live pixel in the strip copies through without the update math (C3's elliptic masks leave 27 % of the
//          margin's box empty: 272 -> 286 Mdisp/s; C2 unchanged).
//   V & 8: the row's wave-level predicates (miss, "every live pixel has four neighbours", the update's rare cases) as lane masks
//          straight from the compares -- a ballot of a COMBINED bool costs a v_cndmask + v_cmp pair, three times per row; all
//          tests stay EARLY in the row's chain: 99.2 M vector wave-instructions per launch against 104.9 M, 0.313 ms against
//          0.322, +2 % on C2 and C3.
//   V & 16: only the cache way the state selects is read from LDS, in a second round trip once dC is there, instead of both
//          ways and four selects: 93.2 M vector wave-instructions against 97.4 M, a quarter less LDS read traffic, +1.5 % on C2
//          and C3 (four same-box repetitions each).
// (Round 4 also took the vector address arithmetic out of the staging: scalar row pointer + 32-bit lane byte offset is the
// load's own addressing mode -- 99.2 M -> 97.4 M.)
// The other two bits hold round-4 restatements that compute the same bits with less work per wave and are SLOWER, kept as
// measured evidence of what bounds this kernel:
//   V & 1: the staging of a row shared by two waves (below);
//   V & 2: the row's predicates as lane masks and the mode-3 update with unscaled divisions (refine_update3m): 8 % fewer
//          vector instructions per launch (95.8 M against 104.1 M), 12 % more scalar ones, 0.328 ms per launch against 0.310.
template <int T, int TOP, int V>
#ifndef RF_SKEW_WPE
#define RF_SKEW_WPE 5 // waves per SIMD the register budget is set for (5: 96 VGPRs, what the 30 KB of LDS per workgroup allow; 4: 128 -- A/B)
#endif
__global__ __launch_bounds__(64 * T) __attribute__((amdgpu_waves_per_eu(RF_SKEW_WPE, RF_SKEW_WPE))) void k_refine_skew(StageArgs a) {
    constexpr int NE = 2 * T + 1;  // rows of cache entries resident: row r is staged in step r - 1 and last used in step r + 2T - 1
    constexpr int UW = 64 - 2 * T; // columns a strip owns
    __shared__ double s_d[T][4][66];       // [level][row & 3][lane + 1]
    __shared__ double2 s_ent[NE][2][64];   // [row slot][way][lane] = (pwp, delta)
    __shared__ uint32_t s_key[NE][64];     // key of way 0 | key of way 1 << 16
    __shared__ unsigned long long s_emit[NE][2]; // [row slot][way]: the lanes whose cache slot this launch already listed a new entry for
    __shared__ double2 s_exp[128];         // the specified exp's table (exp_tab_stage; 2 KB: the workgroup stays within the 32 000 B that
                                           // let five of them share a CU -- 31 632 B at T = 4)
    __shared__ uint8_t s_ml[T][64];        // per wave: the lanes of a row's misses, for the four-lanes-per-entry service
#ifdef RF_SKEW_PAD // timing experiment: what the LDS of 24-byte cache entries (a cached reciprocal of pwp + ws beside pwp and delta: 9 216 B
                   // more at T = 4) would cost in residency alone -- three workgroups per CU instead of five
    __shared__ char s_pad[RF_SKEW_PAD];
    if (threadIdx.x == 0 && a.W < 0) s_pad[a.H & 1023] = 1; // (never true: keeps the array)
    asm volatile("" ::"v"(&s_pad[threadIdx.x]));
#endif
    const DirArgs &d = a.d[blockIdx.z];
    const int W = a.W, H = a.H;
    const int XL = d.own.XL, XR = d.own.XR, YL = d.own.YL, YR = d.own.YR;
    const int xa = XL + 1 + (int)blockIdx.x * UW, xb = min(xa + UW, XR);                 // owned columns [xa, xb)
    const int ya = YL + 1 + (int)blockIdx.y * a.skew_rows, yb = min(ya + a.skew_rows, YR); // owned rows [ya, yb)
    if (xa >= XR || ya >= YR) return; // workgroup-uniform
    exp_tab_stage(s_exp); // (the first update is at least two of the loop's barriers away)
    const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), t = wid + 1; // this wave's sweep within the launch
    const int x = xa - T + lane;
    const int xc = min(max(x, 0), W - 1);
    const int y0 = max(YL, ya - T), y1 = min(YR, yb - 1 + T); // staged rows [y0, y1]
    const double *__restrict__ in = d.f64_a;
    double *__restrict__ out = d.f64_b;
    const size_t way1 = a.rf_stride;
    const bool xown = x >= xa && x < xb;
    const unsigned shard = (blockIdx.x + blockIdx.y * gridDim.x + blockIdx.z * 7u) & (RF_UPD_SHARDS - 1);
    int32_t *cnt = a.upd_cnt + (a.flag3 & 1) * RF_UPD_SHARDS + shard;
    // what sweep t can compute here
    const int cy_lo = max(YL + 1, ya - (T - t)), cy_hi = min(YR - 1, yb - 1 + (T - t));
    const int cx_lo = max(XL + 1, xa - (T - t)), cx_hi = min(XR - 1, xb - 1 + (T - t));
    const bool colok = x >= cx_lo && x <= cx_hi;
    const unsigned long long m_col = __builtin_amdgcn_ballot_w64(colok), m_own = __builtin_amdgcn_ballot_w64(xown);
    const bool wsok = a.ws >= 0x1p-200 && a.ws <= 0x1p200; // refine_update3m's division guard
    // A row is loaded two steps before it is needed (its loads are issued in step row - 2, it goes to LDS at the end of
    // step row - 1 and is first read in step row): one step of update math does not cover the memory latency under
    // load.  Waves 0 and 1 take the even and the odd rows, so a wave's staging registers are busy for two steps and the
    // loop stays rolled.
    double nd = 0, np0 = 0, nq0 = 0, np1 = 0, nq1 = 0;
    uint32_t nk0 = 0, nk1 = 0; // packed only when they go to LDS: nothing may consume a loaded value in the step that issues the load
    const uint16_t *__restrict__ keys = (const uint16_t *)d.rf_key;
    // V & 1 (T = 4): the staging of a row is shared by two waves -- waves 0 / 1 take the state and the keys of the even / odd rows,
    // waves 2 / 3 both ways' (pwp, delta) -- so that every wave stages every other step (3 or 4 loads, 2 or 3 LDS writes) instead
    // of two waves carrying all seven loads and five writes while the other two wait at the barrier.
    constexpr bool BAL = (V & 1) && T == 4;
    const bool st_lo = !BAL || wid < 2, st_hi = !BAL || wid >= 2; // what this wave stages: state + keys / the entries
    unsigned xb8 = (unsigned)xc * 8u, xb2 = (unsigned)xc * 2u; // the column as 32-bit BYTE offsets (not const: see load_row)
    auto load_row = [&](int row) {
        const size_t p = (size_t)row * W;
        // scalar row pointer + 32-bit lane byte offset = the load's own addressing mode: no vector address arithmetic at all
        // (the empty asm keeps the compiler from widening the offsets into 64-bit lane addresses that live across the loop)
        asm volatile("" : "+v"(xb8), "+v"(xb2));
        // (the pointers as they come from the kernel arguments: laundering the cache arrays' bases into scalar registers so that the
        // staging wave does not re-read them every step, with explicitly global loads, measured 5 % SLOWER -- 317 against 335 Mdisp/s)
        auto ld8 = [&](const double *q) { return *(const double *)((const char *)q + xb8); };
        auto ld2 = [&](const uint16_t *q) { return (uint32_t) * (const uint16_t *)((const char *)q + xb2); };
        if (st_lo) {
            nd = ld8(in + p);
            nk0 = ld2(keys + p);
            nk1 = ld2(keys + p + way1);
        }
        if (st_hi) {
            np0 = ld8(d.rf_pwp + p);
            nq0 = ld8(d.rf_delta + p);
            np1 = ld8(d.rf_pwp + p + way1);
            nq1 = ld8(d.rf_delta + p + way1);
        }
    };
    const int spar = BAL ? (wid & 1) : wid; // this wave stages the rows of this parity (!BAL: waves 0 and 1 only)
    if (spar == (y0 & 1)) load_row(y0);
    // This wave's row in step s is r = s - 2t + 1; e = (r - y0) % NE is kept as a counter.
    int r = y0 - 2 * t, e = ((r - y0) % NE + NE) % NE;
// Timing experiments (results invalid; -DRF_SKEW_EXP=bits: 1 no staging loads, 2 no update math, 4 no barrier, 8 misses
// ignored) and the per-phase shader-clock split (-DRF_SKEW_TIMING) exist at compile time only: DESIGN.md 4 quotes them.
#ifdef RF_SKEW_EXP
#define RF_EXP(b) ((RF_SKEW_EXP) & (b))
#else
#define RF_EXP(b) 0
#endif
#ifdef RF_SKEW_TIMING
    unsigned long long tm[5] = {0, 0, 0, 0, 0}, tq0, tq1;
    int nsteps = 0;
#define RF_TICK(i)                                     \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); \
    tq1 = __builtin_amdgcn_s_memtime();                \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); \
    tm[i] += tq1 - tq0;                                \
    tq0 = tq1;
#else
#define RF_TICK(i)
#endif
#pragma unroll 1
    for (int s = y0 - 1; s <= y1 + 2 * T - 1; s++) {
#ifdef RF_SKEW_TIMING
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        tq0 = __builtin_amdgcn_s_memtime();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        nsteps++;
#endif
        if (spar == (s & 1) && s + 2 <= y1 && !RF_EXP(1)) load_row(s + 2);
        RF_TICK(0) // staging loads issued
        if (r >= y0 && r <= y1) { // wave-uniform
            // one LDS round trip: the five state values, the keys and BOTH ways' entries (the way depends on dC)
            double dC, dN, dS, dE, dW;
            uint32_t kk;
            double2 e0 = make_double2(0.0, 0.0), e1 = e0;
            auto operands = [&]() {
                dC = s_d[t - 1][r & 3][lane + 1];
                dN = s_d[t - 1][(r - 1) & 3][lane + 1];
                dS = s_d[t - 1][(r + 1) & 3][lane + 1];
                dE = s_d[t - 1][r & 3][lane + 2];
                dW = s_d[t - 1][r & 3][lane];
                if (!(V & 16)) { // V & 16: only the way the state selects is read, in a second LDS round trip (below)
                    e0 = s_ent[e][0][lane];
                    e1 = s_ent[e][1][lane];
                }
                kk = s_key[e][lane];
            };
            operands();
            if (V & 4) { // (one LDS round trip: none of the reads may sink below the live-row test, which needs dC only)
                asm volatile("" : "+v"(dN), "+v"(dS), "+v"(dE), "+v"(dW), "+v"(kk));
                if (!(V & 16)) asm volatile("" : "+v"(e0.x), "+v"(e0.y), "+v"(e1.x), "+v"(e1.y));
            }
            double val = dC;
            RF_TICK(1) // LDS operands arrived
            if ((V & 10) && r >= cy_lo && r <= cy_hi && (!(V & 4) || (m_col & RF_FNE(dC, (double)NOMATCH)))) { // wave-uniform: rows sweep t can compute here; predicates as lane masks
                const double NM = (double)NOMATCH;
                const unsigned long long m_lv = m_col & RF_FNE(dC, NM); // .cpp:613
                const unsigned long long m_ew = RF_FNE(dE, NM) & RF_FNE(dW, NM), m_ns = RF_FNE(dS, NM) & RF_FNE(dN, NM); // .cpp:620
                const int rel = (int)(dC - 1.5); // .cpp:625 (iMatch - x)
                const int way = rel & 1;
                const int crel = (int)(int16_t)(kk >> (way << 4));
                const unsigned long long m_miss = RF_EXP(8) ? 0ull : m_lv & (m_ew | m_ns) & ~RF_IEQ(crel, rel);
                // (keeps both entry reads in the first LDS batch: they are dead on the miss path, which reloads them, and
                // would otherwise sink below it and add a second LDS round trip to every row)
                if (!(V & 16)) asm volatile("" : "+v"(e0.x), "+v"(e0.y), "+v"(e1.x), "+v"(e1.y));
                if (m_miss) { // wave-uniform, rare
                    skew_miss(a, d, W, H, x, xa - T, r, rel, way, lane, rf_sel(m_miss), xown && r >= ya && r < yb, cnt, shard, s_ent[e], s_key[e], s_emit[e], kk, s_ml[wid], blockIdx.z);
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    operands(); // from LDS again (the entries now with the new ones): nothing lives in registers across the data-term routine
                    asm volatile("" : "+v"(dC), "+v"(dN), "+v"(dS), "+v"(dE), "+v"(dW), "+v"(kk));
                    if (!(V & 16)) asm volatile("" : "+v"(e0.x), "+v"(e0.y), "+v"(e1.x), "+v"(e1.y));
                    val = dC;
                }
                const double2 pd = (V & 16) ? s_ent[e][way][lane] : (way ? e1 : e0);
                if (!RF_EXP(2)) {
                    if (!(m_lv & ~(m_ew & m_ns))) { // every live pixel of the row is mode 3: straight-line code on all lanes
                        const double u = (V & 8) ? refine_update3e(dC, dE, dW, dN, dS, pd.x, pd.y, a.ws, m_lv, s_exp)
                                                 : refine_update3m(dC, dE, dW, dN, dS, pd.x, pd.y, a.ws, m_lv, wsok, s_exp);
                        val = rf_sel(m_lv) ? u : dC;
                    } else if (rf_sel(m_lv)) {
                        const int mode = (int)rf_sel(m_ew) + (int)rf_sel(m_ns) * 2; // .cpp:620
                        if (mode != 0) val = refine_update(mode, dC, dE, dW, dN, dS, pd.x, pd.y, a.ws, s_exp);
                    }
                }
                if (t == T && rf_sel(m_lv & m_own)) { // sweep T's computable rows are the owned rows (xc == x there)
                    asm volatile("" : "+v"(xb8)); // scalar row pointer + 32-bit lane byte offset (no hoisted 64-bit lane address)
                    *(double *)((char *)(out + (size_t)r * W) + xb8) = val;
                }
            }
            // V & 4: a row without a live pixel in this strip (outside an elliptic mask, a hole) copies through without the update
            // math -- one more compare per row, chosen by the launcher where the masked pixels leave much of the margin's box empty
            if (!(V & 10) && r >= cy_lo && r <= cy_hi && (!(V & 4) || (m_col & RF_FNE(dC, (double)NOMATCH)))) { // round 3's form of the same
                const bool lv = colok && dC != (double)NOMATCH; // .cpp:613
                const bool ew = dE != (double)NOMATCH && dW != (double)NOMATCH, ns = dS != (double)NOMATCH && dN != (double)NOMATCH;
                const int rel = (int)(dC - 1.5); // .cpp:625 (iMatch - x)
                const int way = rel & 1;
                const int crel = (int)(int16_t)(kk >> (way << 4));
                const bool miss = lv && (ew || ns) && crel != rel && !RF_EXP(8);
                // (keeps both entry reads in the first LDS batch: they are dead on the miss path, which reloads them, and
                // would otherwise sink below it and add a second LDS round trip to every row)
                asm volatile("" : "+v"(e0.x), "+v"(e0.y), "+v"(e1.x), "+v"(e1.y));
                if (__ballot(miss)) { // wave-uniform, rare
                    skew_miss(a, d, W, H, x, xa - T, r, rel, way, lane, miss, xown && r >= ya && r < yb, cnt, shard, s_ent[e], s_key[e], s_emit[e], kk, s_ml[wid], blockIdx.z);
                    // the operands come from LDS again (the entries now with the new ones) instead of living in
                    // registers across the data-term routine: the common path keeps its 96 registers unspilled
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    operands();
                    // (pins the reloads inside this rare block: merged into the join below they would run for every row)
                    asm volatile("" : "+v"(dC), "+v"(dN), "+v"(dS), "+v"(dE), "+v"(dW), "+v"(kk));
                    asm volatile("" : "+v"(e0.x), "+v"(e0.y), "+v"(e1.x), "+v"(e1.y));
                    val = dC;
                }
                const double2 pd = way ? e1 : e0;
                if (!RF_EXP(2)) {
                    if (!__ballot(lv && !(ew && ns))) { // every live pixel of the row is mode 3: straight-line code on all lanes
                        const double u = refine_update3(dC, dE, dW, dN, dS, pd.x, pd.y, a.ws, lv, s_exp);
                        val = lv ? u : dC;
                    } else if (lv) {
                        const int mode = (int)ew + (int)ns * 2; // .cpp:620
                        if (mode != 0) val = refine_update(mode, dC, dE, dW, dN, dS, pd.x, pd.y, a.ws, s_exp);
                    }
                }
                if (t == T && lv && xown) { // sweep T's computable rows are the owned rows (xc == x there)
                    asm volatile("" : "+v"(xb8)); // scalar row pointer + 32-bit lane byte offset (no hoisted 64-bit lane address)
                    *(double *)((char *)(out + (size_t)r * W) + xb8) = val;
                }
            }
            if (t < T) s_d[t][r & 3][lane + 1] = val;
        }
        RF_TICK(2) // update math + result write
        if (spar == ((s + 1) & 1) && s + 1 <= y1) {
            const int es = (s + 1 - y0) % NE;
            if (st_lo) {
                s_d[0][(s + 1) & 3][lane + 1] = nd;
                s_key[es][lane] = nk0 | (nk1 << 16);
                if (lane < 2) s_emit[es][lane] = 0ull;
            }
            if (st_hi) {
                s_ent[es][0][lane] = make_double2(np0, nq0);
                s_ent[es][1][lane] = make_double2(np1, nq1);
            }
        }
        r++;
        e = (e + 1 == NE) ? 0 : e + 1;
        RF_TICK(3) // staged row written to LDS (waits for its loads)
        if (!RF_EXP(4)) __syncthreads();
        RF_TICK(4) // barrier
    }
#ifdef RF_SKEW_TIMING
    if (TOP && lane == 0 && a.flag3 == 12 && blockIdx.z == 0 && (blockIdx.x % 9) == 4 && (blockIdx.y % 5) == 2)
        printf("skewtime wg %d %d wave %d steps %d: issue %llu lds %llu math %llu stage %llu barrier %llu\n", (int)blockIdx.x, (int)blockIdx.y, wid, nsteps,
               tm[0] / nsteps, tm[1] / nsteps, tm[2] / nsteps, tm[3] / nsteps, tm[4] / nsteps);
#endif
}

#endif // RF_TU == 2
#if RF_TU == 1 // k_refine_skew1.hip
// ---------------------------------------------------------------------------------------------------------------
// The same four time-skewed sweeps with ONE wave per strip (option refine_skew_variant = 64; round 5).
// What bounds k_refine_skew is not the vector unit (busy half of the launch) but the time its waves spend outside the
// update math: a wave per sweep level means a workgroup barrier per row, an LDS round trip for the five state values
// another wave wrote, and two of four waves waiting while the others stage.  tests/micro/ilp_probe.hip:
// two resident waves per SIMD already issue dependent fp64 fma chains at 90 % of what any occupancy reaches -- so here
// a single wave owns the strip and advances all four levels itself, one row each per step:
//   - the state rings (4 levels x 4 rows) live in REGISTERS: the loop is unrolled by the ring period, every index is
//     static; E / W neighbours are lane shifts (v_mov_b32_dpp wave_shr:1 / wave_shl:1), N / S the ring's other rows;
//   - no barrier at all (the two waves of a workgroup share nothing but the exp table), no LDS traffic for the state;
//   - the four row updates of a step are independent (levels two rows apart) and written four wide, straight-line and
//     branch-free: every special case (a row with a border pixel, a weight argument >= 512, pwp == 0) is detected by
//     lane masks BESIDE the chain and redone afterwards by the general form -- rare, and nothing waits for the test;
//   - both cache ways of the 8 rows in flight stay in LDS (32 B per pixel is what limits residency: 18 KB per strip,
//     8 strips per CU = 2 waves per SIMD); each lane reads only what it wrote itself, or what the miss service wrote
//     behind a wave fence.
// Same trapezoid of computable pixels, same miss service (skew_miss), same update list: the bits of four single sweeps.
__device__ __forceinline__ double lane_from_left(double v) { // lane i <- lane i - 1 (lane 0: 0.0; tests/micro/ilp_probe.hip)
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), 0x138, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), 0x138, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double lane_from_right(double v) { // lane i <- lane i + 1 (lane 63: 0.0)
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), 0x130, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), 0x130, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}

// The mode-3 update (.cpp:652-672) of N independent rows at once, every operation written N wide and the stages fenced for
// the scheduler (left alone it finishes one row's chain before it starts the next: a dependent fp64 operation issues every
// other slot at best).  refine_update3m's arithmetic: the t < 512 form of the specified exp and both divisions as the
// hardware sequence without its operand scaling and fix-up (div_unscaled: no VCC, so the N sequences interleave) -- the
// bits of the general form wherever the guard holds; bad[i] = the lanes where it does not (a weight argument above 200, a
// zero or tiny numerator, which includes pwp == 0 with a zero smoothness term), for the caller to redo.  Nothing waits for
// the guard: it is evaluated beside the chain.  ws in [2^-200, 2^200] is the caller's test (kernel-uniform).
#define RF_STAGE // (a stage boundary; the translation unit's scheduling strategy keeps the stages' operations side by side)
template <int N>
__device__ __forceinline__ void refine_update3_wide(const double (&dC)[N], const double (&dE)[N], const double (&dW)[N], const double (&dN)[N],
                                                    const double (&dS)[N], const double2 (&pd)[N], double ws, ExpTab tab, double (&u)[N],
                                                    unsigned long long (&bad)[N]) {
    double t[2 * N], w[2 * N];
#pragma unroll
    for (int i = 0; i < N; i++) {
        const double ex = fabs(dE[i] - dC[i]) - fabs(dW[i] - dC[i]);
        const double ey = fabs(dS[i] - dC[i]) - fabs(dN[i] - dC[i]);
        t[2 * i] = ex * ex;
        t[2 * i + 1] = ey * ey;
    }
    constexpr int G = 2 * N < 4 ? 2 * N : 4; // weights per group
#pragma unroll
    for (int g = 0; g < 2 * N; g += G) { // exp_neg_small's operations (exp_core + the final fma), stage by stage
        double k[G], r[G], r2[G], pa[G], pb[G], tm[G];
        uint32_t ki[G];
        double2 e[G];
#pragma unroll
        for (int i = 0; i < G; i++) {
            k[i] = __builtin_fma(-t[g + i], EXP_INVLN2N, EXP_SHIFT);
            ki[i] = (uint32_t)__double2loint(k[i]);
            e[i] = tab[ki[i] & 127u];
        }
        RF_STAGE;
#pragma unroll
        for (int i = 0; i < G; i++) k[i] = k[i] - EXP_SHIFT;
#pragma unroll
        for (int i = 0; i < G; i++) r[i] = __builtin_fma(k[i], EXP_NEGLN2HIN, -t[g + i]);
        RF_STAGE;
#pragma unroll
        for (int i = 0; i < G; i++) r[i] = __builtin_fma(k[i], EXP_NEGLN2LON, r[i]);
        RF_STAGE;
#pragma unroll
        for (int i = 0; i < G; i++) {
            r2[i] = r[i] * r[i];
            pa[i] = __builtin_fma(r[i], EXP_C3, EXP_C2);
            pb[i] = __builtin_fma(r[i], EXP_C5, EXP_C4);
            tm[i] = r[i] + e[i].x;
        }
        RF_STAGE;
#pragma unroll
        for (int i = 0; i < G; i++) {
            tm[i] = __builtin_fma(pa[i], r2[i], tm[i]);
            r2[i] = r2[i] * r2[i];
        }
        RF_STAGE;
#pragma unroll
        for (int i = 0; i < G; i++) {
            tm[i] = __builtin_fma(r2[i], pb[i], tm[i]);
            k[i] = __hiloint2double(__double2hiint(e[i].y) + (int)(ki[i] << 13), __double2loint(e[i].y)); // the scale 2^e H[j]
        }
        RF_STAGE;
#pragma unroll
        for (int i = 0; i < G; i++) w[g + i] = __builtin_fma(k[i], tm[i], k[i]);
        RF_STAGE;
    }
    // ds = (wx (dE + dW) + wy (dN + dS)) / (2 (wx + wy)), .cpp:669
    double a1[N], b1[N], y[N], q[N], ee[N];
#pragma unroll
    for (int i = 0; i < N; i++) {
        a1[i] = w[2 * i] * (dE[i] + dW[i]) + w[2 * i + 1] * (dN[i] + dS[i]);
        b1[i] = 2 * (w[2 * i] + w[2 * i + 1]);
    }
#define RF_DIV_WIDE(A, B, Q)                                                      \
    _Pragma("unroll") for (int i = 0; i < N; i++) y[i] = __builtin_amdgcn_rcp(B[i]); \
    RF_STAGE;                                                                     \
    _Pragma("unroll") for (int i = 0; i < N; i++) ee[i] = __builtin_fma(-B[i], y[i], 1.0); \
    RF_STAGE;                                                                     \
    _Pragma("unroll") for (int i = 0; i < N; i++) y[i] = __builtin_fma(y[i], ee[i], y[i]); \
    RF_STAGE;                                                                     \
    _Pragma("unroll") for (int i = 0; i < N; i++) ee[i] = __builtin_fma(-B[i], y[i], 1.0); \
    RF_STAGE;                                                                     \
    _Pragma("unroll") for (int i = 0; i < N; i++) y[i] = __builtin_fma(y[i], ee[i], y[i]); \
    RF_STAGE;                                                                     \
    _Pragma("unroll") for (int i = 0; i < N; i++) Q[i] = A[i] * y[i];             \
    RF_STAGE;                                                                     \
    _Pragma("unroll") for (int i = 0; i < N; i++) ee[i] = __builtin_fma(-B[i], Q[i], A[i]); \
    RF_STAGE;                                                                     \
    _Pragma("unroll") for (int i = 0; i < N; i++) Q[i] = __builtin_fma(ee[i], y[i], Q[i]); \
    RF_STAGE;
    RF_DIV_WIDE(a1, b1, q) // (div_unscaled's eight operations)
    double a2[N], b2[N];
#pragma unroll
    for (int i = 0; i < N; i++) {
        a2[i] = (dC[i] + pd[i].y) * pd[i].x + ws * q[i]; // .cpp:671
        b2[i] = pd[i].x + ws;
    }
    RF_DIV_WIDE(a2, b2, u)
#undef RF_DIV_WIDE
#pragma unroll
    for (int i = 0; i < N; i++)
        bad[i] = RF_FGT(fmax(t[2 * i], t[2 * i + 1]), 200.0) | ~(RF_FGT(fabs(a1[i]), 0x1p-300) & RF_FGT(fabs(a2[i]), 0x1p-300));
}

// Timing experiments (results invalid; -DRF_SKEW1_EXP=bits: 1 no loads of the cache rows, 2 no update math, 4 no LDS staging writes,
// 8 no loads of the state, 16 no miss service -- the code-size question, 32 no result store, 128 a store per row whatever happened) exist at compile time only: DESIGN.md 4 quotes them.
#ifdef RF_SKEW1_EXP
#define S1_EXP(b) ((RF_SKEW1_EXP) & (b))
#else
#define S1_EXP(b) 0
#endif
#ifdef RF_SKEW1_TIMING // per-phase shader-clock split of a step (s_memtime), printed by a few waves of launch 12
#define S1_TICK(i)                                     \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); \
    tq1 = __builtin_amdgcn_s_memtime();                \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); \
    tm[i] += tq1 - tq0;                                \
    tq0 = tq1;
#else
#define S1_TICK(i)
#endif
template <int TOP>
__global__ __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_refine_skew1(StageArgs a) {
    constexpr int T = 4, NE = 8, UW = 64 - 2 * T;
    __shared__ double2 s_exp[128];
    __shared__ double2 s_ent[2][NE][2][64];        // [wave][row & 7][way][lane] = (pwp, delta)
    __shared__ uint32_t s_key[2][NE][64];          // key of way 0 | key of way 1 << 16
    __shared__ unsigned long long s_emit[2][NE][2]; // the lanes whose cache slot this launch already listed a new entry for
    __shared__ uint8_t s_ml[2][64];                // the lanes of a row's misses (skew_miss)
    exp_tab_stage(s_exp);
    __syncthreads(); // (the only one)
    // Workgroup -> (strip pair, chunk, direction), XCD-aware: the hardware deals consecutive workgroup ids round-robin to the 8 XCDs,
    // each with its own L2.  Neighbouring strips share 8 of their 64 columns and the two partial 128-byte lines at the ends of
    // every row segment they write: id = 8 j + k is given the j-th tile of the k-th eighth of the launch, so that an XCD works
    // on a contiguous range of strips (option: -DRF_SKEW1_NO_XCD_MAP keeps the plain order).
    unsigned bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
#ifndef RF_SKEW1_NO_XCD_MAP
    {
        const unsigned nwg = gridDim.x * gridDim.y * gridDim.z, lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
        const unsigned per = nwg >> 3, body = per << 3; // (the last nwg % 8 workgroups keep their place)
        if (lin < body) {
            const unsigned m = (lin & 7u) * per + (lin >> 3);
            bx = m % gridDim.x;
            by = (m / gridDim.x) % gridDim.y;
            bz = m / (gridDim.x * gridDim.y);
        }
    }
#endif
    const DirArgs &d = a.d[bz];
    const int W = a.W, H = a.H;
    const int XL = d.own.XL, XR = d.own.XR, YL = d.own.YL, YR = d.own.YR;
    const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int strip = (int)bx * 2 + wid;
    const int xa = XL + 1 + strip * UW, xb = min(xa + UW, XR);                            // owned columns [xa, xb)
    const int ya = YL + 1 + (int)by * a.skew_rows, yb = min(ya + a.skew_rows, YR); // owned rows [ya, yb)
    if (xa >= XR || ya >= YR) return; // wave-uniform
    const int x = xa - T + lane;
    const int xc = min(max(x, 0), W - 1);
    const int y0 = max(YL, ya - T), y1 = min(YR, yb - 1 + T); // staged rows [y0, y1]
    // the five arrays' base pointers stay in scalar registers for the whole launch (laundered: left to itself the compiler
    // re-reads them from the kernel arguments in every step -- three dependent scalar loads and their waits at the head of a step)
    // (explicitly global pointers: behind the asm the compiler no longer knows they came from kernel arguments and would use flat accesses)
    typedef const double __attribute__((address_space(1))) * GCD;
    typedef double __attribute__((address_space(1))) * GD;
    typedef const uint16_t __attribute__((address_space(1))) * GCU16;
    GCD in = (GCD)d.f64_a, c_pwp = (GCD)d.rf_pwp, c_delta = (GCD)d.rf_delta;
    GD out = (GD)d.f64_b;
    GCU16 keys = (GCU16)d.rf_key;
    size_t way1 = a.rf_stride;
    asm volatile("" : "+s"(in), "+s"(c_pwp), "+s"(c_delta), "+s"(out), "+s"(keys), "+s"(way1));
    const bool xown = x >= xa && x < xb;
    const unsigned long long m_own = __builtin_amdgcn_ballot_w64(xown);
    const unsigned shard = ((unsigned)strip + by * gridDim.x * 2u + bz * 7u) & (RF_UPD_SHARDS - 1);
    int32_t *cnt = a.upd_cnt + (a.flag3 & 1) * RF_UPD_SHARDS + shard;
    double2(*ent)[2][64] = s_ent[wid];
    uint32_t(*key)[64] = s_key[wid];
    unsigned long long(*emit)[2] = s_emit[wid];
    const bool wsok = a.ws >= 0x1p-200 && a.ws <= 0x1p200; // refine_update3_wide's division guard
    // what sweep t (1..4) can compute here: the trapezoid of k_refine_skew.  Columns: sweep t computes x in [max(XL + 1, xa - (T - t)),
    // min(XR - 1, xb - 1 + (T - t))], i.e. the lanes with t <= tmax (one compare per level instead of four lane masks in scalar
    // registers); rows: sweep t computes row r of [max(YL + 1, ya - (T - t)), min(YR - 1, yb - 1 + (T - t))], all four of them
    // while s is in [st_lo, st_hi]
    const int tmax = (x >= XL + 1 && x <= XR - 1) ? T - max(0, max(xa - x, x - (xb - 1))) : 0;
    int st_lo = INT_MIN, st_hi = INT_MAX;
#pragma unroll
    for (int t = 1; t <= T; t++) {
        st_lo = max(st_lo, max(YL + 1, ya - (T - t)) + 2 * t - 1);
        st_hi = min(st_hi, min(YR - 1, yb - 1 + (T - t)) + 2 * t - 1);
    }
    double R[T][4];             // [level: 0 = the launch's input, t = sweep t's result][row & 3]
#ifndef RF_SKEW1_PF
#define RF_SKEW1_PF 4
#endif
    constexpr int PF = RF_SKEW1_PF; // a row's loads are issued PF steps before the step that first uses it (2 or 4: the staging registers
                                    // are indexed statically by row & (PF - 1)); with 2 a wave waits most of a step for them
    double gd[PF], gp0[PF], gq0[PF], gp1[PF], gq1[PF]; // staging registers
    uint16_t gk0[PF], gk1[PF];
#pragma unroll
    for (int t = 0; t < T; t++)
#pragma unroll
        for (int j = 0; j < 4; j++) R[t][j] = (double)NOMATCH;
#pragma unroll
    for (int j = 0; j < PF; j++) gd[j] = gp0[j] = gq0[j] = gp1[j] = gq1[j] = 0.0, gk0[j] = gk1[j] = 0;
    unsigned xb8 = (unsigned)xc * 8u, xb2 = (unsigned)xc * 2u; // the column as 32-bit byte offsets beside a scalar row pointer
    const unsigned lds_lane16 = (unsigned)lane * 16u;
    double pend_v = 0.0;          // sweep T's row of the previous step: stored AFTER this step's loads have been issued (a store at the
    unsigned long long pend_m = 0; // end of a step sits in front of the next step's loads in the memory pipeline and holds them up)
    uint32_t kk[T] = {0u, 0u, 0u, 0u}; // the keys of the four rows of the NEXT step: read a step ahead, off the step's critical path

#ifdef RF_SKEW1_TIMING
    unsigned long long tm[6] = {0, 0, 0, 0, 0, 0}, tq0 = 0, tq1 = 0;
    int nsteps = 0;
#endif
    // one step: s = the row the input level has reached; level t advances row s - 2t + 1.  PH = s & 3 (static).
    auto step = [&](int s, auto ph) __attribute__((always_inline)) {
        constexpr int PH = decltype(ph)::value;
#ifdef RF_SKEW1_TIMING
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        tq0 = __builtin_amdgcn_s_memtime();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        nsteps++;
#endif
        { // the loads of row s + PF (PF steps ahead of its first use), into the registers of its residue
            constexpr int g = (PH + PF) & (PF - 1);
            const size_t p = (size_t)(S1_EXP(256) ? y0 + ((s + PF) & 7) : min(max(s + PF, y0), y1)) * W; // (256, timing only: the same eight rows over and over -- cache hits)
            asm volatile("" : "+v"(xb8), "+v"(xb2));
            if (!S1_EXP(8)) gd[g] = *(GCD)((const char __attribute__((address_space(1))) *)(in + p) + xb8);
            if (!S1_EXP(9)) {
                gk0[g] = *(GCU16)((const char __attribute__((address_space(1))) *)(keys + p) + xb2);
                gk1[g] = *(GCU16)((const char __attribute__((address_space(1))) *)(keys + p + way1) + xb2);
                gp0[g] = *(GCD)((const char __attribute__((address_space(1))) *)(c_pwp + p) + xb8);
                gq0[g] = *(GCD)((const char __attribute__((address_space(1))) *)(c_delta + p) + xb8);
                gp1[g] = *(GCD)((const char __attribute__((address_space(1))) *)(c_pwp + p + way1) + xb8);
                gq1[g] = *(GCD)((const char __attribute__((address_space(1))) *)(c_delta + p + way1) + xb8);
            }
        }
        if (rf_sel(pend_m) && !S1_EXP(32) && (!S1_EXP(512) || a.W < 0)) { // the previous step's result row (s - 1) - 2T + 1; owned lanes: xc == x (512, timing only: the store never executes, the math stays)
            asm volatile("" : "+v"(xb8));
            *(GD)((char __attribute__((address_space(1))) *)(out + (size_t)(s - 2 * T) * W) + xb8) = pend_v;
        }
        pend_m = 0ull;
        S1_TICK(0) // staging loads issued
        const double NM = (double)NOMATCH;
        double dC[T], val[T];
        unsigned long long m_lv[T], any_lv = 0ull;
        unsigned long long rm[T] = {~0ull, ~0ull, ~0ull, ~0ull};
        if (__builtin_expect(!(s >= st_lo && s <= st_hi), 0)) { // the chunk's first and last steps: not every level has a computable row
#pragma unroll
            for (int t = 1; t <= T; t++) {
                const int r = s - 2 * t + 1;
                rm[t - 1] = (r >= max(YL + 1, ya - (T - t)) && r <= min(YR - 1, yb - 1 + (T - t))) ? ~0ull : 0ull;
            }
        }
#pragma unroll
        for (int t = 1; t <= T; t++) {
            const int i = t - 1;
            dC[i] = R[i][(PH - 2 * t + 1) & 3];
            m_lv[i] = RF_FNE(dC[i], NM) & RF_IGE(tmax, t) & rm[i]; // .cpp:613
            any_lv |= m_lv[i];
            val[i] = dC[i];
        }
        if (any_lv && !S1_EXP(2)) { // (a strip without a live pixel on any of its four rows copies through: the masked parts of the margin's box)
            int slot[T];
#pragma unroll
            for (int i = 0; i < T; i++) slot[i] = (s - 2 * i - 1) & (NE - 1);
            double dN[T], dS[T], dE[T], dW[T];
            int rel[T], way[T];
            unsigned long long m_ew[T], m_ns[T], m_miss[T], any_miss = 0ull;
#pragma unroll
            for (int t = 1; t <= T; t++) {
                const int i = t - 1, c = (PH - 2 * t + 1) & 3;
                dN[i] = R[i][(c + 3) & 3];
                dS[i] = R[i][(c + 1) & 3];
                dE[i] = lane_from_right(dC[i]);
                dW[i] = lane_from_left(dC[i]);
                m_ew[i] = RF_FNE(dE[i], NM) & RF_FNE(dW[i], NM); // .cpp:620
                m_ns[i] = RF_FNE(dS[i], NM) & RF_FNE(dN[i], NM);
                rel[i] = (int)(dC[i] - 1.5); // .cpp:625 (iMatch - x)
                way[i] = rel[i] & 1;
            }
#pragma unroll
            for (int i = 0; i < T; i++) {
                const int crel = (int)(int16_t)(kk[i] >> (way[i] << 4));
                m_miss[i] = m_lv[i] & (m_ew[i] | m_ns[i]) & ~RF_IEQ(crel, rel[i]);
                any_miss |= m_miss[i];
            }
            S1_TICK(1) // keys arrived, masks
            if (any_miss && !S1_EXP(16)) { // wave-uniform, rare once the iteration has settled: the miss service, a level at a time
#pragma unroll 1
                for (int i = 0; i < T; i++) {
                    const unsigned long long mm = i == 0 ? m_miss[0] : i == 1 ? m_miss[1] : i == 2 ? m_miss[2] : m_miss[3];
                    if (!mm) continue;
                    const int rl = i == 0 ? rel[0] : i == 1 ? rel[1] : i == 2 ? rel[2] : rel[3];
                    const uint32_t kv = i == 0 ? kk[0] : i == 1 ? kk[1] : i == 2 ? kk[2] : kk[3];
                    const int r = s - 2 * i - 1, sl = r & (NE - 1);
                    skew_miss(a, d, W, H, x, xa - T, r, rl, rl & 1, lane, rf_sel(mm), xown && r >= ya && r < yb, cnt, shard, ent[sl], key[sl], emit[sl], kv, s_ml[wid], bz);
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
            double2 pd[T];
#pragma unroll
            for (int i = 0; i < T; i++) // only the way the state selects: [slot][way][lane] of this wave's rows
                pd[i] = *(const double2 *)((const char *)&ent[0][0][0] + (((unsigned)slot[i] << 11) + ((unsigned)way[i] << 10) + lds_lane16));
            S1_TICK(2) // miss branch, entry reads issued + arrived (timing build: waited for)
            double u[T];
            unsigned long long bad[T], any_bad = 0ull;
#ifndef RF_SKEW1_WIDTH
#define RF_SKEW1_WIDTH 2
#endif
            constexpr int NW = RF_SKEW1_WIDTH; // rows per call of the wide update (4: more registers than two waves per SIMD have)
#pragma unroll
            for (int g = 0; g < T; g += NW) {
                double gC[NW], gE[NW], gW[NW], gN[NW], gS[NW], gu[NW];
                double2 gp[NW];
                unsigned long long gb[NW];
#pragma unroll
                for (int i = 0; i < NW; i++) gC[i] = dC[g + i], gE[i] = dE[g + i], gW[i] = dW[g + i], gN[i] = dN[g + i], gS[i] = dS[g + i], gp[i] = pd[g + i];
                refine_update3_wide<NW>(gC, gE, gW, gN, gS, gp, a.ws, s_exp, gu, gb);
#pragma unroll
                for (int i = 0; i < NW; i++) u[g + i] = gu[i], bad[g + i] = gb[i];
            }
#pragma unroll
            for (int i = 0; i < T; i++) {
                bad[i] = m_lv[i] & ((wsok ? bad[i] : ~0ull) | ~(m_ew[i] & m_ns[i]));
                any_bad |= bad[i];
            }
            S1_TICK(3) // update math
            if (any_bad) { // wave-uniform: a row with a pixel at the border of the valid region (.cpp:620's other modes) or one of the update's rare cases
#pragma unroll
                for (int i = 0; i < T; i++)
                    if (bad[i]) {
                        double g = dC[i];
                        if (rf_sel(m_lv[i])) {
                            const int mode = (int)rf_sel(m_ew[i]) + (int)rf_sel(m_ns[i]) * 2;
                            if (mode != 0) g = refine_update(mode, dC[i], dE[i], dW[i], dN[i], dS[i], pd[i].x, pd[i].y, a.ws, s_exp);
                        }
                        u[i] = g;
                    }
            }
#pragma unroll
            for (int i = 0; i < T; i++) val[i] = rf_sel(m_lv[i]) ? u[i] : dC[i];
            pend_v = val[T - 1]; // sweep T's computable rows are the owned rows
            pend_m = m_lv[T - 1] & m_own;
        }
        if (S1_EXP(128)) pend_v = dC[T - 1], pend_m = m_own; // (timing only: a store per row even without the math)
        S1_TICK(4) // rare cases, result store
#pragma unroll
        for (int i = 0; i < T; i++) kk[i] = key[(s - 2 * i) & (NE - 1)][lane]; // the next step's rows s - 2i: staged before this step, none of them touched by this step's miss service (rows s - 2i - 1)
#pragma unroll
        for (int t = 1; t < T; t++) R[t][(PH - 2 * t + 1) & 3] = val[t - 1];
        { // row s + 1 (loaded in step s + 1 - PF) becomes visible: the state to the ring, the entries to this wave's LDS rows
            constexpr int g = (PH + 1) & (PF - 1);
            asm volatile("" : "+v"(gk0[g]), "+v"(gk1[g])); // (the keys' zero-extension HERE: hoisted, it would wait for the loads steps early)
            const int es = (s + 1) & (NE - 1);
            R[0][(PH + 1) & 3] = gd[g];
            if (!S1_EXP(4)) {
                key[es][lane] = (uint32_t)gk0[g] | ((uint32_t)gk1[g] << 16);
                ent[es][0][lane] = make_double2(gp0[g], gq0[g]);
                ent[es][1][lane] = make_double2(gp1[g], gq1[g]);
                if (lane < 2) emit[es][lane] = 0ull;
            }
        }
        S1_TICK(5) // staged row to the ring and LDS (waits for its loads)
    };
    // the first staged row is y0: its loads go out in step y0 - PF; the loop starts on a multiple of the ring period
    const int s_first = ((y0 - PF + 8) & ~3) - 8; // (y0 >= 0)
#pragma unroll 1
    for (int s = s_first; s <= y1 + 2 * T - 1; s += 4) {
        step(s, std::integral_constant<int, 0>());
        step(s + 1, std::integral_constant<int, 1>());
        step(s + 2, std::integral_constant<int, 2>());
        step(s + 3, std::integral_constant<int, 3>());
    }
    // (nothing is pending here: the loop runs past the last computable row of sweep T by at least one step)
#ifdef RF_SKEW1_TIMING
    if (TOP && lane == 0 && a.flag3 == 5 && bz == 0 && (bx % 9) == 4 && (by % 5) == 2)
        printf("skew1time wg %d %d wave %d steps %d: issue %llu keys %llu entries %llu math %llu rare+store %llu stage %llu\n", (int)bx, (int)by, wid, nsteps,
               tm[0] / nsteps, tm[1] / nsteps, tm[2] / nsteps, tm[3] / nsteps, tm[4] / nsteps, tm[5] / nsteps);
#endif
}

// (its own translation unit, k_refine_skew1.hip: compiled with the max-ILP scheduling strategy, which is what interleaves the rows' chains)
void launch_refine_skew1(const StageArgs &a, dim3 grid, hipStream_t st) {
    if (a.flag) hipLaunchKernelGGL(k_refine_skew1<1>, grid, dim3(128), 0, st, a);
    else hipLaunchKernelGGL(k_refine_skew1<0>, grid, dim3(128), 0, st, a);
}
#endif // RF_TU == 1

#if RF_TU == 2
// T sweeps f64_a -> f64_b in one launch (a.flag3 = launch index, a.skew_rows = rows per chunk) + the launch that applies
// its cache updates.  T in {2, 3, 4}.
void launch_refine_skew(const StageArgs &a, int T, hipStream_t st, hipEvent_t ev0, hipEvent_t ev1) {
    int rows = 0, cols = 0;
    for (int v = 0; v < a.ndir; v++) {
        rows = max(rows, a.d[v].own.YR - a.d[v].own.YL - 1);
        cols = max(cols, a.d[v].own.XR - a.d[v].own.XL - 1);
    }
    if (rows <= 0 || cols <= 0 || a.skew_rows <= 0) return;
    const int uw = 64 - 2 * T;
    const dim3 grid((cols + uw - 1) / uw, (rows + a.skew_rows - 1) / a.skew_rows, a.ndir);
    if (ev0) (void)hipEventRecord(ev0, st);
    if (T == 4 && (a.skew_variant & 64)) { // one wave per strip, two strips per workgroup
        launch_refine_skew1(a, dim3((grid.x + 1) / 2, grid.y, grid.z), st);
        if (ev1) (void)hipEventRecord(ev1, st);
        launch_refine_apply(a, st);
        return;
    }
#define RF_LAUNCH_V(TT, VV)                                                                                   \
    do {                                                                                                      \
        if (a.flag) hipLaunchKernelGGL((k_refine_skew<TT, 1, VV>), grid, dim3(64 * TT), 0, st, a);          \
        else hipLaunchKernelGGL((k_refine_skew<TT, 0, VV>), grid, dim3(64 * TT), 0, st, a);                 \
    } while (0)
    if (T == 2) RF_LAUNCH_V(2, 0);
    else if (T == 3) RF_LAUNCH_V(3, 0);
    else // the variants exist for T = 4 only (option refine_skew_variant; 28 = the shipped kernel)
        switch (a.skew_variant & 31) {
        case 1: RF_LAUNCH_V(4, 1); break;
        case 2: RF_LAUNCH_V(4, 2); break;
        case 3: RF_LAUNCH_V(4, 3); break;
        case 4: RF_LAUNCH_V(4, 4); break;
        case 12: RF_LAUNCH_V(4, 12); break;
        case 28: RF_LAUNCH_V(4, 28); break;
        default: RF_LAUNCH_V(4, 0); break;
        }
#undef RF_LAUNCH_V
#undef RF_LAUNCH
    if (ev1) (void)hipEventRecord(ev1, st);
    launch_refine_apply(a, st);
}
#endif // RF_TU == 2

#if RF_TU == 0
// scatters a launch's update list into the cache (after k_refine_skew / k_refine_skew1 / k_refine_multi)
void launch_refine_apply(const StageArgs &a, hipStream_t st) { hipLaunchKernelGGL(k_refine_apply, dim3(8, RF_UPD_SHARDS), dim3(256), 0, st, a); }

void launch_refine_multi(const StageArgs &a, hipStream_t st, hipEvent_t ev0, hipEvent_t ev1) {
    int rows = 0, cols = 0;
    for (int v = 0; v < a.ndir; v++) {
        rows = max(rows, a.d[v].own.YR - a.d[v].own.YL - 1);
        cols = max(cols, a.d[v].own.XR - a.d[v].own.XL - 1);
    }
    if (rows <= 0 || cols <= 0) return;
    const dim3 grid((cols + RM_TW - 3) / (RM_TW - 2), (rows + RM_TR - 1) / RM_TR, a.ndir);
    if (ev0) (void)hipEventRecord(ev0, st);
    if (a.flag) hipLaunchKernelGGL(k_refine_multi<1>, grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL(k_refine_multi<0>, grid, dim3(256), 0, st, a);
    if (ev1) (void)hipEventRecord(ev1, st);
    hipLaunchKernelGGL(k_refine_apply, dim3(8, RF_UPD_SHARDS), dim3(256), 0, st, a);
}

// One sweep f64_a -> f64_b (a.flag2 = sweep index, a.flag = top level) over the interior rows [a.row_lo, a.row_hi)
// (the first sweep always covers the whole interior); ev0 / ev1 (optional) bracket the launch.
void launch_refine_sweep(const StageArgs &a, hipStream_t st, hipEvent_t ev0, hipEvent_t ev1) {
    int rows = 0, cols = 0;
    for (int v = 0; v < a.ndir; v++) {
        int lo = a.d[v].own.YL + 1, hi = a.d[v].own.YR; // interior rows [lo, hi)
        if (a.flag2 != 0) {
            lo = max(lo, a.row_lo);
            hi = min(hi, a.row_hi);
        }
        rows = max(rows, hi - lo);
        cols = max(cols, a.d[v].own.XR - a.d[v].own.XL - 1);
    }
    if (rows <= 0 || cols <= 0) return;
    const dim3 grid((cols + 255) / 256, rows, a.ndir);
    if (ev0) (void)hipEventRecord(ev0, st);
    if (a.flag2 == 0) { // first sweep: every pixel needs its data term
        hipLaunchKernelGGL(k_refine_first, grid, dim3(256), 0, st, a);
    } else {
        const dim3 sgrid(grid.x, (grid.y + RF_PPT - 1) / RF_PPT, grid.z);
        if (a.defer) {
            if (a.flag) hipLaunchKernelGGL((k_refine_sweep<1, 1>), sgrid, dim3(256), 0, st, a);
            else hipLaunchKernelGGL((k_refine_sweep<0, 1>), sgrid, dim3(256), 0, st, a);
        } else {
            if (a.flag) hipLaunchKernelGGL((k_refine_sweep<1, 0>), sgrid, dim3(256), 0, st, a);
            else hipLaunchKernelGGL((k_refine_sweep<0, 0>), sgrid, dim3(256), 0, st, a);
        }
    }
    if (ev1) (void)hipEventRecord(ev1, st);
}

#endif
