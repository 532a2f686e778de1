// refine_common.h -- device functions shared by the DisparityRefine kernels (k_refine.hip, k_refine_skew.hip):
// the specified exp(-t), the update of CStereoMatching.cpp:652-672, the division without operand scaling, and the three
// bit-faithful restatements of the data term (CManageData.cpp:81-90 + arma::dot, .cpp:624-650).
#pragma once
#include "rsm_dev.h"

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
    __builtin_amdgcn_sched_barrier(0);
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

// a cache entry for iMatch - x = rel into its way: the 16-byte record and the way's half of the key dword
__device__ __forceinline__ void rf_store(const DirArgs &d, size_t rf_stride, size_t pix, int rel, double pwp, double delta) {
    const int way = rel & 1;
    d.rf_ent[pix + (size_t)way * rf_stride] = make_double2(pwp, delta);
    ((int16_t *)d.rf_key)[2 * pix + way] = (int16_t)rel;
}

// Lane masks straight from a compare (one v_cmp into a scalar pair), combined with scalar logic; rf_sel turns a mask back into
// a select / branch condition at no cost.  A ballot of a COMBINED bool costs two vector instructions (v_cndmask 0 / 1 + v_cmp).
#define RF_FNE(x, y) __builtin_amdgcn_fcmp((x), (y), 14) // unordered or not equal: C's !=
#define RF_FGT(x, y) __builtin_amdgcn_fcmp((x), (y), 2)  // ordered and greater: C's >
#define RF_FLE(x, y) __builtin_amdgcn_fcmp((x), (y), 5)  // ordered and less-or-equal: C's <= (LLVM FCMP_OLE)
#define RF_IEQ(x, y) __builtin_amdgcn_sicmp((x), (y), 32)
#define rf_sel(m) __builtin_amdgcn_inverse_ballot_w64(m)

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
