// rectify_host.cpp -- see rectify_host.h.  Plain host C++ (compiled with -ffp-contract=off).
#include "rectify_host.h"

#include <cfloat>
#include <cmath>
#include <cstring>

namespace {
struct M3 {
    double a[9];
    double &operator()(int i, int j) { return a[3 * i + j]; }
    double operator()(int i, int j) const { return a[3 * i + j]; }
};
M3 mul(const M3 &A, const M3 &B) {
    M3 C;
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) C(i, j) = A(i, 0) * B(0, j) + A(i, 1) * B(1, j) + A(i, 2) * B(2, j);
    return C;
}
M3 mul_bt(const M3 &A, const M3 &B) { // A * B^T
    M3 C;
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) C(i, j) = A(i, 0) * B(j, 0) + A(i, 1) * B(j, 1) + A(i, 2) * B(j, 2);
    return C;
}
M3 tr(const M3 &A) {
    M3 C;
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) C(i, j) = A(j, i);
    return C;
}
void mv(const M3 &A, const double *v, double *o) {
    double t[3];
    for (int i = 0; i < 3; i++) t[i] = A(i, 0) * v[0] + A(i, 1) * v[1] + A(i, 2) * v[2];
    std::memcpy(o, t, sizeof t);
}
// cvRodrigues2: rotation vector -> matrix
M3 rodrigues(const double *r) {
    M3 R;
    const double theta = std::sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
    if (theta < DBL_EPSILON) {
        for (int i = 0; i < 9; i++) R.a[i] = (i % 4 == 0) ? 1.0 : 0.0;
        return R;
    }
    const double c = std::cos(theta), s = std::sin(theta), c1 = 1. - c, it = 1. / theta;
    const double x = r[0] * it, y = r[1] * it, z = r[2] * it;
    const double rrt[9] = {x * x, x * y, x * z, x * y, y * y, y * z, x * z, y * z, z * z};
    const double rxm[9] = {0, -z, y, z, 0, -x, -y, x, 0};
    for (int k = 0; k < 9; k++) R.a[k] = c * ((k % 4 == 0) ? 1.0 : 0.0) + c1 * rrt[k] + s * rxm[k];
    return R;
}
// cvRodrigues2: matrix -> rotation vector (OpenCV first re-orthonormalises by SVD; the input here is a product
// of rotation matrices, orthonormal to ~1e-16, so that step is omitted)
void rodrigues_inv(const M3 &R, double *r) {
    double rx = R(2, 1) - R(1, 2), ry = R(0, 2) - R(2, 0), rz = R(1, 0) - R(0, 1);
    const double s = std::sqrt((rx * rx + ry * ry + rz * rz) * 0.25);
    double c = (R(0, 0) + R(1, 1) + R(2, 2) - 1) * 0.5;
    c = c > 1. ? 1. : c < -1. ? -1. : c;
    const double theta = std::acos(c);
    if (s < 1e-5) {
        if (c > 0) {
            r[0] = r[1] = r[2] = 0;
            return;
        }
        double t = (R(0, 0) + 1) * 0.5;
        rx = std::sqrt(t > 0 ? t : 0.);
        t = (R(1, 1) + 1) * 0.5;
        ry = std::sqrt(t > 0 ? t : 0.) * (R(0, 1) < 0 ? -1. : 1.);
        t = (R(2, 2) + 1) * 0.5;
        rz = std::sqrt(t > 0 ? t : 0.) * (R(0, 2) < 0 ? -1. : 1.);
        if (std::fabs(rx) < std::fabs(ry) && std::fabs(rx) < std::fabs(rz) && (R(1, 2) > 0) != (ry * rz > 0)) rz = -rz;
        const double nrm = theta / std::sqrt(rx * rx + ry * ry + rz * rz);
        r[0] = rx * nrm;
        r[1] = ry * nrm;
        r[2] = rz * nrm;
        return;
    }
    const double vth = 1 / (2 * s) * theta;
    r[0] = rx * vth;
    r[1] = ry * vth;
    r[2] = rz * vth;
}
void inv3(const double *M, double *I) {
    const double a = M[0], b = M[1], c = M[2], d = M[3], e = M[4], f = M[5], g = M[6], h = M[7], i = M[8];
    double det = a * (e * i - f * h) - b * (d * i - f * g) + c * (d * h - e * g);
    det = det != 0 ? 1. / det : 0;
    I[0] = (e * i - f * h) * det;
    I[1] = (c * h - b * i) * det;
    I[2] = (b * f - c * e) * det;
    I[3] = (f * g - d * i) * det;
    I[4] = (a * i - c * g) * det;
    I[5] = (c * d - a * f) * det;
    I[6] = (d * h - e * g) * det;
    I[7] = (b * g - a * h) * det;
    I[8] = (a * e - b * d) * det;
}
} // namespace

// cvStereoRectify with D1 = D2 = 0, flags = 0, alpha = -1, newImgSize = imageSize.
void stereo_rectify_host(const double *K1, const double *K2, int nx, int ny, const double *Rin, const double *T,
                         double *R1, double *R2, double *P1, double *P2, double *Q) {
    M3 R;
    std::memcpy(R.a, Rin, sizeof R.a);
    double om[3], t[3], uu[3] = {0, 0, 0}, ww[3];
    rodrigues_inv(R, om);
    for (int i = 0; i < 3; i++) om[i] *= -0.5; // average rotation
    const M3 r_r = rodrigues(om);
    mv(r_r, T, t);
    const int idx = std::fabs(t[0]) > std::fabs(t[1]) ? 0 : 1;
    const double c = t[idx], nt = std::sqrt(t[0] * t[0] + t[1] * t[1] + t[2] * t[2]);
    uu[idx] = c > 0 ? 1 : -1;
    ww[0] = t[1] * uu[2] - t[2] * uu[1]; // global Z rotation axis: t x uu
    ww[1] = t[2] * uu[0] - t[0] * uu[2];
    ww[2] = t[0] * uu[1] - t[1] * uu[0];
    const double nw = std::sqrt(ww[0] * ww[0] + ww[1] * ww[1] + ww[2] * ww[2]);
    if (nw > 0.0) {
        const double sc = std::acos(std::fabs(c) / nt) / nw;
        for (int i = 0; i < 3; i++) ww[i] *= sc;
    }
    const M3 wR = rodrigues(ww);
    const M3 Ra = mul_bt(wR, r_r), Rb = mul(wR, r_r);
    std::memcpy(R1, Ra.a, sizeof Ra.a);
    std::memcpy(R2, Rb.a, sizeof Rb.a);
    mv(Rb, T, t);
    double fc_new = DBL_MAX;
    for (int k = 0; k < 2; k++) {
        const double *A = k == 0 ? K1 : K2;
        const double fc = A[(idx ^ 1) * 3 + (idx ^ 1)];
        fc_new = fc_new < fc ? fc_new : fc;
    }
    double cc[2][2];
    for (int k = 0; k < 2; k++) {
        const double *A = k == 0 ? K1 : K2;
        const double *Rk = k == 0 ? R1 : R2;
        double ax = 0, ay = 0;
        for (int i = 0; i < 4; i++) { // the four image corners, through float32 like OpenCV's point buffers
            const int j = (i < 2) ? 0 : 1;
            const float u = (float)((i % 2) * nx), v = (float)(j * ny);
            const double ifx = 1. / A[0], ify = 1. / A[4];
            const float px = (float)(((double)u - A[2]) * ifx), py = (float)(((double)v - A[5]) * ify);
            const double X = px, Y = py;
            const double x = Rk[0] * X + Rk[1] * Y + Rk[2];
            const double y = Rk[3] * X + Rk[4] * Y + Rk[5];
            double z = Rk[6] * X + Rk[7] * Y + Rk[8];
            z = z ? 1. / z : 1;
            ax += (float)(x * z * fc_new + 0.0);
            ay += (float)(y * z * fc_new + 0.0);
        }
        cc[k][0] = nx / 2 - ax / 4;
        cc[k][1] = ny / 2 - ay / 4;
    }
    if (idx == 0) cc[0][1] = cc[1][1] = (cc[0][1] + cc[1][1]) * 0.5;
    else cc[0][0] = cc[1][0] = (cc[0][0] + cc[1][0]) * 0.5;
    std::memset(P1, 0, 12 * sizeof(double));
    std::memset(P2, 0, 12 * sizeof(double));
    P1[0] = P1[5] = fc_new;
    P1[2] = cc[0][0];
    P1[6] = cc[0][1];
    P1[10] = 1;
    P2[0] = P2[5] = fc_new;
    P2[2] = cc[1][0];
    P2[6] = cc[1][1];
    P2[10] = 1;
    P2[idx * 4 + 3] = t[idx] * fc_new;
    const double q[16] = {1, 0, 0, -cc[0][0], 0, 1, 0, -cc[0][1], 0, 0, 0, fc_new, 0, 0, -1. / t[idx],
                          (idx == 0 ? cc[0][0] - cc[1][0] : cc[0][1] - cc[1][1]) / t[idx]};
    std::memcpy(Q, q, sizeof q);
}

void rectify_plan(const double *K0, const double *K1, const double *E0, const double *E1, int originW, int originH,
                  int lowW, int lowH, int N, RectifyPlan *o) {
    o->W = lowW << (N - 1);
    o->H = lowH << (N - 1);
    o->ksize = 3 * (1 << (N - 1));
    M3 R0, R1;
    double t0[3], t1[3];
    for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++) {
            R0(i, j) = E0[4 * i + j];
            R1(i, j) = E1[4 * i + j];
        }
        t0[i] = E0[4 * i + 3];
        t1[i] = E1[4 * i + 3];
    }
    const M3 R = mul_bt(R1, R0); // :125
    double Rt0[3], T[3];
    mv(R, t0, Rt0);
    for (int i = 0; i < 3; i++) T[i] = -Rt0[i] + t1[i]; // :126
    double P[2][12];
    stereo_rectify_host(K0, K1, originW, originH, R.a, T, o->Rn[0], o->Rn[1], P[0], P[1], o->Q);
    M3 Rn0;
    std::memcpy(Rn0.a, o->Rn[0], sizeof Rn0.a);
    const M3 Rf = mul(tr(R0), tr(Rn0)); // :132
    std::memcpy(o->R_final, Rf.a, sizeof Rf.a);
    double tmp[3];
    mv(tr(R0), t0, tmp);
    for (int i = 0; i < 3; i++) o->T_final[i] = -tmp[i]; // :133
    double Ef[16] = {0};
    const M3 Rft = tr(Rf);
    double mt[3];
    mv(Rft, o->T_final, mt);
    for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++) Ef[4 * i + j] = Rft(i, j);
        Ef[4 * i + 3] = -mt[i];
    }
    Ef[15] = 1;
    o->Q[14] = -o->Q[14]; // :138
    const double scale = (double)lowW / originW * (1 << (N - 1)); // :140
    for (int v = 0; v < 2; v++) {
        for (int j = 0; j < 8; j++) P[v][j] *= scale; // :143
        std::memcpy(o->P[v], P[v], sizeof P[v]);
        const double newA[9] = {P[v][0], P[v][1], P[v][2], P[v][4], P[v][5], P[v][6], P[v][8], P[v][9], P[v][10]};
        M3 A, Rv;
        std::memcpy(A.a, newA, sizeof newA);
        std::memcpy(Rv.a, o->Rn[v], sizeof Rv.a);
        const M3 AR = mul(A, Rv);
        inv3(AR.a, o->ir[v]);
        for (int i = 0; i < 3; i++) // :145
            for (int j = 0; j < 4; j++) {
                double s = 0;
                for (int k = 0; k < 4; k++) s += P[v][4 * i + k] * Ef[4 * k + j];
                o->Pext[v][4 * i + j] = s;
            }
    }
}

void rectify_inv3(const double *M, double *I) { inv3(M, I); }
