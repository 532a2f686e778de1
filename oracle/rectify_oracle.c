/*
 * rectify_oracle.c -- TEST INFRASTRUCTURE ONLY.  CPU restatement of CStereoMatching::Rectify
 * (reconstruction/CStereoMatching.cpp:117-168) and of the OpenCV 2.4.5 operations it calls:
 *   cv::stereoRectify (zero distortion, flags = 0, alpha = -1, newImageSize = imageSize)   .cpp:128-131
 *   cv::initUndistortRectifyMap (CV_16SC2 fixed-point maps, INTER_BITS = 5)                 .cpp:144
 *   cv::remap (INTER_LINEAR, BORDER_CONSTANT 0)                                             .cpp:154,156
 *   cv::erode by getStructuringElement(MORPH_ELLIPSE, 3*2^(N-1))                            .cpp:157-158
 * OpenCV's sources are not in the reference tree (only headers; the binaries are missing): these are restated
 * from OpenCV 2.4's published algorithm.  PARITY UNPINNED.  Known simplification:
 * cvRodrigues2(matrix -> vector) first re-orthonormalises its input by SVD; the input here is a product of
 * rotation matrices (orthonormal to ~1e-16), so that step is skipped.
 */
#include "stereo_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

static void mat3_mul(const double *A, const double *B, double *C) { /* C = A*B */
    double t[9];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) t[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
    memcpy(C, t, sizeof t);
}
static void mat3_mul_bt(const double *A, const double *B, double *C) { /* C = A*B^T */
    double t[9];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) t[3 * i + j] = A[3 * i] * B[3 * j] + A[3 * i + 1] * B[3 * j + 1] + A[3 * i + 2] * B[3 * j + 2];
    memcpy(C, t, sizeof t);
}
static void mat3_t(const double *A, double *C) {
    double t[9];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) t[3 * i + j] = A[3 * j + i];
    memcpy(C, t, sizeof t);
}
static void mat3_vec(const double *A, const double *v, double *o) {
    double t[3];
    for (int i = 0; i < 3; i++) t[i] = A[3 * i] * v[0] + A[3 * i + 1] * v[1] + A[3 * i + 2] * v[2];
    memcpy(o, t, sizeof t);
}

/* cvRodrigues2, rotation vector -> matrix */
void orc_rodrigues_v2m(const double *r, double *R) {
    const double theta = sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
    if (theta < DBL_EPSILON) {
        for (int i = 0; i < 9; i++) R[i] = (i % 4 == 0) ? 1.0 : 0.0;
        return;
    }
    const double c = cos(theta), s = sin(theta), c1 = 1. - c, it = 1. / theta;
    const double rx = r[0] * it, ry = r[1] * it, rz = r[2] * it;
    const double rrt[9] = {rx * rx, rx * ry, rx * rz, rx * ry, ry * ry, ry * rz, rx * rz, ry * rz, rz * rz};
    const double rxm[9] = {0, -rz, ry, rz, 0, -rx, -ry, rx, 0};
    const double I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    for (int k = 0; k < 9; k++) R[k] = c * I[k] + c1 * rrt[k] + s * rxm[k];
}

/* cvRodrigues2, matrix -> rotation vector (without the SVD re-orthonormalisation, see header) */
void orc_rodrigues_m2v(const double *R, double *r) {
    double rx = R[7] - R[5], ry = R[2] - R[6], rz = R[3] - R[1];
    const double s = sqrt((rx * rx + ry * ry + rz * rz) * 0.25);
    double c = (R[0] + R[4] + R[8] - 1) * 0.5;
    c = c > 1. ? 1. : c < -1. ? -1. : c;
    const double theta = acos(c);
    if (s < 1e-5) {
        if (c > 0) {
            r[0] = r[1] = r[2] = 0;
        } else {
            double t;
            t = (R[0] + 1) * 0.5;
            rx = sqrt(t > 0 ? t : 0.);
            t = (R[4] + 1) * 0.5;
            ry = sqrt(t > 0 ? t : 0.) * (R[1] < 0 ? -1. : 1.);
            t = (R[8] + 1) * 0.5;
            rz = sqrt(t > 0 ? t : 0.) * (R[2] < 0 ? -1. : 1.);
            if (fabs(rx) < fabs(ry) && fabs(rx) < fabs(rz) && (R[5] > 0) != (ry * rz > 0)) rz = -rz;
            const double nrm = theta / sqrt(rx * rx + ry * ry + rz * rz);
            r[0] = rx * nrm;
            r[1] = ry * nrm;
            r[2] = rz * nrm;
        }
        return;
    }
    const double vth = 1 / (2 * s) * theta;
    r[0] = rx * vth;
    r[1] = ry * vth;
    r[2] = rz * vth;
}

/* cvStereoRectify with D1 = D2 = 0, flags = 0, alpha = -1, newImgSize = imageSize.
 * K1,K2 3x3; R 3x3, T 3 (pose of camera 2 w.r.t. camera 1); outputs R1,R2 3x3, P1,P2 3x4, Q 4x4 (row-major). */
void orc_stereo_rectify(const double *K1, const double *K2, int nx, int ny, const double *R, const double *T,
                        double *R1, double *R2, double *P1, double *P2, double *Q) {
    double om[3], r_r[9], t[3], uu[3] = {0, 0, 0}, ww[3], wR[9], Ri[9];
    orc_rodrigues_m2v(R, om);
    for (int i = 0; i < 3; i++) om[i] *= -0.5; /* average rotation */
    orc_rodrigues_v2m(om, r_r);
    mat3_vec(r_r, T, t);
    const int idx = fabs(t[0]) > fabs(t[1]) ? 0 : 1;
    const double c = t[idx], nt = sqrt(t[0] * t[0] + t[1] * t[1] + t[2] * t[2]);
    uu[idx] = c > 0 ? 1 : -1;
    ww[0] = t[1] * uu[2] - t[2] * uu[1]; /* t x uu */
    ww[1] = t[2] * uu[0] - t[0] * uu[2];
    ww[2] = t[0] * uu[1] - t[1] * uu[0];
    const double nw = sqrt(ww[0] * ww[0] + ww[1] * ww[1] + ww[2] * ww[2]);
    if (nw > 0.0) {
        const double sc = acos(fabs(c) / nt) / nw;
        for (int i = 0; i < 3; i++) ww[i] *= sc;
    }
    orc_rodrigues_v2m(ww, wR);
    mat3_mul_bt(wR, r_r, Ri); /* R1 = wR * r_r^T */
    memcpy(R1, Ri, sizeof Ri);
    mat3_mul(wR, r_r, Ri); /* R2 = wR * r_r */
    memcpy(R2, Ri, sizeof Ri);
    mat3_vec(Ri, T, t);

    /* new focal length: min over the two cameras of fy (horizontal stereo) / fx (vertical); no distortion */
    double fc_new = DBL_MAX;
    for (int k = 0; k < 2; k++) {
        const double *A = k == 0 ? K1 : K2;
        const double fc = A[(idx ^ 1) * 3 + (idx ^ 1)];
        fc_new = fc_new < fc ? fc_new : fc;
    }
    /* principal points: image corners -> undistortPoints (float32 storage) -> rotate -> project (float32) -> mean */
    double cc_new[2][2];
    for (int k = 0; k < 2; k++) {
        const double *A = k == 0 ? K1 : K2;
        const double *Rk = k == 0 ? R1 : R2;
        float px[4], py[4];
        for (int i = 0; i < 4; i++) {
            const int j = (i < 2) ? 0 : 1;
            const float u = (float)((i % 2) * nx), v = (float)(j * ny);
            const double ifx = 1. / A[0], ify = 1. / A[4];
            px[i] = (float)(((double)u - A[2]) * ifx); /* cvUndistortPoints, zero distortion */
            py[i] = (float)(((double)v - A[5]) * ify);
        }
        double ax = 0, ay = 0;
        for (int i = 0; i < 4; i++) {
            const double X = px[i], Y = py[i], Z = 1.0; /* cvConvertPointsHomogeneous -> (x, y, 1) float32 */
            const double x = Rk[0] * X + Rk[1] * Y + Rk[2] * Z;
            const double y = Rk[3] * X + Rk[4] * Y + Rk[5] * Z;
            double z = Rk[6] * X + Rk[7] * Y + Rk[8] * Z;
            z = z ? 1. / z : 1;
            const float qx = (float)(x * z * fc_new + 0.0), qy = (float)(y * z * fc_new + 0.0); /* cvProjectPoints2 */
            ax += qx;
            ay += qy;
        }
        cc_new[k][0] = nx / 2 - ax / 4; /* (nx)/2: integer division, as in the source */
        cc_new[k][1] = ny / 2 - ay / 4;
    }
    if (idx == 0) cc_new[0][1] = cc_new[1][1] = (cc_new[0][1] + cc_new[1][1]) * 0.5; /* horizontal stereo */
    else cc_new[0][0] = cc_new[1][0] = (cc_new[0][0] + cc_new[1][0]) * 0.5;
    memset(P1, 0, 12 * sizeof(double));
    memset(P2, 0, 12 * sizeof(double));
    P1[0] = P1[5] = fc_new;
    P1[2] = cc_new[0][0];
    P1[6] = cc_new[0][1];
    P1[10] = 1;
    P2[0] = P2[5] = fc_new;
    P2[2] = cc_new[1][0];
    P2[6] = cc_new[1][1];
    P2[10] = 1;
    P2[idx * 4 + 3] = t[idx] * fc_new; /* baseline * focal length */
    /* alpha = -1: no scaling block; newImgSize == imageSize: principal points unchanged */
    const double q[16] = {1, 0, 0, -cc_new[0][0], 0, 1, 0, -cc_new[0][1], 0, 0, 0, fc_new, 0, 0, -1. / t[idx],
                          (idx == 0 ? cc_new[0][0] - cc_new[1][0] : cc_new[0][1] - cc_new[1][1]) / t[idx]};
    memcpy(Q, q, sizeof q);
}

/* 3x3 inverse, Gaussian elimination with partial pivoting (cv::invert DECOMP_LU on a 3x3 uses the closed
 * form; either way the result is exact to ~1e-16 -- the maps are insensitive except at 1/32-px rounding ties) */
static void inv3(const double *M, double *I) {
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

static int sat_int(double v) { /* saturate_cast<int>(double) = cvRound */
    return (int)lrint(v);
}

/* cv::initUndistortRectifyMap(A, zero dist, R, newA (3x3 of P), size, CV_16SC2): map1 = (x,y) int16 pairs,
 * map2 = 5+5 bit fractional index.  The per-row running sums (_x += ir[0] ...) are kept: they fix the rounding. */
void orc_init_rectify_map(const double *A, const double *R, const double *newA, int W, int H, int16_t *map1,
                          uint16_t *map2) {
    double AR[9], ir[9];
    mat3_mul(newA, R, AR);
    inv3(AR, ir);
    const double u0 = A[2], v0 = A[5], fx = A[0], fy = A[4];
    for (int i = 0; i < H; i++) {
        double _x = i * ir[1] + ir[2], _y = i * ir[4] + ir[5], _w = i * ir[7] + ir[8];
        for (int j = 0; j < W; j++, _x += ir[0], _y += ir[3], _w += ir[6]) {
            const double w = 1. / _w, x = _x * w, y = _y * w;
            const double u = fx * x + u0, v = fy * y + v0;
            const int iu = sat_int(u * 32), iv = sat_int(v * 32);
            map1[((long)i * W + j) * 2] = (int16_t)(iu >> 5);
            map1[((long)i * W + j) * 2 + 1] = (int16_t)(iv >> 5);
            map2[(long)i * W + j] = (uint16_t)((iv & 31) * 32 + (iu & 31));
        }
    }
}

/* cv::remap(src, dst, map1, map2, INTER_LINEAR, BORDER_CONSTANT, 0) for 8U, C channels.
 * weights (32-fy)(32-fx)*32 ... sum 32768; result (sum + 16384) >> 15. */
void orc_remap_linear_u8(const uint8_t *src, int Ws, int Hs, int C, const int16_t *map1, const uint16_t *map2,
                         int W, int H, uint8_t *dst) {
#pragma omp parallel for
    for (int i = 0; i < H; i++)
        for (int j = 0; j < W; j++) {
            const int sx = map1[((long)i * W + j) * 2], sy = map1[((long)i * W + j) * 2 + 1];
            const int f = map2[(long)i * W + j] & 1023, fx = f & 31, fy = f >> 5;
            const int w00 = (32 - fy) * (32 - fx) * 32, w01 = (32 - fy) * fx * 32, w10 = fy * (32 - fx) * 32, w11 = fy * fx * 32;
            for (int c = 0; c < C; c++) {
                int v[4];
                for (int k = 0; k < 4; k++) {
                    const int xx = sx + (k & 1), yy = sy + (k >> 1);
                    v[k] = (xx >= 0 && xx < Ws && yy >= 0 && yy < Hs) ? src[((long)yy * Ws + xx) * C + c] : 0;
                }
                dst[((long)i * W + j) * C + c] = (uint8_t)((v[0] * w00 + v[1] * w01 + v[2] * w10 + v[3] * w11 + (1 << 14)) >> 15);
            }
        }
}

/* CStereoMatching::Rectify for one pair (reconstruction/CStereoMatching.cpp:117-168).
 * K[v] 3x3, E[v] 3x4 (MatIntrinsics / MatExtrinsics), origin size (m_OriginSize), lowest size, N = PyrmNum,
 * raw images/masks of origin size.  Outputs: rectified image/mask [2] of size lowest*2^(N-1), Q (sign of
 * Q(3,2) flipped, :138), R_final, T_final, P[v] 3x4 (= scaled P * Extrinsic_final, :143-145). */
void orc_rectify_pair(const double *K0, const double *K1, const double *E0, const double *E1, int originW, int originH,
                      int lowW, int lowH, int N, const uint8_t *const img[2], const uint8_t *const msk[2],
                      uint8_t *rimg[2], uint8_t *rmsk[2], double *Q, double *R_final, double *T_final, double *Pout[2]) {
    const int W = lowW << (N - 1), H = lowH << (N - 1);
    double R0[9], R1m[9], t0[3], t1[3], R[9], T[3];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            R0[3 * i + j] = E0[4 * i + j];
            R1m[3 * i + j] = E1[4 * i + j];
        }
    for (int i = 0; i < 3; i++) {
        t0[i] = E0[4 * i + 3];
        t1[i] = E1[4 * i + 3];
    }
    mat3_mul_bt(R1m, R0, R); /* R = R1 * R0^T, :125 */
    double Rt0[3];
    mat3_vec(R, t0, Rt0);
    for (int i = 0; i < 3; i++) T[i] = -Rt0[i] + t1[i]; /* T = -R*t0 + t1, :126 */
    double Rn[2][9], P[2][12];
    orc_stereo_rectify(K0, K1, originW, originH, R, T, Rn[0], Rn[1], P[0], P[1], Q);
    double R0t[9], Rn0t[9];
    mat3_t(R0, R0t);
    mat3_t(Rn[0], Rn0t);
    mat3_mul(R0t, Rn0t, R_final); /* :132 */
    double tmp[3];
    mat3_vec(R0t, t0, tmp);
    for (int i = 0; i < 3; i++) T_final[i] = -tmp[i]; /* :133 */
    double Ef[16] = {0}; /* Extrinsic_final = [R_final^T | -R_final^T T_final; 0 0 0 1], :134-137 */
    double Rft[9], mt[3];
    mat3_t(R_final, Rft);
    mat3_vec(Rft, T_final, mt);
    for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++) Ef[4 * i + j] = Rft[3 * i + j];
        Ef[4 * i + 3] = -mt[i];
    }
    Ef[15] = 1;
    Q[14] = -Q[14]; /* :138 */
    const double scale = (double)lowW / originW * (1 << (N - 1)); /* :140 */
    const int ksize = 3 * (1 << (N - 1));                          /* :157 */
    int16_t *map1 = (int16_t *)malloc(sizeof(int16_t) * 2 * (size_t)W * H);
    uint16_t *map2 = (uint16_t *)malloc(sizeof(uint16_t) * (size_t)W * H);
    uint8_t *tmpm = (uint8_t *)malloc((size_t)W * H);
    for (int v = 0; v < 2; v++) {
        const double *Kv = v == 0 ? K0 : K1;
        for (int j = 0; j < 8; j++) P[v][j] *= scale; /* P.rowRange(0,2) *= scale, :143 */
        const double newA[9] = {P[v][0], P[v][1], P[v][2], P[v][4], P[v][5], P[v][6], P[v][8], P[v][9], P[v][10]};
        orc_init_rectify_map(Kv, Rn[v], newA, W, H, map1, map2); /* :144 (P as 3x4: its left 3x3 is used) */
        if (Pout && Pout[v]) { /* P = P * Extrinsic_final, :145 */
            for (int i = 0; i < 3; i++)
                for (int j = 0; j < 4; j++) {
                    double s = 0;
                    for (int k = 0; k < 4; k++) s += P[v][4 * i + k] * Ef[4 * k + j];
                    Pout[v][4 * i + j] = s;
                }
        }
        orc_remap_linear_u8(img[v], originW, originH, 3, map1, map2, W, H, rimg[v]); /* :154 */
        orc_remap_linear_u8(msk[v], originW, originH, 1, map1, map2, W, H, tmpm);    /* :156 */
        orc_erode_ellipse_u8(tmpm, W, H, ksize, rmsk[v]);                            /* :157-158 */
    }
    free(map1);
    free(map2);
    free(tmpm);
}
