// rectify_host.h -- host-side fp64 part of CStereoMatching::Rectify (reconstruction/CStereoMatching.cpp:117-145):
// relative pose, cv::stereoRectify (zero distortion, flags 0, alpha -1), R_final / T_final / Extrinsic_final, the
// Q sign flip, scaled P and the inverse matrices the device map kernel needs.  OpenCV 2.4.5's source is not in
// the reference tree; this follows its published algorithm (parity unpinned).
#pragma once

struct RectifyPlan {
    double Rn[2][9];   // R_new[v] (stereoRectify's R1 / R2)
    double P[2][12];   // P[v] with rows 0..1 scaled (:143) -- the new camera matrix of initUndistortRectifyMap
    double Pext[2][12];// P[v] * Extrinsic_final (:145), what cam[v].P holds afterwards
    double ir[2][9];   // (P[v](3x3) * Rn[v])^-1, row-major
    double Q[16];      // after Q(3,2) = -Q(3,2) (:138)
    double R_final[9], T_final[3];
    int W, H;          // largestSize (:120)
    int ksize;         // mask erosion ellipse size 3 * 2^(N-1) (:157)
};

// K[v] 3x3, E[v] 3x4 (MatIntrinsics / MatExtrinsics of cam[pair][v])
void rectify_plan(const double *K0, const double *K1, const double *E0, const double *E1, int originW, int originH,
                  int lowW, int lowH, int N, RectifyPlan *out);
void stereo_rectify_host(const double *K1, const double *K2, int nx, int ny, const double *R, const double *T,
                         double *R1, double *R2, double *P1, double *P2, double *Q);
void rectify_inv3(const double *M, double *I); // 3x3 inverse used for (newCameraMatrix * R)^-1
