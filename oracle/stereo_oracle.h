/*
 * stereo_oracle.h -- TEST INFRASTRUCTURE ONLY (parity oracle / CPU baseline).
 *
 * Plain-C restatement of the CStereoMatching pyramidal dense-stereo path of
 * seed93/reconstruction (reference files cited per function as file:line,
 * relative to the reference checkout).  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load this.  The product path
 * (reconstruction_amd/, include/rsm.h) never links or calls it.
 *
 * PARITY PINNING STATUS (see DESIGN.md "Oracle"):
 *   - The reference cannot be linked here (OpenCV 2.4.5 binaries are absent and
 *     may not be stood in for), and it ships no tests/golden vectors.
 *   - Pinned against the real reference: the Armadillo 4.200 primitives this
 *     path uses (mean/norm/dot/median; vendored header-only library, compiled
 *     where it lies by oracle/ref_probe) and the reference functions that
 *     need no OpenCV library symbol (WindowToVec, FindMargin, OrderConstraint,
 *     UniquenessContraint_) -- golden vectors under tests/golden/.
 *   - Everything else (stages that allocate cv::Mat, pyrDown, erode): PARITY
 *     UNPINNED -- restated line by line from the source, checked only by
 *     known-answer tests derivable from the code.
 *   - DisparityRefine's exp() (the C runtime's: MSVC's for the reference, absent here) is evaluated by a fully specified
 *     routine shared with the GPU kernels: orc_exp_neg = the published table-driven algorithm of glibc 2.35's exp
 *     (e_exp.c, third-party, pinned version = this image's libm) in the operation order of its FMA build; pinned against
 *     that libm itself: bit-equal to exp() on 8.7 M arguments (tests/test_oracle_known_answers.py), see stereo_oracle.c.
 */
#ifndef STEREO_ORACLE_H
#define STEREO_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_NOMATCH (-10000) /* CStereoMatching.h:9 */

/* struct Boundary, CManageData.h:10-14 */
typedef struct orc_boundary {
    int YL, YR, XL, XR;
    int width, height;
} orc_boundary;

typedef struct orc_pair_in {
    const uint8_t *image[2]; /* top-level rectified BGR, W*H*3, contiguous rows */
    const uint8_t *mask[2];  /* top-level rectified mask, W*H */
    int width, height;       /* top level = LowestLevelSize * 2^(N-1) */
    int pyr_levels;          /* m_PyrmNum */
    int radius;              /* MatchBlockRadius (CReconstruction.cpp:17 -> 2) */
    double ws;               /* m_ws (0.03) */
    int offset;              /* m_offset (2) */
    int origin_width;        /* m_OriginSize.width, for `scale` CStereoMatching.cpp:692 */
    double Q[16];            /* 4x4 row-major, AFTER the sign flip of CStereoMatching.cpp:138 */
    double R_final[9];       /* 3x3 row-major */
    double T_final[3];
    int verbose;
} orc_pair_in;

typedef struct orc_pair_out {
    double *disparity[2];    /* caller-allocated W*H each (may be NULL) */
    orc_boundary margin[2];  /* top-level margins */
    int64_t n_points;        /* points written */
    int64_t max_points;      /* capacity of xyz/bgr (points) */
    double *xyz;             /* caller-allocated 3*max_points (may be NULL) */
    uint8_t *bgr;            /* caller-allocated 3*max_points (may be NULL) */
    double level_seconds[16];/* wall seconds per pyramid level */
    double refine_seconds;   /* wall seconds inside DisparityRefine, all levels */
    double match_seconds;    /* wall seconds inside the three NCC matchers */
    int64_t v_top;           /* masked view-0 pixels inside margin[0] at the top level */
} orc_pair_out;

/* CManageData.cpp:81-90 (uchar** overload). rows[i] = pointer to row i of the window. */
double orc_window_to_vec(const uint8_t *const *rows, int x, int window_size, double *u);
/* arma primitives as used by the path (op_dot_meat.hpp:20-55, op_mean_meat.hpp:77-86,
 * fn_norm.hpp:84-171, op_median_meat.hpp:361-373) */
double orc_arma_dot(const double *a, const double *b, int n);
double orc_arma_mean(const double *a, int n);
double orc_arma_norm2(const double *a, int n);
int orc_arma_median_int(int *v, int n);

void orc_find_margin(const uint8_t *mask, int W, int H, int r, orc_boundary *m);
void orc_pyr_down_u8(const uint8_t *src, int W, int H, int C, uint8_t *dst);
void orc_erode_ellipse_u8(const uint8_t *src, int W, int H, int ksize, uint8_t *dst);

void orc_lowest_level_initial_match(const uint8_t *img_own, const uint8_t *img_oth,
                                    const uint8_t *mask_own, const uint8_t *mask_oth,
                                    int W, int H, int r,
                                    const orc_boundary *own, const orc_boundary *oth,
                                    int16_t *disp);
void orc_high_level_initial_match(const uint8_t *img_own, const uint8_t *img_oth,
                                  const uint8_t *mask_own, const uint8_t *mask_oth,
                                  int W, int H, int r, int offset,
                                  const orc_boundary *own, const orc_boundary *oth,
                                  const double *parent, int Wp, int Hp,
                                  int16_t *disp);
void orc_smooth_constraint(int16_t *disp, int W, int H, const orc_boundary *own);
void orc_order_constraint(int16_t *disp, int W, int H, const orc_boundary *own);
void orc_uniqueness_pass_s16(int16_t *p, const int16_t *q, int W, int H,
                             const orc_boundary *own, const orc_boundary *oth);
void orc_uniqueness_pass_f64(double *p, const double *q, int W, int H,
                             const orc_boundary *own, const orc_boundary *oth);
/* UniquenessContraint<T>: three passes, margins m0 = margin[0], m1 = margin[1] */
void orc_uniqueness_s16(int16_t *d0, int16_t *d1, int W, int H,
                        const orc_boundary *m0, const orc_boundary *m1);
void orc_uniqueness_f64(double *d0, double *d1, int W, int H,
                        const orc_boundary *m0, const orc_boundary *m1);
int orc_set_boundary_smooth(const int16_t *disp, const uint8_t *mask_own, int W, int H,
                            const orc_boundary *own, const orc_boundary *oth,
                            int16_t *BL, int16_t *BR);
int orc_rematch(const uint8_t *img_own, const uint8_t *img_oth,
                const uint8_t *mask_own, const uint8_t *mask_oth,
                int W, int H, int r,
                const orc_boundary *own, const orc_boundary *oth,
                int16_t *disp);
void orc_median_filter(int16_t *disp, const uint8_t *mask_own, int W, int H,
                       const orc_boundary *own);
void orc_disparity_refine(const int16_t *disp_in, double *disp_out,
                          const uint8_t *img_own, const uint8_t *img_oth,
                          int W, int H, int iterations, double ws,
                          const orc_boundary *own);
int64_t orc_disparity_to_cloud(const double *disp, const uint8_t *mask_org,
                               const uint8_t *img_own, int W, int H,
                               const double *Q, double scale,
                               const double *R_final, const double *T_final,
                               const orc_boundary *own,
                               double *xyz, uint8_t *bgr, int64_t max_points);

/* MatchAllLayer body for one pair (CStereoMatching.cpp:21-29), starting from the
 * rectified top-level images. Returns 0, or <0 on error (-2: degenerate margin,
 * the reference's exit(0) at :827-830). */
int orc_match_pair(const orc_pair_in *in, orc_pair_out *out);

/* ---- Rectify (rectify_oracle.c; OpenCV 2.4 operations restated, parity unpinned) ---- */
void orc_rodrigues_v2m(const double *r, double *R);
void orc_rodrigues_m2v(const double *R, double *r);
void orc_stereo_rectify(const double *K1, const double *K2, int nx, int ny, const double *R, const double *T,
                        double *R1, double *R2, double *P1, double *P2, double *Q);
void orc_init_rectify_map(const double *A, const double *R, const double *newA, int W, int H, int16_t *map1,
                          uint16_t *map2);
void orc_remap_linear_u8(const uint8_t *src, int Ws, int Hs, int C, const int16_t *map1, const uint16_t *map2,
                         int W, int H, uint8_t *dst);
void orc_rectify_pair(const double *K0, const double *K1, const double *E0, const double *E1, int originW, int originH,
                      int lowW, int lowH, int N, const uint8_t *const img[2], const uint8_t *const msk[2],
                      uint8_t *rimg[2], uint8_t *rmsk[2], double *Q, double *R_final, double *T_final, double *Pout[2]);

/* the fully specified exp(-t) of the refine weights (see stereo_oracle.c); mode 1 = host libm instead */
double orc_exp_neg(double t);
void orc_set_exp_mode(int mode); /* 0 specified, 1 host libm exp, 2 host expl rounded, 3 rounds 3-4's Taylor chain (control) */
void orc_set_exp_soft_fma(int soft); /* 1: evaluate fma() through the C library even where the CPU has the instruction */
void orc_exp_neg_array(const double *t, long long n, double *out);
/* DisparityRefine's matching cost xi (CStereoMatching.cpp:624-629) for every (row, own column, other-view left edge) of a small image pair */
void orc_refine_xi_table(const uint8_t *img_own, const uint8_t *img_oth, int W, int H, double *out);

int orc_num_threads(void);
void orc_set_num_threads(int n);

#ifdef __cplusplus
}
#endif
#endif
