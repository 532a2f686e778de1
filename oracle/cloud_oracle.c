/*
 * cloud_oracle.c -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement of the per-pair cloud filter of CCloudOptimization::filter
 * (CloudOptimization/CCloudOptimization.cpp:82-121 of seed93/reconstruction):
 *   pcl::StatisticalOutlierRemoval (meanK = 100, stddevMulThresh = 1; CReconstruction.cpp:18)  :85-89
 *   pcl::NormalEstimationOMP with radius search 2.5 (m_mls_radius)                               :103-109
 *   normals turned toward CamCenter                                                              :114-121
 * PCL 1.6 / 1.7.2 is a third-party dependency that is NOT in /root/reference (CCloudOptimization.h:22-40 only
 * includes it); the algorithms are restated from PCL 1.7.2's published sources
 * (filters/impl/statistical_outlier_removal.hpp: applyFilterIndices; features/normal_3d.h: computePointNormal,
 * flipNormalTowardsViewpoint; common/impl/centroid.hpp: computeMeanAndCovarianceMatrix; common/impl/eigen.hpp:
 * computeRoots, eigen33).  PARITY UNPINNED: the reference holds no test or vector for this stage.
 * Deliberate, documented choices where PCL's result depends on its search structure or on float accumulation:
 *   - neighbour distances are float32, (dx*dx + dy*dy) + dz*dz on float32 differences (FLANN L2_Simple), the k
 *     nearest are the k smallest of those values (a k-d tree returns the same multiset), ties by value;
 *   - radius search keeps dist^2 < r^2 (FLANN RadiusResultSet), the query point included;
 *   - centroid / covariance sums are accumulated in double (PCL: float), the 3x3 eigen-solve in double.
 * Brute force O(n^2): meant for clouds of up to ~10^5 points.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static int cmp_float(const void *a, const void *b) {
    const float x = *(const float *)a, y = *(const float *)b;
    return (x > y) - (x < y);
}

static inline int finite3(const float *p) { return isfinite(p[0]) && isfinite(p[1]) && isfinite(p[2]); }

static inline float dist2f(const float *a, const float *b) {
    const float dx = a[0] - b[0], dy = a[1] - b[1], dz = a[2] - b[2];
    return (dx * dx + dy * dy) + dz * dz;
}

/* statistical_outlier_removal.hpp: mean distance of every point to its mean_k nearest neighbours (the point itself,
 * first of the mean_k + 1 results, skipped), then mean / stddev over the cloud, keep distance <= mean + mul * stddev.
 * dist[i]: the per-point mean distance (float, as PCL stores it); stats = {mean, stddev, threshold}. */
void orc_sor_filter(const float *xyz, int64_t n, int mean_k, double std_mul, uint8_t *keep, float *dist, double *stats) {
#pragma omp parallel
    {
        float *d2 = (float *)malloc(sizeof(float) * (size_t)(n > 0 ? n : 1));
#pragma omp for schedule(dynamic, 64)
        for (int64_t i = 0; i < n; i++) {
            if (!finite3(xyz + 3 * i)) { /* "distances[iii] = 0.0; continue;": skipped by the searches, kept by the filter */
                dist[i] = 0.0f;
                continue;
            }
            int64_t nn = 0;
            for (int64_t j = 0; j < n; j++)
                if (finite3(xyz + 3 * j)) d2[nn++] = dist2f(xyz + 3 * i, xyz + 3 * j);
            qsort(d2, (size_t)nn, sizeof(float), cmp_float);
            const int64_t m = (nn < (int64_t)mean_k + 1) ? nn : (int64_t)mean_k + 1;
            double sum = 0.0;
            for (int64_t k = 1; k < m; k++) sum += (double)sqrtf(d2[k]); /* nn_dists[0] is the point itself */
            dist[i] = (float)(sum / mean_k);
        }
        free(d2);
    }
    double sum = 0.0, sq_sum = 0.0;
    for (int64_t i = 0; i < n; i++) {
        sum += dist[i];
        sq_sum += dist[i] * dist[i]; /* float * float, as in PCL */
    }
    int64_t nv = 0; /* valid_distances */
    for (int64_t i = 0; i < n; i++) nv += finite3(xyz + 3 * i);
    const double mean = sum / (double)nv;
    const double variance = (sq_sum - sum * sum / (double)nv) / ((double)nv - 1);
    const double stddev = sqrt(variance);
    const double thr = mean + std_mul * stddev;
    for (int64_t i = 0; i < n; i++) keep[i] = !(dist[i] > thr);
    stats[0] = mean;
    stats[1] = stddev;
    stats[2] = thr;
}

/* eigen.hpp: computeRoots2 / computeRoots (roots ascending) */
static void roots2(double b, double c, double *r) {
    r[0] = 0.0;
    double d = b * b - 4.0 * c;
    if (d < 0.0) d = 0.0;
    const double sd = sqrt(d);
    r[2] = 0.5 * (b + sd);
    r[1] = 0.5 * (b - sd);
}
static void roots3(const double m[9], double *r) {
    const double c0 = m[0] * m[4] * m[8] + 2.0 * m[1] * m[2] * m[5] - m[0] * m[5] * m[5] - m[4] * m[2] * m[2] - m[8] * m[1] * m[1];
    const double c1 = m[0] * m[4] - m[1] * m[1] + m[0] * m[8] - m[2] * m[2] + m[4] * m[8] - m[5] * m[5];
    const double c2 = m[0] + m[4] + m[8];
    if (fabs(c0) < 2.220446049250313e-16) {
        roots2(c2, c1, r);
        return;
    }
    const double s_inv3 = 1.0 / 3.0, s_sqrt3 = sqrt(3.0);
    const double c2_over_3 = c2 * s_inv3;
    double a_over_3 = (c1 - c2 * c2_over_3) * s_inv3;
    if (a_over_3 > 0.0) a_over_3 = 0.0;
    const double half_b = 0.5 * (c0 + c2_over_3 * (2.0 * c2_over_3 * c2_over_3 - c1));
    double q = half_b * half_b + a_over_3 * a_over_3 * a_over_3;
    if (q > 0.0) q = 0.0;
    const double rho = sqrt(-a_over_3);
    const double theta = atan2(sqrt(-q), half_b) * s_inv3;
    const double ct = cos(theta), st = sin(theta);
    r[0] = c2_over_3 + 2.0 * rho * ct;
    r[1] = c2_over_3 - rho * (ct + s_sqrt3 * st);
    r[2] = c2_over_3 - rho * (ct - s_sqrt3 * st);
    if (r[0] >= r[1]) { const double t = r[0]; r[0] = r[1]; r[1] = t; }
    if (r[1] >= r[2]) {
        const double t = r[1]; r[1] = r[2]; r[2] = t;
        if (r[0] >= r[1]) { const double u = r[0]; r[0] = r[1]; r[1] = u; }
    }
    if (r[0] <= 0.0) roots2(c2, c1, r);
}

/* eigen33 (smallest eigenvalue and its vector) + solvePlaneParameters' curvature; cov row-major symmetric */
void orc_plane_from_cov(const double cov[9], double nrm[3], double *curvature) {
    double scale = 0.0;
    for (int i = 0; i < 9; i++) scale = fmax(scale, fabs(cov[i]));
    if (scale <= 2.2250738585072014e-308) scale = 1.0;
    double m[9];
    for (int i = 0; i < 9; i++) m[i] = cov[i] / scale;
    double r[3];
    roots3(m, r);
    const double ev = r[0] * scale;
    m[0] -= r[0];
    m[4] -= r[0];
    m[8] -= r[0];
    const double *r0 = m, *r1 = m + 3, *r2 = m + 6;
    const double v1[3] = {r0[1] * r1[2] - r0[2] * r1[1], r0[2] * r1[0] - r0[0] * r1[2], r0[0] * r1[1] - r0[1] * r1[0]};
    const double v2[3] = {r0[1] * r2[2] - r0[2] * r2[1], r0[2] * r2[0] - r0[0] * r2[2], r0[0] * r2[1] - r0[1] * r2[0]};
    const double v3[3] = {r1[1] * r2[2] - r1[2] * r2[1], r1[2] * r2[0] - r1[0] * r2[2], r1[0] * r2[1] - r1[1] * r2[0]};
    const double l1 = v1[0] * v1[0] + v1[1] * v1[1] + v1[2] * v1[2];
    const double l2 = v2[0] * v2[0] + v2[1] * v2[1] + v2[2] * v2[2];
    const double l3 = v3[0] * v3[0] + v3[1] * v3[1] + v3[2] * v3[2];
    const double *v = v3;
    double l = l3;
    if (l1 >= l2 && l1 >= l3) { v = v1; l = l1; }
    else if (l2 >= l1 && l2 >= l3) { v = v2; l = l2; }
    const double s = sqrt(l);
    for (int i = 0; i < 3; i++) nrm[i] = v[i] / s;
    const double tr = cov[0] + cov[4] + cov[8];
    *curvature = (tr != 0.0) ? fabs(ev / tr) : 0.0;
}

/* normal_3d.h computePointNormal over the radius neighbourhood + flipNormalTowardsViewpoint(vp = origin, the
 * default: the reference calls setViewPoint only after compute, :109) + the reference's own turn toward CamCenter
 * (:114-121).  normals: n x 4 floats (nx, ny, nz, curvature); fewer than 3 neighbours -> NaN. */
void orc_cloud_normals(const float *xyz, int64_t n, double radius, const float *cam_center, float *normals) {
    const float r2 = (float)(radius * radius);
#pragma omp parallel for schedule(dynamic, 64)
    for (int64_t i = 0; i < n; i++) {
        double a[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        int64_t cnt = 0;
        for (int64_t j = 0; j < n && finite3(xyz + 3 * i); j++) {
            if (!finite3(xyz + 3 * j) || !(dist2f(xyz + 3 * i, xyz + 3 * j) < r2)) continue;
            const double x = xyz[3 * j], y = xyz[3 * j + 1], z = xyz[3 * j + 2];
            a[0] += x * x; a[1] += x * y; a[2] += x * z; a[3] += y * y; a[4] += y * z; a[5] += z * z;
            a[6] += x; a[7] += y; a[8] += z;
            cnt++;
        }
        float *o = normals + 4 * i;
        if (cnt < 3) {
            o[0] = o[1] = o[2] = o[3] = NAN;
            continue;
        }
        for (int k = 0; k < 9; k++) a[k] /= (double)cnt;
        double cov[9];
        cov[0] = a[0] - a[6] * a[6];
        cov[1] = cov[3] = a[1] - a[6] * a[7];
        cov[2] = cov[6] = a[2] - a[6] * a[8];
        cov[4] = a[3] - a[7] * a[7];
        cov[5] = cov[7] = a[4] - a[7] * a[8];
        cov[8] = a[5] - a[8] * a[8];
        double nv[3], curv;
        orc_plane_from_cov(cov, nv, &curv);
        const double px = xyz[3 * i], py = xyz[3 * i + 1], pz = xyz[3 * i + 2];
        if ((0.0 - px) * nv[0] + (0.0 - py) * nv[1] + (0.0 - pz) * nv[2] < 0.0) { nv[0] = -nv[0]; nv[1] = -nv[1]; nv[2] = -nv[2]; }
        float nf[3] = {(float)nv[0], (float)nv[1], (float)nv[2]};
        /* :116-120 in float (Eigen::Vector3f) */
        const float cx = cam_center[0] - xyz[3 * i], cy = cam_center[1] - xyz[3 * i + 1], cz = cam_center[2] - xyz[3 * i + 2];
        if (nf[0] * cx + nf[1] * cy + nf[2] * cz < 0.0f) { nf[0] = -nf[0]; nf[1] = -nf[1]; nf[2] = -nf[2]; }
        o[0] = nf[0]; o[1] = nf[1]; o[2] = nf[2]; o[3] = (float)curv;
    }
}
