// ref_probe.cpp -- runs the parts of the REAL reference that need no OpenCV library symbol and the
// vendored Armadillo primitives on inputs from a blob file, writes their outputs to another blob.
//
// Linked with the reference's own objects (ref_tu_stereo.o / ref_tu_data.o, compiled from the sources
// where they lie) using --unresolved-symbols=ignore-all: nothing is stood in for OpenCV; the functions
// called here (CManageData::WindowToVec, CStereoMatching::FindMargin / OrderConstraint /
// UniquenessContraint<T>) only use header-inline cv::Mat accessors over caller-owned buffers.
// Everything that allocates a cv::Mat (the other stages) cannot run and stays "parity unpinned".
#define __declspec(x)
#define _Longlong long long
#include "SharedInclude.h"
#define private public
#include "CStereoMatching.h"
#undef private

#include <cstdint>
#include <cstdio>
#include <map>
#include <string>
#include <vector>

struct Arr {
    int code, nd;
    std::vector<long long> dims;
    std::vector<unsigned char> data;
    long long count() const { long long c = 1; for (auto d : dims) c *= d; return c; }
    template <class T> T *p() { return (T *)data.data(); }
};
static const int ESZ[5] = {1, 2, 4, 8, 8};
typedef std::map<std::string, Arr> Blob;

static Blob read_blob(const char *path) {
    Blob b;
    FILE *f = fopen(path, "rb");
    if (!f) { perror(path); exit(2); }
    int n; fread(&n, 4, 1, f);
    for (int i = 0; i < n; i++) {
        int ln; fread(&ln, 4, 1, f);
        std::string name(ln, ' '); fread(&name[0], 1, ln, f);
        Arr a; fread(&a.code, 4, 1, f); fread(&a.nd, 4, 1, f);
        a.dims.resize(a.nd); if (a.nd) fread(a.dims.data(), 8, a.nd, f);
        a.data.resize(a.count() * ESZ[a.code]); fread(a.data.data(), 1, a.data.size(), f);
        b[name] = a;
    }
    fclose(f);
    return b;
}
static void write_blob(const char *path, Blob &b) {
    FILE *f = fopen(path, "wb");
    int n = (int)b.size(); fwrite(&n, 4, 1, f);
    for (auto &kv : b) {
        int ln = (int)kv.first.size(); fwrite(&ln, 4, 1, f); fwrite(kv.first.data(), 1, ln, f);
        fwrite(&kv.second.code, 4, 1, f); fwrite(&kv.second.nd, 4, 1, f);
        if (kv.second.nd) fwrite(kv.second.dims.data(), 8, kv.second.nd, f);
        fwrite(kv.second.data.data(), 1, kv.second.data.size(), f);
    }
    fclose(f);
}
static Arr make(int code, std::vector<long long> dims) {
    Arr a; a.code = code; a.nd = (int)dims.size(); a.dims = dims; a.data.resize(a.count() * ESZ[code]); return a;
}
static Boundary bd(const int *m) { Boundary b; b.YL = m[0]; b.YR = m[1]; b.XL = m[2]; b.XR = m[3]; b.width = m[4]; b.height = m[5]; return b; }

int main(int argc, char **argv) {
    if (argc < 3) { fprintf(stderr, "usage: ref_probe in.blob out.blob\n"); return 2; }
    Blob in = read_blob(argv[1]), out;
    CManageData data;
    CStereoMatching sm;
    sm.Init(&data, NULL, 2, 0.03);
    sm.Verbose = 0;
    char key[128];
    fprintf(stderr, "probe: %zu inputs\n", in.size());

    // ---- Armadillo primitives exactly as the path calls them
    for (int i = 0;; i++) {
        snprintf(key, sizeof key, "arma_vec_%d", i);
        if (!in.count(key)) break;
        Arr &v = in[key];
        snprintf(key, sizeof key, "arma_vec_b_%d", i);
        Arr &w = in[key];
        const arma::uword n = (arma::uword)v.count();
        arma::vec a(n), b(n);
        for (arma::uword k = 0; k < n; k++) { a(k) = v.p<double>()[k]; b(k) = w.p<double>()[k]; }
        Arr r = make(3, {4});
        r.p<double>()[0] = arma::mean(a);
        r.p<double>()[1] = arma::norm(a);
        r.p<double>()[2] = arma::dot(a, b);
        arma::vec c = a; c -= arma::mean(c);
        r.p<double>()[3] = arma::norm(c);
        snprintf(key, sizeof key, "arma_out_%d", i); out[key] = r;
    }
    for (int i = 0;; i++) {
        snprintf(key, sizeof key, "median_in_%d", i);
        if (!in.count(key)) break;
        Arr &v = in[key];
        const int n = (int)v.count();
        arma::ivec u(9);
        for (int k = 0; k < n; k++) u(k) = v.p<int>()[k];
        Arr r = make(2, {1});
        r.p<int>()[0] = (int)arma::median(u.rows(0, n - 1)); // CStereoMatching.cpp:799
        snprintf(key, sizeof key, "median_out_%d", i); out[key] = r;
    }
    fprintf(stderr, "probe: arma done\n");
    // ---- CManageData::WindowToVec (CManageData.cpp:81-90)
    if (in.count("w2v_img")) {
        Arr &img = in["w2v_img"]; // H x W x 3
        const int H = (int)img.dims[0], W = (int)img.dims[1];
        Arr &cs = in["w2v_cases"]; // n x 3: x (left edge), y (top row), w
        const int nc = (int)cs.dims[0];
        for (int i = 0; i < nc; i++) {
            const int x = cs.p<int>()[3 * i], y = cs.p<int>()[3 * i + 1], w = cs.p<int>()[3 * i + 2];
            std::vector<uchar *> rows(w);
            for (int k = 0; k < w; k++) rows[k] = img.p<uchar>() + (size_t)(y + k) * W * 3;
            arma::vec u(w * w * 3);
            const double nrm = data.WindowToVec(rows.data(), x, w, u);
            Arr r = make(3, {(long long)u.n_elem + 1});
            r.p<double>()[0] = nrm;
            for (arma::uword k = 0; k < u.n_elem; k++) r.p<double>()[1 + k] = u(k);
            snprintf(key, sizeof key, "w2v_out_%d", i); out[key] = r;
        }
        (void)H;
    }
    fprintf(stderr, "probe: w2v done\n");
    // ---- CStereoMatching::FindMargin (.cpp:1011-1038)
    for (int i = 0;; i++) {
        snprintf(key, sizeof key, "fm_mask_%d", i);
        if (!in.count(key)) break;
        Arr &m = in[key];
        snprintf(key, sizeof key, "fm_r_%d", i);
        sm.MatchBlockRadius = in[key].p<int>()[0];
        cv::Mat mask((int)m.dims[0], (int)m.dims[1], CV_8UC1, m.p<uchar>());
        Boundary b;
        sm.FindMargin(b, mask);
        Arr r = make(2, {6});
        int *o = r.p<int>(); o[0] = b.YL; o[1] = b.YR; o[2] = b.XL; o[3] = b.XR; o[4] = b.width; o[5] = b.height;
        snprintf(key, sizeof key, "fm_out_%d", i); out[key] = r;
    }
    fprintf(stderr, "probe: margin done\n");
    // ---- CStereoMatching::OrderConstraint (.cpp:310-368)
    for (int i = 0;; i++) {
        snprintf(key, sizeof key, "oc_disp_%d", i);
        if (!in.count(key)) break;
        Arr d = in[key];
        snprintf(key, sizeof key, "oc_margin_%d", i);
        const int *m = in[key].p<int>();
        sm.margin[0] = bd(m);       // IsZeroOne = true reads margin[!true] = margin[0]
        cv::Mat disp((int)d.dims[0], (int)d.dims[1], CV_16SC1, d.p<short>());
        sm.OrderConstraint(disp, true);
        snprintf(key, sizeof key, "oc_out_%d", i); out[key] = d;
    }
    fprintf(stderr, "probe: order done\n");
    // ---- CStereoMatching::UniquenessContraint<T> (.cpp:450-461: three UniquenessContraint_<T> passes, :463-497;
    //      the single pass is inlined away in the reference object, the three-pass wrapper is what links)
    for (int i = 0;; i++) {
        snprintf(key, sizeof key, "uq_p_%d", i);
        if (!in.count(key)) break;
        Arr p = in[key];
        snprintf(key, sizeof key, "uq_q_%d", i);
        Arr q = in[key];
        snprintf(key, sizeof key, "uq_margins_%d", i);
        const int *m = in[key].p<int>(); // own(6), oth(6)
        sm.margin[0] = bd(m);     // IsZeroOne = true: own = margin[0], other = margin[1]
        sm.margin[1] = bd(m + 6);
        cv::Mat dd[2];
        if (p.code == 1) {
            dd[0] = cv::Mat((int)p.dims[0], (int)p.dims[1], CV_16SC1, p.p<short>());
            dd[1] = cv::Mat((int)q.dims[0], (int)q.dims[1], CV_16SC1, q.p<short>());
            sm.UniquenessContraint<short>(dd);
        } else {
            dd[0] = cv::Mat((int)p.dims[0], (int)p.dims[1], CV_64FC1, p.p<double>());
            dd[1] = cv::Mat((int)q.dims[0], (int)q.dims[1], CV_64FC1, q.p<double>());
            sm.UniquenessContraint<double>(dd);
        }
        snprintf(key, sizeof key, "uq_out0_%d", i); out[key] = p;
        snprintf(key, sizeof key, "uq_out1_%d", i); out[key] = q;
    }
    fprintf(stderr, "probe: uniqueness done\n");
    // ---- the NCC score of the matchers, evaluated with the reference's own WindowToVec and Armadillo exactly as the
    //      call sites do (.cpp:202-211, :255-257, :290-291): vecL /= normL; CurrentValue = arma::dot(vecL, vecR) / normR.
    //      Output: for every row y in [r, H - r) and every window centre pair (x, c) in [r, W - r)^2 the score.
    for (int i = 0;; i++) {
        snprintf(key, sizeof key, "ncc_imgA_%d", i);
        if (!in.count(key)) break;
        Arr &A = in[key];
        snprintf(key, sizeof key, "ncc_imgB_%d", i);
        Arr &B = in[key];
        snprintf(key, sizeof key, "ncc_r_%d", i);
        const int r = in[key].p<int>()[0];
        const int H = (int)A.dims[0], W = (int)A.dims[1];
        const int ws = 2 * r + 1, nv = ws * ws * 3, ny = H - 2 * r, nx = W - 2 * r;
        Arr sc = make(3, {ny, nx, nx});
        for (int y = r; y < H - r; y++) {
            std::vector<uchar *> rowsL(ws), rowsR(ws);
            for (int k = -r; k <= r; k++) {
                rowsL[k + r] = A.p<uchar>() + (size_t)(y + k) * W * 3;
                rowsR[k + r] = B.p<uchar>() + (size_t)(y + k) * W * 3;
            }
            for (int x = r; x < W - r; x++) {
                arma::vec vecL(nv), vecR(nv);
                double normL = data.WindowToVec(rowsL.data(), x - r, ws, vecL);
                vecL /= normL;
                for (int c = r; c < W - r; c++) {
                    double normR = data.WindowToVec(rowsR.data(), c - r, ws, vecR);
                    double CurrentValue = arma::dot(vecL, vecR) / normR;
                    sc.p<double>()[((size_t)(y - r) * nx + (x - r)) * nx + (c - r)] = CurrentValue;
                }
            }
        }
        snprintf(key, sizeof key, "ncc_scores_%d", i); out[key] = sc;
    }
    // ---- DisparityRefine's data term (.cpp:624-629), evaluated with the reference's own WindowToVec and Armadillo exactly as
    //      the call site does: normL = WindowToVec(window_ptrL, x-1, 3, vecL); normR = WindowToVec(window_ptrR, iMatch+i, 3, vecR);
    //      xi = (1 - arma::dot(vecL, vecR) / (normL * normR)) / 2.  Output: for every row y in [1, H-1), own column x in
    //      [1, W-1) and right-window left edge col in [0, W-3] the matching cost.
    for (int i = 0;; i++) {
        snprintf(key, sizeof key, "xi_imgA_%d", i);
        if (!in.count(key)) break;
        Arr &A = in[key];
        snprintf(key, sizeof key, "xi_imgB_%d", i);
        Arr &B = in[key];
        const int H = (int)A.dims[0], W = (int)A.dims[1];
        Arr tb = make(3, {H - 2, W - 2, W - 2});
        for (int y = 1; y < H - 1; y++) {
            uchar *window_ptrL[3], *window_ptrR[3];
            for (int k = -1; k <= 1; k++) {
                window_ptrL[k + 1] = A.p<uchar>() + (size_t)(y + k) * W * 3;
                window_ptrR[k + 1] = B.p<uchar>() + (size_t)(y + k) * W * 3;
            }
            for (int x = 1; x < W - 1; x++) {
                arma::vec vecL(27), vecR(27);
                double normL = data.WindowToVec(window_ptrL, x - 1, 3, vecL);
                for (int col = 0; col <= W - 3; col++) {
                    double normR = data.WindowToVec(window_ptrR, col, 3, vecR);
                    double xi = (1 - arma::dot(vecL, vecR) / (normL * normR)) / 2;
                    tb.p<double>()[((size_t)(y - 1) * (W - 2) + (x - 1)) * (W - 2) + col] = xi;
                }
            }
        }
        snprintf(key, sizeof key, "xi_table_%d", i); out[key] = tb;
    }
    fprintf(stderr, "probe: refine data term done\n");
    write_blob(argv[2], out);
    printf("ref_probe: %zu outputs\n", out.size());
    return 0;
}
