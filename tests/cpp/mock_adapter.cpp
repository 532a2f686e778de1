// mock_adapter.cpp -- drives include/rsm_stereo_adapter.hpp (the C++ glue a maintainer drops into the reference tree)
// for real: mock stand-ins of CStereoMatching / CManageData behind a traits type, no OpenCV.  Built with g++ on the
// GPU box by tests/test_gpu_cpp_adapter.py, linked against librsm_mi355.so.
//   mock_adapter <in.bin> <out.bin>
// in.bin : int32 n_pairs, W, H, levels, radius, offset, origin_w, isoutput, bad_pair; double ws;
//          per pair: double Q[16], R[9], T[3]; u8 img0[WH3], img1[WH3], mask0[WH], mask1[WH]
// out.bin: per pair: int32 ok, status; int32 margin[2][6]; int64 n_points (InsertPoint calls); int32 filter_arg;
//          double xyz[n_points*3] as handed to InsertPoint; double disparity0[WH]
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "rsm_stereo_adapter.hpp"

struct MockPair {
    double Q[16], R[9], T[3];
    std::vector<unsigned char> img[2], msk[2];
    rsm_boundary bound[2];
};
struct MockStereo { // the fields of CStereoMatching + CManageData the adapter touches
    int pyr_levels, lowest_w, lowest_h, origin_w, radius, offset, verbose, isoutput;
    double ws;
    int W, H;
    std::vector<MockPair> pairs;
    int cur; // pair whose Q / R_final / T_final are "current" (Rectify sets them per pair)
    rsm_boundary margin[2];
    std::vector<double> inserted;
    std::vector<int> filtered;
};
struct MockTraits {
    typedef MockStereo Stereo;
    static int pyr_levels(Stereo &s) { return s.pyr_levels; }
    static int lowest_width(Stereo &s) { return s.lowest_w; }
    static int lowest_height(Stereo &s) { return s.lowest_h; }
    static int origin_width(Stereo &s) { return s.origin_w; }
    static int radius(Stereo &s) { return s.radius; }
    static double ws(Stereo &s) { return s.ws; }
    static int offset(Stereo &s) { return s.offset; }
    static int verbose(Stereo &s) { return s.verbose; }
    static bool isoutput(Stereo &s) { return s.isoutput != 0; }
    static bool image(Stereo &s, int pair, int v, const unsigned char *&p, int &w, int &h) {
        if (s.pairs[pair].img[v].empty()) return false;
        p = s.pairs[pair].img[v].data(); w = s.W; h = s.H;
        return true;
    }
    static bool mask(Stereo &s, int pair, int v, const unsigned char *&p, int &w, int &h) {
        if (s.pairs[pair].msk[v].empty()) return false;
        p = s.pairs[pair].msk[v].data(); w = s.W; h = s.H;
        return true;
    }
    static double Q(Stereo &s, int i, int j) { return s.pairs[s.cur].Q[4 * i + j]; }
    static double R_final(Stereo &s, int i, int j) { return s.pairs[s.cur].R[3 * i + j]; }
    static double T_final(Stereo &s, int i) { return s.pairs[s.cur].T[i]; }
    static void set_margin(Stereo &s, int pair, int v, const rsm_boundary &m) {
        s.margin[v] = m;
        s.pairs[pair].bound[v] = m;
    }
    static void insert_point(Stereo &s, const double xyz[3]) { s.inserted.insert(s.inserted.end(), xyz, xyz + 3); }
    static void filter(Stereo &s, int pair) { s.filtered.push_back(pair); }
};

template <typename T>
static bool rd(FILE *f, T *p, size_t n) { return fread(p, sizeof(T), n, f) == n; }
template <typename T>
static void wr(FILE *f, const T *p, size_t n) { fwrite(p, sizeof(T), n, f); }

int main(int argc, char **argv) {
    if (argc < 3) return 2;
    FILE *fi = fopen(argv[1], "rb");
    if (!fi) return 2;
    int32_t hdr[9];
    MockStereo s;
    if (!rd(fi, hdr, 9) || !rd(fi, &s.ws, 1)) return 2;
    const int n_pairs = hdr[0];
    s.W = hdr[1]; s.H = hdr[2]; s.pyr_levels = hdr[3]; s.radius = hdr[4]; s.offset = hdr[5]; s.origin_w = hdr[6];
    s.isoutput = hdr[7];
    const int bad_pair = hdr[8];
    s.verbose = 0;
    s.lowest_w = s.W >> (s.pyr_levels - 1);
    s.lowest_h = s.H >> (s.pyr_levels - 1);
    const size_t px = (size_t)s.W * s.H;
    s.pairs.resize(n_pairs);
    for (int p = 0; p < n_pairs; p++) {
        MockPair &mp = s.pairs[p];
        if (!rd(fi, mp.Q, 16) || !rd(fi, mp.R, 9) || !rd(fi, mp.T, 3)) return 2;
        for (int v = 0; v < 2; v++) { mp.img[v].resize(px * 3); if (!rd(fi, mp.img[v].data(), px * 3)) return 2; }
        for (int v = 0; v < 2; v++) { mp.msk[v].resize(px); if (!rd(fi, mp.msk[v].data(), px)) return 2; }
        if (p == bad_pair) for (int v = 0; v < 2; v++) memset(mp.msk[v].data(), 0, px); // empty mask: degenerate margin
    }
    fclose(fi);
    RsmStereoAdapter<MockTraits> gpu(0);
    if (!gpu.Ok()) { fprintf(stderr, "%s\n", gpu.LastError()); return 3; }
    FILE *fo = fopen(argv[2], "wb");
    if (!fo) return 2;
    for (int p = 0; p < n_pairs; p++) { // the pair loop of MatchAllLayer, .cpp:17-33 (Rectify already done)
        s.cur = p;
        s.inserted.clear();
        s.filtered.clear();
        const bool ok = gpu.MatchPair(s, p);
        const int32_t okst[2] = {ok ? 1 : 0, gpu.LastStatus()};
        wr(fo, okst, 2);
        if (!ok) { fprintf(stderr, "pair %d: %s\n", p, gpu.LastError()); continue; }
        int32_t mg[12];
        for (int v = 0; v < 2; v++) {
            const rsm_boundary &b = s.pairs[p].bound[v];
            const int32_t t[6] = {b.YL, b.YR, b.XL, b.XR, b.width, b.height};
            memcpy(mg + 6 * v, t, sizeof t);
        }
        wr(fo, mg, 12);
        const int64_t n = (int64_t)s.inserted.size() / 3;
        wr(fo, &n, 1);
        const int32_t farg = s.filtered.size() == 1 ? s.filtered[0] : -1;
        wr(fo, &farg, 1);
        wr(fo, s.inserted.data(), s.inserted.size());
        wr(fo, gpu.disparity[0].data(), px);
    }
    fclose(fo);
    return 0;
}
