// mock_adapter.cpp -- drives include/rsm_stereo_adapter.hpp (the C++ glue a maintainer drops into the reference tree)
// for real: mock stand-ins of CStereoMatching / CManageData behind a traits type, no OpenCV.  Built with g++ on the
// GPU box by tests/test_gpu_cpp_adapter.py, linked against librsm_mi355.so.
//   mock_adapter <in.bin> <out.bin>
//   mock_adapter <in.bin> <out.bin> [mode] [flags]   mode 0: a MatchPair call per pair (default); N >= 1: ONE MatchAll call
//                                               with N pairs in flight per device
//   flags (bit set): 1 fp64_points (fp64 xyz + BGR cross PCIe instead of the 16-byte records); 2 MatchAllFiltered (the per-pair
//   cloud filter on the GPU: out.bin then holds the kept records + normals instead of the InsertPoint stream); 4 the adapter
//   is given the device list {0, 0} (two "GPUs", both ordinal 0: the several-devices code path on a one-GPU box); 8 the
//   mock's filter() throws at the third pair of a first MatchAll, which must leave the adapter usable for the second one
// in.bin : int32 n_pairs, W, H, levels, radius, offset, origin_w, isoutput, bad_pair; double ws;
//          per pair: double Q[16], R[9], T[3]; u8 img0[WH3], img1[WH3], mask0[WH], mask1[WH]
// out.bin: per pair: int32 ok, status; int32 margin[2][6]; int64 n_points (InsertPoint calls); int32 filter_arg;
//          double xyz[n_points*3] as handed to InsertPoint; double disparity0[WH]
//          (flags & 2: int64 n_kept, n_raw; rsm_point16 points[n_kept]; float normals[n_kept*4]; double disparity0[WH])
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <stdexcept>
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
    // what each pair's replay produced, captured when filter(pair) is called (the last call of a pair's replay)
    struct Got {
        bool seen;
        std::vector<double> xyz, disp0;
        int filter_arg;
        std::vector<rsm_point16> kept; // MatchAllFiltered
        std::vector<float> normals;
        int64_t n_raw;
        Got() : seen(false), filter_arg(-1), n_raw(0) {}
    };
    int throw_at; // filter() throws at this pair (-1: never)
    std::vector<Got> got;
    const std::vector<double> *adapter_disp0; // the adapter's `disparity[0]`
    std::vector<int> prepared;                // order of the prepare (Rectify) calls
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
    static bool prepare(Stereo &s, int pair) { // the reference's Rectify(CamPair, Q): Q / R_final / T_final become this pair's
        s.cur = pair;
        s.prepared.push_back(pair);
        return true;
    }
    static void insert_point(Stereo &s, const double xyz[3]) { s.inserted.insert(s.inserted.end(), xyz, xyz + 3); }
    static void cam_center(Stereo &, int pair, float c[3]) { c[0] = 10.0f * pair; c[1] = -5.0f; c[2] = 3.0f; }
    static void filtered_cloud(Stereo &s, int pair, const rsm_point16 *pts, const float *nrm, int64_t n_kept, int64_t n_raw) {
        s.filtered.push_back(pair);
        MockStereo::Got &g = s.got[pair];
        g.seen = true;
        g.filter_arg = pair;
        g.kept.assign(pts, pts + n_kept);
        g.normals.assign(nrm, nrm + 4 * n_kept);
        g.n_raw = n_raw;
        if (s.adapter_disp0) g.disp0 = *s.adapter_disp0;
    }
    static void filter(Stereo &s, int pair) {
        if (pair == s.throw_at) throw std::runtime_error("mock filter failed");
        s.filtered.push_back(pair);
        MockStereo::Got &g = s.got[pair];
        g.seen = true;
        g.filter_arg = pair;
        g.xyz.swap(s.inserted);
        s.inserted.clear();
        if (s.adapter_disp0) g.disp0 = *s.adapter_disp0;
    }
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
    const int mode = argc > 3 ? atoi(argv[3]) : 0;
    const int flags = argc > 4 ? atoi(argv[4]) : 0;
    std::vector<int> devs(1, 0);
    if (flags & 4) devs.push_back(0);
    RsmStereoAdapter<MockTraits> gpu(devs, mode > 0 ? mode : 1);
    if (!gpu.Ok()) { fprintf(stderr, "%s\n", gpu.LastError()); return 3; }
    if ((flags & 4) && (gpu.Slots() != 2 * (mode > 0 ? mode : 1) || gpu.DeviceOfSlot(1) != 0)) return 7;
    gpu.want_disparity = true;
    gpu.fp64_points = (flags & 1) != 0;
    s.throw_at = -1;
    s.adapter_disp0 = &gpu.disparity[0];
    s.got.resize(n_pairs);
    std::vector<int> status(n_pairs, 0);
    std::vector<int> okv(n_pairs, 0);
    if (mode > 0 && (flags & 8)) { // a callback throws in the middle of the loop: pairs are in flight at that moment
        s.throw_at = 2;
        bool thrown = false;
        try {
            gpu.MatchAll(s, n_pairs, status.data());
        } catch (const std::runtime_error &) {
            thrown = true;
        }
        if (!thrown) { fprintf(stderr, "the callback's exception did not reach the caller\n"); return 8; }
        s.throw_at = -1; // ... and the adapter must be usable again: the run below is the one that is checked
        s.prepared.clear();
        s.filtered.clear();
        s.inserted.clear();
        s.got.assign(n_pairs, MockStereo::Got());
    }
    if (mode > 0) { // the pair loop of MatchAllLayer (.cpp:17-33) in one call, pairs in flight
        const int nok = (flags & 2) ? gpu.MatchAllFiltered(s, n_pairs, status.data()) : gpu.MatchAll(s, n_pairs, status.data());
        int cnt = 0;
        for (int p = 0; p < n_pairs; p++) { okv[p] = status[p] == RSM_OK; cnt += okv[p]; }
        if (cnt != nok) { fprintf(stderr, "MatchAll returned %d, status says %d\n", nok, cnt); return 4; }
        for (int p = 0; p < n_pairs; p++)
            if ((int)s.prepared.size() != n_pairs || s.prepared[p] != p) { fprintf(stderr, "prepare order\n"); return 4; }
        int last = -1; // filter(CamPair) calls in ascending pair order, one per good pair
        for (size_t i = 0; i < s.filtered.size(); i++) { if (s.filtered[i] <= last) { fprintf(stderr, "replay order\n"); return 4; } last = s.filtered[i]; }
    } else {
        for (int p = 0; p < n_pairs; p++) { // the loop written out, a MatchPair call per pair (Rectify already done)
            MockTraits::prepare(s, p);
            okv[p] = gpu.MatchPair(s, p) ? 1 : 0;
            status[p] = gpu.LastStatus();
            if (!okv[p]) fprintf(stderr, "pair %d: %s\n", p, gpu.LastError());
        }
    }
    FILE *fo = fopen(argv[2], "wb");
    if (!fo) return 2;
    for (int p = 0; p < n_pairs; p++) {
        const int32_t okst[2] = {okv[p], status[p]};
        wr(fo, okst, 2);
        if (!okv[p]) {
            if (mode > 0) fprintf(stderr, "pair %d: status %d\n", p, status[p]);
            if (s.got[p].seen) return 5; // a failed pair must not reach InsertPoint / filter
            continue;
        }
        int32_t mg[12];
        for (int v = 0; v < 2; v++) {
            const rsm_boundary &b = s.pairs[p].bound[v];
            const int32_t t[6] = {b.YL, b.YR, b.XL, b.XR, b.width, b.height};
            memcpy(mg + 6 * v, t, sizeof t);
        }
        wr(fo, mg, 12);
        const MockStereo::Got &g = s.got[p];
        const int64_t n = (int64_t)g.xyz.size() / 3;
        wr(fo, &n, 1);
        const int32_t farg = g.seen ? g.filter_arg : -1;
        wr(fo, &farg, 1);
        wr(fo, g.xyz.data(), g.xyz.size());
        if (flags & 2) {
            const int64_t kn[2] = {(int64_t)g.kept.size(), g.n_raw};
            wr(fo, kn, 2);
            wr(fo, g.kept.data(), g.kept.size());
            wr(fo, g.normals.data(), g.normals.size());
        }
        if (g.disp0.size() != px) return 6;
        wr(fo, g.disp0.data(), px);
    }
    fclose(fo);
    return 0;
}
