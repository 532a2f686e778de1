// adapter_bench.cpp -- times include/rsm_stereo_adapter.hpp's MatchAll (the pair loop of CStereoMatching::MatchAllLayer,
// reconstruction/CStereoMatching.cpp:17-33) on the bench workload, PCIe included: host images in (pageable, as the
// reference's cv::Mat), InsertPoint stream out.  Mock CStereoMatching / CManageData behind the traits, no OpenCV.
// Built and run by bench.py (value_adapter_pcie_inclusive) and tests/test_gpu_cpp_adapter.py.
//   adapter_bench <in.bin> <n_pairs_total> <pairs_in_flight> [want_disparity] [flags] [n_devices]
//   flags (bit set): 1 fp64 points over PCIe (27 B per point; default: the 16-byte records); 2 MatchAllFiltered (the per-pair
//   cloud filter on the GPU inside the loop, CCloudOptimization.cpp:82-121); 4 no page-locked input staging (the images go up
//   from pageable memory through the runtime's own staging: A/B); n_devices: the first n visible GPUs, pairs_in_flight
//   slots on each (default 1 = device 0; 0 = all)
// in.bin: the format of mock_adapter.cpp; its pairs are cycled until n_pairs_total pairs have been matched.  One
// untimed MatchAll over pairs_in_flight pairs first (contexts, workspaces, page-locked buffers), then the timed one.
// stdout: one JSON line.
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <chrono>
#include <vector>

#include "rsm_stereo_adapter.hpp"

struct BPair {
    double Q[16], R[9], T[3];
    std::vector<unsigned char> img[2], msk[2];
};
struct BStereo {
    int pyr_levels, lowest_w, lowest_h, origin_w, radius, offset;
    double ws;
    int W, H, cur;
    std::vector<BPair> pairs; // the distinct inputs; pair p of the run uses pairs[p % size]
    std::vector<float> cloud; // what InsertPoint keeps: float xyz (CCloudOptimization.cpp:61); sized once, filled through `fill`
    size_t fill;
    int64_t points, filters, kept;
    std::vector<double> done_at; // when each pair's replay finished (seconds on the steady clock): the loop's steady-state rate
    void stamp() { done_at.push_back(std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count()); }
};
struct BTraits {
    typedef BStereo Stereo;
    static const BPair &pr(Stereo &s, int pair) { return s.pairs[(size_t)pair % s.pairs.size()]; }
    static int pyr_levels(Stereo &s) { return s.pyr_levels; }
    static int lowest_width(Stereo &s) { return s.lowest_w; }
    static int lowest_height(Stereo &s) { return s.lowest_h; }
    static int origin_width(Stereo &s) { return s.origin_w; }
    static int radius(Stereo &s) { return s.radius; }
    static double ws(Stereo &s) { return s.ws; }
    static int offset(Stereo &s) { return s.offset; }
    static int verbose(Stereo &) { return 0; }
    static bool isoutput(Stereo &) { return false; }
    static bool prepare(Stereo &s, int pair) { s.cur = pair; return true; }
    static bool image(Stereo &s, int pair, int v, const unsigned char *&p, int &w, int &h) { p = pr(s, pair).img[v].data(); w = s.W; h = s.H; return true; }
    static bool mask(Stereo &s, int pair, int v, const unsigned char *&p, int &w, int &h) { p = pr(s, pair).msk[v].data(); w = s.W; h = s.H; return true; }
    static double Q(Stereo &s, int i, int j) { return pr(s, s.cur).Q[4 * i + j]; }
    static double R_final(Stereo &s, int i, int j) { return pr(s, s.cur).R[3 * i + j]; }
    static double T_final(Stereo &s, int i) { return pr(s, s.cur).T[i]; }
    static void set_margin(Stereo &, int, int, const rsm_boundary &) {}
    static void insert_point(Stereo &s, const double xyz[3]) { // the cast and append of CCloudOptimization::InsertPoint
        float *d = &s.cloud[s.fill];
        d[0] = (float)xyz[0];
        d[1] = (float)xyz[1];
        d[2] = (float)xyz[2];
        s.fill += 3;
        s.points++;
    }
    static void filter(Stereo &s, int) {
        s.filters++;
        s.fill = 0; // (the reference's filter() consumes cloud_in and clears it, CCloudOptimization.cpp:84-121)
        s.stamp();
    }
    static void cam_center(Stereo &, int, float c[3]) { c[0] = c[1] = c[2] = 0.0f; }
    static void filtered_cloud(Stereo &s, int, const rsm_point16 *pts, const float *nrm, int64_t n_kept, int64_t n_raw) {
        // the reference's cloud_normal (PointNormal: xyz + normal + curvature), CCloudOptimization.cpp:110-123: copied out of the
        // page-locked buffers as `*cloud_normals += *cloud_normal` would
        float *d = &s.cloud[0];
        for (int64_t i = 0; i < n_kept; i++) {
            d[0] = pts[i].x + nrm[4 * i];
            d[1] = pts[i].y + nrm[4 * i + 1];
            d[2] = pts[i].z + nrm[4 * i + 2];
            d += 3;
        }
        s.points += n_raw;
        s.kept += n_kept;
        s.filters++;
        s.stamp();
    }
};

template <typename T>
static bool rd(FILE *f, T *p, size_t n) { return fread(p, sizeof(T), n, f) == n; }

int main(int argc, char **argv) {
    if (argc < 4) return 2;
    FILE *fi = fopen(argv[1], "rb");
    if (!fi) return 2;
    const int n_total = atoi(argv[2]), inflight = atoi(argv[3]);
    const bool want_disp = argc > 4 && atoi(argv[4]) != 0;
    const int flags = argc > 5 ? atoi(argv[5]) : 0;
    int n_dev = argc > 6 ? atoi(argv[6]) : 1;
    int32_t hdr[9];
    BStereo s;
    if (!rd(fi, hdr, 9) || !rd(fi, &s.ws, 1)) return 2;
    s.W = hdr[1]; s.H = hdr[2]; s.pyr_levels = hdr[3]; s.radius = hdr[4]; s.offset = hdr[5]; s.origin_w = hdr[6];
    s.lowest_w = s.W >> (s.pyr_levels - 1);
    s.lowest_h = s.H >> (s.pyr_levels - 1);
    const size_t px = (size_t)s.W * s.H;
    s.pairs.resize(hdr[0]);
    for (size_t p = 0; p < s.pairs.size(); p++) {
        BPair &bp = s.pairs[p];
        if (!rd(fi, bp.Q, 16) || !rd(fi, bp.R, 9) || !rd(fi, bp.T, 3)) return 2;
        for (int v = 0; v < 2; v++) { bp.img[v].resize(px * 3); if (!rd(fi, bp.img[v].data(), px * 3)) return 2; }
        for (int v = 0; v < 2; v++) { bp.msk[v].resize(px); if (!rd(fi, bp.msk[v].data(), px)) return 2; }
    }
    fclose(fi);
    std::vector<int> devs = RsmStereoAdapter<BTraits>::AllDevices();
    if (devs.empty()) { fprintf(stderr, "no GPU\n"); return 3; }
    if (n_dev > 0 && n_dev < (int)devs.size()) devs.resize((size_t)n_dev);
    n_dev = (int)devs.size();
    RsmStereoAdapter<BTraits> gpu(devs, inflight);
    if (!gpu.Ok()) { fprintf(stderr, "%s\n", gpu.LastError()); return 3; }
    gpu.want_disparity = want_disp;
    gpu.fp64_points = (flags & 1) != 0;
    gpu.stage_inputs = (flags & 4) == 0; // 4: the plain pageable upload (A/B)
    if (const char *mr = getenv("RSM_ADAPTER_MAX_RUNNING")) gpu.max_running = atoi(mr); // (A/B of the run gate; default 3)
    const bool filtered = (flags & 2) != 0;
    s.cloud.assign(px * 3, 0.0f); // a cloud can hold a point per pixel
    s.fill = 0;
    s.points = s.filters = s.kept = 0;
    { // warm-up: contexts, workspaces and page-locked buffers of every slot (the filter's arena too)
        BStereo w = s;
        const int nw = gpu.Slots();
        if ((filtered ? gpu.MatchAllFiltered(w, nw) : gpu.MatchAll(w, nw)) != nw) { fprintf(stderr, "warm-up: %s\n", gpu.LastError()); return 4; }
    }
    std::vector<int> status((size_t)n_total, 0);
    const auto t0 = std::chrono::steady_clock::now();
    const int ok = filtered ? gpu.MatchAllFiltered(s, n_total, status.data()) : gpu.MatchAll(s, n_total, status.data());
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (ok != n_total) { fprintf(stderr, "MatchAll: %d of %d pairs, %s\n", ok, n_total, gpu.LastError()); return 5; }
    // steady state: from the first pair's replay to the last one's (the job's fill -- the first pair's upload, match and download with
    // nothing to overlap -- left out): seconds per pair once the pipeline runs
    double steady = 0.0;
    if (s.done_at.size() >= 2) steady = (s.done_at.back() - s.done_at.front()) / (double)(s.done_at.size() - 1);
    printf("{\"pairs\": %d, \"pairs_in_flight\": %d, \"devices\": %d, \"seconds\": %.6f, \"points\": %lld, \"kept\": %lld, \"filters\": %lld, \"v_top_last\": %lld, "
           "\"want_disparity\": %d, \"fp64_points\": %d, \"gpu_filter\": %d, \"caller_submit_s\": %.4f, \"caller_wait_s\": %.4f, \"caller_replay_s\": %.4f, "
           "\"steady_s_per_pair\": %.6f}\n",
           n_total, inflight, n_dev, dt, (long long)s.points, (long long)s.kept, (long long)s.filters, (long long)gpu.LastVTop(), want_disp ? 1 : 0, (flags & 1) ? 1 : 0,
           filtered ? 1 : 0, gpu.LastSubmitSeconds(), gpu.LastWaitSeconds(), gpu.LastReplaySeconds(), steady);
    return 0;
}
