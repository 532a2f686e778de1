// rsm_stereo_adapter.hpp -- the marshalling between a CStereoMatching-shaped host object and the C ABI of rsm.h,
// written against a small ACCESSOR TRAITS type instead of cv::Mat / CManageData, so that it is plain C++11 (no OpenCV,
// no reference headers) and can be compiled and RUN wherever librsm_mi355.so runs: tests/cpp/mock_adapter.cpp drives
// it on the GPU box with mock types, tests/cpp/adapter_bench.cpp times it on the bench workload.
// include/CStereoMatchingMI355.hpp supplies the cv::Mat-facing traits for the reference tree (a ~60-line shim).
//
// Entry points:
//   MatchAll(sm, n_pairs) -- the pair LOOP of CStereoMatching::MatchAllLayer (reconstruction/CStereoMatching.cpp:17-33)
//       with pairs in flight on ONE OR SEVERAL GPUs of the node: the adapter owns `pairs_in_flight` slots per device it
//       was given (slot i lives on devices[i % n_devices], so consecutive pairs go to different GPUs first: the
//       `pair % n_gpus` sharding of SURVEY 8(e)); a worker thread per slot uploads / runs / downloads its pair
//       (rsm_upload_pair, rsm_run_pair, rsm_download_pair into page-locked buffers from rsm_host_alloc), the calling
//       thread prepares the pairs (Traits::prepare = the reference's Rectify(CamPair, Q), .cpp:20) and replays the
//       results strictly in pair order -- `cam[pair][v].bound = margin[v]` (:27-28), the optional cloud%d.ply
//       (:707-757), InsertPoint per point in row-major pixel order (:749-751), filter(CamPair) (:31) -- exactly the
//       sequence of calls the reference makes, while the next pairs are already being matched.  Nothing changes for the
//       caller between one GPU and eight:   RsmStereoMI355 gpu(RsmStereoMI355::AllDevices(), 3);
//   MatchAllFiltered(sm, n_pairs) -- the same loop with the first half of CCloudOptimization::filter
//       (CloudOptimization/CCloudOptimization.cpp:82-121: StatisticalOutlierRemoval, radius normals, the turn toward
//       CamCenter) done on the pair's GPU right after the match, while the other slots' pairs are being matched: instead
//       of InsertPoint per point + filter(CamPair) the pipeline receives Traits::filtered_cloud(sm, pair, points, normals,
//       n_kept, n_raw) = the reference's `cloud_normal` (:110-121) -- and 13 % fewer, 32-byte records cross PCIe.
//   MatchPair(sm, CamPair) -- the loop BODY (.cpp:21-31) for one pair, synchronous (Rectify stays with the caller).
//
// What crosses PCIe per point: by default the 16-byte record rsm_point16 -- float xyz, which is what
// CCloudOptimization::InsertPoint keeps of the fp64 point (CCloudOptimization.cpp:61: pcl::PointXYZ(p[0], p[1], p[2]))
// and what the cloud%d.ply holds (.cpp:754), + BGR -- packed on the GPU; insert_point receives (double)float, so the
// float a PCL-side InsertPoint stores is bit-identical to the one it would make of the fp64 point.  `fp64_points = true`
// downloads the fp64 points themselves (24 + 3 bytes per point) for a pipeline whose InsertPoint keeps doubles.
// The fp64 disparity maps (the reference's local `cv::Mat disparity[2]`, .cpp:22, never read after :29) are
// downloaded only on request (want_disparity): 2 x 100 MB per 12-MP pair that nothing consumes.
//
// Traits (all static; `S` = the stereo-matching object, e.g. CStereoMatching):
//   typedef ... Stereo;
//   int    pyr_levels(S&), lowest_width(S&), lowest_height(S&), origin_width(S&);   CManageData.h:31-40
//   int    radius(S&), offset(S&), verbose(S&);  double ws(S&);                     CStereoMatching.h:40-45
//   bool   isoutput(S&);                                                            CManageData.h:33
//   bool   prepare(S&, int pair);    MatchAll only: makes the pair's inputs current (Rectify(CamPair, Q), .cpp:20);
//                                    called on the calling thread, in pair order, before image / mask / Q / R / T are read
//   bool   image(S&, int pair, int v, const unsigned char *&bgr, int &w, int &h);   rectified 8UC3, contiguous; must stay
//   bool   mask (S&, int pair, int v, const unsigned char *&m,   int &w, int &h);   valid until the pair is replayed
//   double Q(S&, int i, int j), R_final(S&, int i, int j), T_final(S&, int i);      CStereoMatching.h:43 (copied at submit)
//   void   set_margin(S&, int pair, int v, const rsm_boundary&);   margin[v] and cam[pair][v].bound
//   void   insert_point(S&, const double xyz[3]);                  CCloudOptimization::InsertPoint
//   void   filter(S&, int pair);                                   CCloudOptimization::filter
// MatchAllFiltered only:
//   void   cam_center(S&, int pair, float c[3]);                   CamCenter[pair] (CCloudOptimization.cpp:114), read at submit
//   void   filtered_cloud(S&, int pair, const rsm_point16 *points, const float *normals4, int64_t n_kept, int64_t n_raw);
// Callbacks may throw: MatchAll waits for the pairs in flight, leaves every slot idle and rethrows.
#ifndef RSM_STEREO_ADAPTER_HPP
#define RSM_STEREO_ADAPTER_HPP

#include <stdio.h>
#include <string.h>

#include <chrono>
#include <condition_variable>
#include <mutex>
#include <string>
#include <thread>
#include <type_traits>
#include <vector>

#include "rsm.h"

template <class Traits>
class RsmStereoAdapter {
public:
    typedef typename Traits::Stereo Stereo;

    // pairs_in_flight: contexts (= pairs on the GPU at once) MatchAll may use; they and their page-locked result
    // buffers (16 B per pixel each; + 16 B per pixel with want_disparity) are created on first use.
    explicit RsmStereoAdapter(int hip_device = 0, int pairs_in_flight = 3) { init(std::vector<int>(1, hip_device), pairs_in_flight); }
    // Several GPUs of the node: pairs_in_flight_per_device slots on each of `devices` (HIP ordinals; a device may be listed
    // more than once).  An empty list means device 0.
    RsmStereoAdapter(const std::vector<int> &devices, int pairs_in_flight_per_device) { init(devices, pairs_in_flight_per_device); }
    // every GPU this process sees: {0, 1, ..., rsm_device_count() - 1}
    static std::vector<int> AllDevices() {
        std::vector<int> d;
        const int n = rsm_device_count();
        for (int i = 0; i < n; i++) d.push_back(i);
        return d;
    }
    ~RsmStereoAdapter() {
        for (int i = 0; i < nslots_; i++) {
            Slot &s = slots_[i];
            if (s.worker.joinable()) {
                {
                    std::lock_guard<std::mutex> g(s.mu);
                    s.state = QUIT;
                }
                s.cv.notify_all();
                s.worker.join();
            }
            free_buffers(s);
            rsm_destroy(s.ctx);
        }
        delete[] slots_;
        delete[] gates_;
    }
    bool Ok() const { return create_status_ == RSM_OK; }
    int LastStatus() const { return create_status_ != RSM_OK ? create_status_ : status_; }
    const char *LastError() const {
        if (create_status_ != RSM_OK) return "rsm_create failed (no MI355X / HIP runtime?)";
        return err_.c_str();
    }
    rsm_ctx *Context() { return slots_[0].ctx; }
    int Slots() const { return nslots_; }
    int DeviceOfSlot(int i) const { return devices_[(size_t)i % devices_.size()]; }

    bool want_disparity;              // download the fp64 disparity maps too (into `disparity`)
    std::vector<double> disparity[2]; // the LAST replayed pair's maps when want_disparity (the reference's local, .cpp:22)
    bool fp64_points;                 // download fp64 xyz + BGR (27 B per point) instead of the 16-byte records
    int max_running;                  // slots of one GPU inside rsm_run_pair at once (0 = no limit, the default): measured on C2 with 4 ... 7 slots
                                      // and a limit of 3 -- 272-280 Mdisp/s with and without it (profiles/r05_gate.log): the loop is not
                                      // bound by too many pairs matching at once; kept for rigs whose pairs are smaller
    bool stage_inputs;                // (default) a slot's worker copies the pair's images / masks into page-locked staging of its own before
                                      // the upload: the reference's cv::Mats are pageable, and the runtime's own staging of pageable uploads is
                                      // serialised across threads -- five workers copying in parallel keep the DMA at the link's rate
    rsm_filter_params filter_params;  // MatchAllFiltered: CReconstruction.cpp:18's values (100, 1, 2.5); cam_center comes from the traits

    // One pair, synchronous.  A failed pair (e.g. RSM_E_DEGENERATE_MARGIN, the reference's exit(0) at .cpp:827-830)
    // returns false and leaves the adapter usable for the next pair.
    bool MatchPair(Stereo &sm, int CamPair) {
        if (create_status_ != RSM_OK) return false;
        Slot &s = slots_[0];
        if (!submit(sm, s, CamPair, false)) return false;
        run_slot(s);
        return replay(sm, s, std::false_type());
    }

    // The pair loop (.cpp:17-33) over pairs 0 .. n_pairs-1 with pairs in flight.  status (optional, n_pairs entries)
    // receives every pair's RSM_* code; a failed pair is skipped (no InsertPoint / filter for it) and the others still
    // run.  Returns the number of pairs that succeeded.
    int MatchAll(Stereo &sm, int n_pairs, int *status = 0) { return match_all(sm, n_pairs, status, std::false_type()); }
    // ... with the per-pair cloud filter on the GPU (see the header comment)
    int MatchAllFiltered(Stereo &sm, int n_pairs, int *status = 0) { return match_all(sm, n_pairs, status, std::true_type()); }
    int64_t LastPointCount() const { return n_points_; }
    int64_t LastVTop() const { return v_top_; }
    // Where the calling thread spent the last MatchAll[Filtered] (seconds): preparing + submitting pairs (Traits::prepare =
    // Rectify, the marshalling), blocked until the next pair in order had finished, replaying results into the pipeline
    // (set_margin, InsertPoint per point, filter / filtered_cloud).  The loop runs at max(GPU time per pair, submit + replay per
    // pair): when `wait` is near zero the host side of the PIPELINE bounds it, not the GPU or PCIe.
    double LastSubmitSeconds() const { return t_submit_; }
    double LastWaitSeconds() const { return t_wait_; }
    double LastReplaySeconds() const { return t_replay_; }

private:
    enum State { IDLE, SUBMITTED, DONE, QUIT };
    // At most `limit` slots of a device inside rsm_run_pair at once (0 = no limit; option max_running): lets the loop have more
    // slots than pairs worth matching side by side, the extra ones overlapping their upload, staging copy and download.
    struct RunGate {
        std::mutex mu;
        std::condition_variable cv;
        int running, limit;
        RunGate() : running(0), limit(0) {}
        struct Pass {
            RunGate *g;
            explicit Pass(RunGate *gate) : g(gate) {
                if (!g || g->limit <= 0) { g = 0; return; }
                std::unique_lock<std::mutex> l(g->mu);
                while (g->running >= g->limit) g->cv.wait(l);
                g->running++;
            }
            ~Pass() {
                if (!g) return;
                { std::lock_guard<std::mutex> l(g->mu); g->running--; }
                g->cv.notify_one();
            }
        };
    };
    struct Slot {
        rsm_ctx *ctx;
        int device;
        RunGate *gate;
        std::thread worker;
        std::mutex mu;
        std::condition_variable cv;
        State state;
        int pair, run_status;
        bool dump, filtered, busy;
        rsm_pair_in in;
        rsm_pair_out out;
        rsm_filter_params fprm;
        int64_t n_kept;
        size_t cap_px; // pixels the page-locked buffers are sized for
        bool have_disp, have_fp64, have_nrm;
        rsm_point16 *pts;
        float *nrm;
        unsigned char *stage; // page-locked copy of the pair's inputs (2 images + 2 masks), or 0
        bool staged;
        double *xyz, *disp[2];
        unsigned char *bgr;
        std::string err;
        Slot() : ctx(0), device(0), gate(0), state(IDLE), pair(-1), run_status(RSM_OK), dump(false), filtered(false), busy(false), n_kept(0), cap_px(0),
                 have_disp(false), have_fp64(false), have_nrm(false), pts(0), nrm(0), stage(0), staged(false), xyz(0), bgr(0) {
            disp[0] = disp[1] = 0;
            memset(&in, 0, sizeof in);
            memset(&out, 0, sizeof out);
            memset(&fprm, 0, sizeof fprm);
        }

    private:
        Slot(const Slot &);
        Slot &operator=(const Slot &);
    };

    void init(const std::vector<int> &devices, int per_device) {
        want_disparity = false;
        fp64_points = false;
        stage_inputs = true;
        memset(&filter_params, 0, sizeof filter_params);
        filter_params.sor_mean_k = 100; // CReconstruction.cpp:18
        filter_params.sor_std_mul = 1.0;
        filter_params.normal_radius = 2.5;
        devices_ = devices.empty() ? std::vector<int>(1, 0) : devices;
        create_status_ = status_ = RSM_OK;
        n_points_ = v_top_ = 0;
        t_submit_ = t_wait_ = t_replay_ = 0.0;
        const int per = per_device < 1 ? 1 : per_device;
        nslots_ = per * (int)devices_.size();
        slots_ = new Slot[(size_t)nslots_];
        max_running = 0;
        gates_ = new RunGate[devices_.size()];
        for (int i = 0; i < nslots_; i++) {
            slots_[i].device = devices_[(size_t)i % devices_.size()];
            slots_[i].gate = &gates_[(size_t)i % devices_.size()];
        }
        create_status_ = rsm_create(&slots_[0].ctx, slots_[0].device); // the first context now: "is there a GPU" is answered here
    }
    static void free_buffers(Slot &s) {
        rsm_host_free(s.pts);
        rsm_host_free(s.nrm);
        rsm_host_free(s.stage);
        s.stage = 0;
        rsm_host_free(s.xyz);
        rsm_host_free(s.bgr);
        rsm_host_free(s.disp[0]);
        rsm_host_free(s.disp[1]);
        s.pts = 0;
        s.nrm = 0;
        s.xyz = 0;
        s.bgr = 0;
        s.disp[0] = s.disp[1] = 0;
        s.cap_px = 0;
        s.have_disp = s.have_fp64 = s.have_nrm = false;
    }
    // context + worker thread of a slot, on first use
    bool ensure_slot(Slot &s) {
        if (!s.ctx) {
            const int st = rsm_create(&s.ctx, s.device);
            if (st != RSM_OK) {
                status_ = st;
                err_ = "rsm_create failed for a further pair in flight";
                return false;
            }
        }
        if (nslots_ > (int)devices_.size()) (void)rsm_set_option(s.ctx, "shared_gpu", 1); // several slots per GPU: pairs in flight
        if (!s.worker.joinable()) s.worker = std::thread(&RsmStereoAdapter::worker_loop, this, &s);
        return true;
    }
    void worker_loop(Slot *s) {
        for (;;) {
            {
                std::unique_lock<std::mutex> g(s->mu);
                while (s->state != SUBMITTED && s->state != QUIT) s->cv.wait(g);
                if (s->state == QUIT) return;
            }
            run_slot(*s);
            {
                std::lock_guard<std::mutex> g(s->mu);
                if (s->state == QUIT) return; // the adapter is going away: its QUIT must not be overwritten
                s->state = DONE;
            }
            s->cv.notify_all();
        }
    }
    // upload -> run -> download of the slot's pair (worker thread, or the caller's for MatchPair)
    static void run_slot(Slot &s) {
        if (s.staged) { // pageable -> page-locked on this worker's core; the upload below is then one DMA per buffer
            const size_t px = (size_t)s.in.width * s.in.height;
            unsigned char *d = s.stage;
            for (int v = 0; v < 2; v++) {
                memcpy(d, s.in.image[v], px * 3);
                s.in.image[v] = d;
                d += px * 3;
            }
            for (int v = 0; v < 2; v++) {
                memcpy(d, s.in.mask[v], px);
                s.in.mask[v] = d;
                d += px;
            }
        }
        if (!s.filtered) { // rsm_match_pair's three steps, the middle one behind the device's gate (max_running)
            int st = rsm_upload_pair(s.ctx, &s.in);
            if (st == RSM_OK) {
                typename RunGate::Pass pass(s.gate);
                st = rsm_run_pair(s.ctx);
            }
            if (st == RSM_OK) st = rsm_download_pair(s.ctx, &s.out);
            s.run_status = st;
        } else { // ... with the per-pair cloud filter in between: only what survives it (and its normals) comes down
            int st = rsm_upload_pair(s.ctx, &s.in);
            if (st == RSM_OK) {
                typename RunGate::Pass pass(s.gate);
                st = rsm_run_pair(s.ctx);
            }
            if (st == RSM_OK) st = rsm_download_pair(s.ctx, &s.out); // margins, counts, the optional maps / raw records
            s.n_kept = 0;
            if (st == RSM_OK) st = rsm_filter_last_cloud_host(s.ctx, &s.fprm, s.pts, s.nrm, (int64_t)s.cap_px, &s.n_kept, 0);
            s.run_status = st;
        }
        s.err = s.run_status == RSM_OK ? "" : rsm_last_error(s.ctx);
    }
    // fills the slot's rsm_pair_in / rsm_pair_out for `pair` (calling thread)
    bool submit(Stereo &sm, Slot &s, int pair, bool filtered) {
        rsm_pair_in &in = s.in;
        const int top = 1 << (Traits::pyr_levels(sm) - 1);
        in.width = Traits::lowest_width(sm) * top; // largestSize, .cpp:120
        in.height = Traits::lowest_height(sm) * top;
        in.pyr_levels = Traits::pyr_levels(sm);
        in.radius = Traits::radius(sm);
        in.ws = Traits::ws(sm);
        in.offset = Traits::offset(sm);
        in.origin_width = Traits::origin_width(sm);
        in.verbose = Traits::verbose(sm);
        for (int v = 0; v < 2; v++) {
            int w = 0, h = 0;
            if (!Traits::image(sm, pair, v, in.image[v], w, h) || w != in.width || h != in.height ||
                !Traits::mask(sm, pair, v, in.mask[v], w, h) || w != in.width || h != in.height) {
                status_ = RSM_E_INVALID; // the reference returns silently on unreadable images (.cpp:147-151)
                err_ = "read image error: rectified image / mask missing, not contiguous or of the wrong size";
                return false;
            }
        }
        for (int i = 0; i < 4; i++)
            for (int j = 0; j < 4; j++) in.Q[4 * i + j] = Traits::Q(sm, i, j);
        for (int i = 0; i < 3; i++) {
            for (int j = 0; j < 3; j++) in.R_final[3 * i + j] = Traits::R_final(sm, i, j);
            in.T_final[i] = Traits::T_final(sm, i);
        }
        const size_t px = (size_t)in.width * in.height;
        s.dump = Traits::isoutput(sm);
        s.filtered = filtered;
        const bool fp64 = fp64_points && !filtered;
        // page-locked result buffers (one DMA each at the link's rate; pageable memory goes through a staging copy at
        // a sixth of it), kept across pairs
        if (s.cap_px != px || (want_disparity && !s.have_disp) || (fp64 && !s.have_fp64) || (filtered && !s.have_nrm)) {
            const bool keep_fp64 = fp64 || s.have_fp64, keep_nrm = filtered || s.have_nrm;
            free_buffers(s);
            s.pts = (rsm_point16 *)rsm_host_alloc(px * sizeof(rsm_point16));
            s.stage = stage_inputs ? (unsigned char *)rsm_host_alloc(px * 8) : 0; // (no staging memory: the plain pageable upload)
            if (keep_nrm) s.nrm = (float *)rsm_host_alloc(px * 4 * sizeof(float));
            if (keep_fp64) {
                s.xyz = (double *)rsm_host_alloc(px * 3 * sizeof(double));
                s.bgr = (unsigned char *)rsm_host_alloc(px * 3);
            }
            if (want_disparity)
                for (int v = 0; v < 2; v++) s.disp[v] = (double *)rsm_host_alloc(px * sizeof(double));
            if (!s.pts || (keep_nrm && !s.nrm) || (keep_fp64 && (!s.xyz || !s.bgr)) || (want_disparity && (!s.disp[0] || !s.disp[1]))) {
                free_buffers(s);
                status_ = RSM_E_NOMEM;
                err_ = "rsm_host_alloc failed for the result buffers";
                return false;
            }
            s.cap_px = px;
            s.have_disp = want_disparity;
            s.have_fp64 = keep_fp64;
            s.have_nrm = keep_nrm;
        }
        s.staged = stage_inputs && s.stage != 0;
        if (s.gate) {
            std::lock_guard<std::mutex> l(s.gate->mu);
            s.gate->limit = max_running;
        }
        memset(&s.out, 0, sizeof s.out);
        for (int v = 0; v < 2; v++) s.out.disparity[v] = want_disparity ? s.disp[v] : 0;
        if (filtered) { // the raw cloud stays on the GPU; only counts and margins come down before the filter
            s.out.max_points = 0;
            s.fprm = filter_params;
        } else {
            s.out.max_points = (int64_t)px;
            s.out.xyz = fp64 ? s.xyz : 0;
            s.out.bgr = fp64 ? s.bgr : 0;
            s.out.points16 = fp64 ? 0 : s.pts;
        }
        s.pair = pair;
        return true;
    }
    // filtered mode: the pair's CamCenter travels with it (a separate step so that Traits without cam_center still compile)
    void submit_extra(Stereo &sm, Slot &s, int pair, std::true_type) { Traits::cam_center(sm, pair, s.fprm.cam_center); }
    void submit_extra(Stereo &, Slot &, int, std::false_type) {}

    // what the reference does with a finished pair, on the calling thread (.cpp:27-31)
    bool replay(Stereo &sm, Slot &s, std::false_type) {
        if (!replay_common(sm, s)) return false;
        const rsm_pair_out &out = s.out;
        if (s.out.points16) {
            if (s.dump) { // the in-call cloud%d.ply of DisparityToCloud (.cpp:707-730, 753-757): float xyz + BGR = the record
                char name[64];
                snprintf(name, sizeof name, "cloud%d.ply", s.pair);
                (void)rsm_write_ply16(name, s.pts, out.n_points);
            }
            for (int64_t i = 0; i < out.n_points; i++) { // .cpp:749-751; InsertPoint's own cast (CCloudOptimization.cpp:61) already done
                const rsm_point16 &r = s.pts[(size_t)i];
                const double p[3] = {(double)r.x, (double)r.y, (double)r.z};
                Traits::insert_point(sm, p);
            }
        } else {
            if (s.dump) {
                char name[64];
                snprintf(name, sizeof name, "cloud%d.ply", s.pair);
                (void)rsm_write_ply(name, s.xyz, s.bgr, out.n_points);
            }
            for (int64_t i = 0; i < out.n_points; i++) Traits::insert_point(sm, &s.xyz[3 * (size_t)i]); // .cpp:749-751
        }
        Traits::filter(sm, s.pair); // .cpp:31
        return true;
    }
    bool replay(Stereo &sm, Slot &s, std::true_type) {
        if (!replay_common(sm, s)) return false;
        Traits::filtered_cloud(sm, s.pair, s.pts, s.nrm, s.n_kept, s.out.n_points); // InsertPoint x n + CCloudOptimization.cpp:82-121
        return true;
    }
    bool replay_common(Stereo &sm, Slot &s) {
        status_ = s.run_status;
        if (status_ != RSM_OK) {
            err_ = s.err;
            return false;
        }
        err_.clear();
        const rsm_pair_out &out = s.out;
        n_points_ = out.n_points; // (current for the callbacks below)
        v_top_ = out.v_top;
        for (int v = 0; v < 2; v++) Traits::set_margin(sm, s.pair, v, out.margin[v]); // .cpp:27-28
        if (want_disparity && s.have_disp)
            for (int v = 0; v < 2; v++) disparity[v].assign(s.disp[v], s.disp[v] + s.cap_px);
        return true;
    }

    // waits for every pair in flight and leaves all slots idle (a callback threw: the slots' buffers must not be
    // rewritten under a running worker by a later call)
    void drain() {
        for (int i = 0; i < nslots_; i++) {
            Slot &s = slots_[i];
            if (!s.busy) continue;
            std::unique_lock<std::mutex> g(s.mu);
            while (s.state == SUBMITTED) s.cv.wait(g);
            if (s.state == DONE) s.state = IDLE;
            s.busy = false;
        }
    }

    template <class Filtered>
    int match_all(Stereo &sm, int n_pairs, int *status, Filtered filtered) {
        if (create_status_ != RSM_OK) return 0;
        int ok = 0, submitted = 0, replayed = 0;
        const int S = nslots_;
        typedef std::chrono::steady_clock Clock;
        t_submit_ = t_wait_ = t_replay_ = 0.0;
        Clock::time_point t0 = Clock::now(), t1;
#define RSM_LAP(acc)                                       \
    t1 = Clock::now();                                     \
    acc += std::chrono::duration<double>(t1 - t0).count(); \
    t0 = t1;
        std::vector<int> slot_of((size_t)(n_pairs > 0 ? n_pairs : 0), -1);
        try {
            while (replayed < n_pairs) {
                // hand pairs to free slots, in pair order (slot = pair % S keeps the replay order trivial; with several
                // devices consecutive pairs land on different GPUs)
                while (submitted < n_pairs && submitted - replayed < S) {
                    Slot &s = slots_[(size_t)(submitted % S)];
                    const int p = submitted++;
                    slot_of[(size_t)p] = -2; // failed before submission (until proven otherwise)
                    if (!ensure_slot(s)) {
                        if (status) status[p] = status_;
                        continue;
                    }
                    if (!Traits::prepare(sm, p)) {
                        status_ = RSM_E_INVALID;
                        err_ = "prepare (Rectify) failed";
                        if (status) status[p] = status_;
                        continue;
                    }
                    if (!submit(sm, s, p, Filtered::value)) {
                        if (status) status[p] = status_;
                        continue;
                    }
                    submit_extra(sm, s, p, filtered);
                    slot_of[(size_t)p] = p % S;
                    {
                        std::lock_guard<std::mutex> g(s.mu);
                        s.state = SUBMITTED;
                        s.busy = true;
                    }
                    s.cv.notify_all();
                }
                RSM_LAP(t_submit_)
                const int p = replayed++;
                if (slot_of[(size_t)p] < 0) continue;
                Slot &s = slots_[(size_t)slot_of[(size_t)p]];
                {
                    std::unique_lock<std::mutex> g(s.mu);
                    while (s.state != DONE) s.cv.wait(g);
                    s.state = IDLE;
                    s.busy = false;
                }
                RSM_LAP(t_wait_)
                const bool good = replay(sm, s, filtered);
                RSM_LAP(t_replay_)
                if (status) status[p] = status_;
                if (good) ok++;
            }
#undef RSM_LAP
        } catch (...) {
            drain();
            throw;
        }
        return ok;
    }

    std::vector<int> devices_;
    int create_status_, status_;
    std::string err_;
    int64_t n_points_, v_top_;
    double t_submit_, t_wait_, t_replay_;
    int nslots_;
    Slot *slots_;
    RunGate *gates_;
    RsmStereoAdapter(const RsmStereoAdapter &);
    RsmStereoAdapter &operator=(const RsmStereoAdapter &);
};

#endif
