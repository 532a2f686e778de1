// rsm_stereo_adapter.hpp -- the marshalling between a CStereoMatching-shaped host object and the C ABI of rsm.h,
// written against a small ACCESSOR TRAITS type instead of cv::Mat / CManageData, so that it is plain C++11 (no OpenCV,
// no reference headers) and can be compiled and RUN wherever librsm_mi355.so runs: tests/cpp/mock_adapter.cpp drives
// it on the GPU box with mock types, tests/cpp/adapter_bench.cpp times it on the bench workload.
// include/CStereoMatchingMI355.hpp supplies the cv::Mat-facing traits for the reference tree (a ~60-line shim).
//
// Two entry points:
//   MatchAll(sm, n_pairs) -- the pair LOOP of CStereoMatching::MatchAllLayer (reconstruction/CStereoMatching.cpp:17-33)
//       with `pairs_in_flight` pairs on the GPU at once: a worker thread per slot uploads / runs / downloads its pair
//       (rsm_upload_pair, rsm_run_pair, rsm_download_pair into page-locked buffers from rsm_host_alloc), the calling
//       thread prepares the pairs (Traits::prepare = the reference's Rectify(CamPair, Q), .cpp:20) and replays the
//       results strictly in pair order -- `cam[pair][v].bound = margin[v]` (:27-28), the optional cloud%d.ply
//       (:707-757), InsertPoint per point in row-major pixel order (:749-751), filter(CamPair) (:31) -- exactly the
//       sequence of calls the reference makes, while the next pairs are already being matched.
//   MatchPair(sm, CamPair) -- the loop BODY (.cpp:21-31) for one pair, synchronous (Rectify stays with the caller).
//
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
#ifndef RSM_STEREO_ADAPTER_HPP
#define RSM_STEREO_ADAPTER_HPP

#include <stdio.h>
#include <string.h>

#include <condition_variable>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "rsm.h"

template <class Traits>
class RsmStereoAdapter {
public:
    typedef typename Traits::Stereo Stereo;

    // pairs_in_flight: contexts (= pairs on the GPU at once) MatchAll may use; they and their page-locked result
    // buffers (24 B per pixel each; + 16 B per pixel with want_disparity) are created on first use.
    explicit RsmStereoAdapter(int hip_device = 0, int pairs_in_flight = 3)
        : want_disparity(false), device_(hip_device), create_status_(RSM_OK), status_(RSM_OK), n_points_(0), v_top_(0),
          nslots_(pairs_in_flight < 1 ? 1 : pairs_in_flight), slots_(new Slot[(size_t)(pairs_in_flight < 1 ? 1 : pairs_in_flight)]) {
        create_status_ = rsm_create(&slots_[0].ctx, hip_device); // the first context now: "is there a GPU" is answered here
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
    }
    bool Ok() const { return create_status_ == RSM_OK; }
    int LastStatus() const { return create_status_ != RSM_OK ? create_status_ : status_; }
    const char *LastError() const {
        if (create_status_ != RSM_OK) return "rsm_create failed (no MI355X / HIP runtime?)";
        return err_.c_str();
    }
    rsm_ctx *Context() { return slots_[0].ctx; }

    bool want_disparity;            // download the fp64 disparity maps too (into `disparity`)
    std::vector<double> disparity[2]; // the LAST replayed pair's maps when want_disparity (the reference's local, .cpp:22)

    // One pair, synchronous.  A failed pair (e.g. RSM_E_DEGENERATE_MARGIN, the reference's exit(0) at .cpp:827-830)
    // returns false and leaves the adapter usable for the next pair.
    bool MatchPair(Stereo &sm, int CamPair) {
        if (create_status_ != RSM_OK) return false;
        Slot &s = slots_[0];
        if (!submit(sm, s, CamPair)) return false;
        run_slot(s);
        return replay(sm, s);
    }

    // The pair loop (.cpp:17-33) over pairs 0 .. n_pairs-1 with pairs in flight.  status (optional, n_pairs entries)
    // receives every pair's RSM_* code; a failed pair is skipped (no InsertPoint / filter for it) and the others still
    // run.  Returns the number of pairs that succeeded.
    int MatchAll(Stereo &sm, int n_pairs, int *status = 0) {
        if (create_status_ != RSM_OK) return 0;
        int ok = 0, submitted = 0, replayed = 0;
        const int S = nslots_;
        std::vector<int> slot_of((size_t)(n_pairs > 0 ? n_pairs : 0), -1);
        while (replayed < n_pairs) {
            // hand pairs to free slots, in pair order (slot = pair % S keeps the replay order trivial)
            while (submitted < n_pairs && submitted - replayed < S) {
                Slot &s = slots_[(size_t)(submitted % S)];
                const int p = submitted++;
                slot_of[(size_t)p] = -1;
                if (!ensure_slot(s)) {
                    if (status) status[p] = status_;
                    slot_of[(size_t)p] = -2; // failed before submission
                    continue;
                }
                if (!Traits::prepare(sm, p)) {
                    status_ = RSM_E_INVALID;
                    err_ = "prepare (Rectify) failed";
                    if (status) status[p] = status_;
                    slot_of[(size_t)p] = -2;
                    continue;
                }
                if (!submit(sm, s, p)) {
                    if (status) status[p] = status_;
                    slot_of[(size_t)p] = -2;
                    continue;
                }
                slot_of[(size_t)p] = p % S;
                {
                    std::lock_guard<std::mutex> g(s.mu);
                    s.state = SUBMITTED;
                }
                s.cv.notify_all();
            }
            const int p = replayed++;
            if (slot_of[(size_t)p] < 0) continue;
            Slot &s = slots_[(size_t)slot_of[(size_t)p]];
            {
                std::unique_lock<std::mutex> g(s.mu);
                while (s.state != DONE) s.cv.wait(g);
                s.state = IDLE;
            }
            const bool good = replay(sm, s);
            if (status) status[p] = status_;
            if (good) ok++;
        }
        return ok;
    }
    int64_t LastPointCount() const { return n_points_; }
    int64_t LastVTop() const { return v_top_; }

private:
    enum State { IDLE, SUBMITTED, DONE, QUIT };
    struct Slot {
        rsm_ctx *ctx;
        std::thread worker;
        std::mutex mu;
        std::condition_variable cv;
        State state;
        int pair, run_status;
        bool dump;
        rsm_pair_in in;
        rsm_pair_out out;
        size_t cap_px; // pixels the page-locked buffers are sized for
        bool have_disp, have_bgr;
        double *xyz, *disp[2];
        unsigned char *bgr;
        std::string err;
        Slot() : ctx(0), state(IDLE), pair(-1), run_status(RSM_OK), dump(false), cap_px(0), have_disp(false), have_bgr(false), xyz(0), bgr(0) {
            disp[0] = disp[1] = 0;
            memset(&in, 0, sizeof in);
            memset(&out, 0, sizeof out);
        }

    private:
        Slot(const Slot &);
        Slot &operator=(const Slot &);
    };

    static void free_buffers(Slot &s) {
        rsm_host_free(s.xyz);
        rsm_host_free(s.bgr);
        rsm_host_free(s.disp[0]);
        rsm_host_free(s.disp[1]);
        s.xyz = 0;
        s.bgr = 0;
        s.disp[0] = s.disp[1] = 0;
        s.cap_px = 0;
        s.have_disp = s.have_bgr = false;
    }
    // context + worker thread of a slot, on first use
    bool ensure_slot(Slot &s) {
        if (!s.ctx) {
            const int st = rsm_create(&s.ctx, device_);
            if (st != RSM_OK) {
                status_ = st;
                err_ = "rsm_create failed for a further pair in flight";
                return false;
            }
        }
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
                s->state = DONE;
            }
            s->cv.notify_all();
        }
    }
    // upload -> run -> download of the slot's pair (worker thread, or the caller's for MatchPair)
    static void run_slot(Slot &s) {
        s.run_status = rsm_match_pair(s.ctx, &s.in, &s.out);
        s.err = s.run_status == RSM_OK ? "" : rsm_last_error(s.ctx);
    }
    // fills the slot's rsm_pair_in / rsm_pair_out for `pair` (calling thread)
    bool submit(Stereo &sm, Slot &s, int pair) {
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
        // page-locked result buffers (one DMA each at the link's rate; pageable memory goes through a staging copy at
        // a sixth of it), kept across pairs
        if (s.cap_px != px || (want_disparity && !s.have_disp) || (s.dump && !s.have_bgr)) {
            free_buffers(s);
            s.xyz = (double *)rsm_host_alloc(px * 3 * sizeof(double));
            if (s.dump) s.bgr = (unsigned char *)rsm_host_alloc(px * 3);
            if (want_disparity)
                for (int v = 0; v < 2; v++) s.disp[v] = (double *)rsm_host_alloc(px * sizeof(double));
            if (!s.xyz || (s.dump && !s.bgr) || (want_disparity && (!s.disp[0] || !s.disp[1]))) {
                free_buffers(s);
                status_ = RSM_E_NOMEM;
                err_ = "rsm_host_alloc failed for the result buffers";
                return false;
            }
            s.cap_px = px;
            s.have_disp = want_disparity;
            s.have_bgr = s.dump;
        }
        memset(&s.out, 0, sizeof s.out);
        for (int v = 0; v < 2; v++) s.out.disparity[v] = want_disparity ? s.disp[v] : 0;
        s.out.max_points = (int64_t)px;
        s.out.xyz = s.xyz;
        s.out.bgr = s.dump ? s.bgr : 0;
        s.pair = pair;
        return true;
    }
    // what the reference does with a finished pair, on the calling thread (.cpp:27-31)
    bool replay(Stereo &sm, Slot &s) {
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
        if (s.dump) { // the in-call cloud%d.ply of DisparityToCloud (.cpp:707-730, 753-757)
            char name[64];
            snprintf(name, sizeof name, "cloud%d.ply", s.pair);
            (void)rsm_write_ply(name, s.xyz, s.bgr, out.n_points);
        }
        if (want_disparity && s.have_disp)
            for (int v = 0; v < 2; v++) disparity[v].assign(s.disp[v], s.disp[v] + s.cap_px);
        for (int64_t i = 0; i < out.n_points; i++) Traits::insert_point(sm, &s.xyz[3 * (size_t)i]); // .cpp:749-751
        Traits::filter(sm, s.pair);                                                                  // .cpp:31
        return true;
    }

    int device_;
    int create_status_, status_;
    std::string err_;
    int64_t n_points_, v_top_;
    int nslots_;
    Slot *slots_;
    RsmStereoAdapter(const RsmStereoAdapter &);
    RsmStereoAdapter &operator=(const RsmStereoAdapter &);
};

#endif
