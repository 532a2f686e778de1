// rsm_stereo_adapter.hpp -- the marshalling between a CStereoMatching-shaped host object and the C ABI of rsm.h,
// written against a small ACCESSOR TRAITS type instead of cv::Mat / CManageData, so that it is plain C++ (no OpenCV,
// no reference headers) and can be compiled and RUN wherever librsm_mi355.so runs: tests/cpp/mock_adapter.cpp drives
// it on the GPU box with mock types.  include/CStereoMatchingMI355.hpp supplies the cv::Mat-facing traits for the
// reference tree (a ~40-line shim).
//
// It replaces the body of the per-pair loop of CStereoMatching::MatchAllLayer (reconstruction/CStereoMatching.cpp:
// 21-31): ConstructPyrm, the PyrmNum MatchOneLayer calls, the `cam[pair][v].bound = margin[v]` stores (:27-28),
// DisparityToCloud<double> with its optional cloud%d.ply (:29, :707-757), InsertPoint per point in row-major
// pixel order (:749-751) and filter(CamPair) (:31).  Rectify (:20) stays with the caller.
//
// Traits (all static; `S` = the stereo-matching object, e.g. CStereoMatching):
//   typedef ... Stereo;
//   int    pyr_levels(S&), lowest_width(S&), lowest_height(S&), origin_width(S&);   CManageData.h:31-40
//   int    radius(S&), offset(S&), verbose(S&);  double ws(S&);                     CStereoMatching.h:40-45
//   bool   isoutput(S&);                                                            CManageData.h:33
//   bool   image(S&, int pair, int v, const unsigned char *&bgr, int &w, int &h);   rectified 8UC3, contiguous
//   bool   mask (S&, int pair, int v, const unsigned char *&m,   int &w, int &h);   rectified 8UC1, contiguous
//   double Q(S&, int i, int j), R_final(S&, int i, int j), T_final(S&, int i);      CStereoMatching.h:43
//   void   set_margin(S&, int pair, int v, const rsm_boundary&);   margin[v] and cam[pair][v].bound
//   void   insert_point(S&, const double xyz[3]);                  CCloudOptimization::InsertPoint
//   void   filter(S&, int pair);                                   CCloudOptimization::filter
#ifndef RSM_STEREO_ADAPTER_HPP
#define RSM_STEREO_ADAPTER_HPP

#include <stdio.h>

#include <vector>

#include "rsm.h"

template <class Traits>
class RsmStereoAdapter {
public:
    typedef typename Traits::Stereo Stereo;

    explicit RsmStereoAdapter(int hip_device = 0) : ctx_(0), create_status_(rsm_create(&ctx_, hip_device)), status_(RSM_OK) {}
    ~RsmStereoAdapter() { rsm_destroy(ctx_); }
    bool Ok() const { return create_status_ == RSM_OK; }
    int LastStatus() const { return create_status_ != RSM_OK ? create_status_ : status_; }
    const char *LastError() const {
        if (create_status_ != RSM_OK) return "rsm_create failed (no MI355X / HIP runtime?)";
        return local_err_ ? local_err_ : rsm_last_error(ctx_);
    }
    rsm_ctx *Context() { return ctx_; }

    // the last pair's fp64 disparity maps (the reference keeps them in a local, .cpp:22)
    std::vector<double> disparity[2];

    // One pair.  A failed pair (e.g. RSM_E_DEGENERATE_MARGIN, the reference's exit(0) at .cpp:827-830) returns false
    // and leaves the adapter usable for the next pair.
    bool MatchPair(Stereo &sm, int CamPair) {
        local_err_ = 0;
        if (create_status_ != RSM_OK) return false;
        rsm_pair_in in;
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
            if (!Traits::image(sm, CamPair, v, in.image[v], w, h) || w != in.width || h != in.height ||
                !Traits::mask(sm, CamPair, v, in.mask[v], w, h) || w != in.width || h != in.height) {
                status_ = RSM_E_INVALID; // the reference returns silently on unreadable images (.cpp:147-151)
                local_err_ = "read image error: rectified image / mask missing, not contiguous or of the wrong size";
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
        const bool dump = Traits::isoutput(sm);
        xyz_.resize(3 * px);
        if (dump) bgr_.resize(3 * px);
        rsm_pair_out out;
        for (int v = 0; v < 2; v++) {
            disparity[v].resize(px);
            out.disparity[v] = disparity[v].data();
        }
        out.max_points = (int64_t)px;
        out.xyz = xyz_.data();
        out.bgr = dump ? bgr_.data() : 0;
        status_ = rsm_match_pair(ctx_, &in, &out);
        if (status_ != RSM_OK) return false;
        for (int v = 0; v < 2; v++) Traits::set_margin(sm, CamPair, v, out.margin[v]); // .cpp:27-28
        if (dump) { // the in-call cloud%d.ply of DisparityToCloud (.cpp:707-730, 753-757)
            char name[64];
            snprintf(name, sizeof name, "cloud%d.ply", CamPair);
            (void)rsm_write_ply(name, xyz_.data(), bgr_.data(), out.n_points);
        }
        for (int64_t i = 0; i < out.n_points; i++) Traits::insert_point(sm, &xyz_[3 * (size_t)i]); // .cpp:749-751
        Traits::filter(sm, CamPair);                                                                 // .cpp:31
        n_points_ = out.n_points;
        v_top_ = out.v_top;
        return true;
    }
    int64_t LastPointCount() const { return n_points_; }
    int64_t LastVTop() const { return v_top_; }

private:
    rsm_ctx *ctx_;
    int create_status_, status_;
    const char *local_err_ = 0;
    int64_t n_points_ = 0, v_top_ = 0;
    std::vector<double> xyz_;
    std::vector<unsigned char> bgr_;
    RsmStereoAdapter(const RsmStereoAdapter &);
    RsmStereoAdapter &operator=(const RsmStereoAdapter &);
};

#endif
