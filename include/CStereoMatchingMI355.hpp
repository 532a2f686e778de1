// CStereoMatchingMI355.hpp -- header-only C++ glue between the reference's CStereoMatching object and the
// C ABI of rsm.h.  It is compiled INSIDE the reference tree (it needs the reference's own headers for
// cv::Mat / CManageData / CStereoMatching); this repository only syntax-checks it against those headers.
//
// Replaces the body of the per-pair loop of CStereoMatching::MatchAllLayer
// (reconstruction/CStereoMatching.cpp:21-31): ConstructPyrm, the PyrmNum MatchOneLayer calls, the
// `cam[pair][v].bound = margin[v]` stores, DisparityToCloud<double> and the InsertPoint / filter calls.
// Rectify (:20, OpenCV on the host) stays where it is and runs before this.
//
//   #include "SharedInclude.h"
//   #include "CStereoMatching.h"
//   #include "CStereoMatchingMI355.hpp"
//   ...
//   RsmStereoMI355 gpu(0);                    // one per process and GPU
//   for (int CamPair = 0; CamPair < m_data->m_CampairNum; CamPair++) {
//       Rectify(CamPair, Q);                  // unchanged reference code
//       if (!gpu.MatchPair(*this, CamPair)) { printf("%s\n", gpu.LastError()); return; }
//   }
#ifndef CSTEREOMATCHING_MI355_HPP
#define CSTEREOMATCHING_MI355_HPP

#include <vector>

#include "rsm.h"

class RsmStereoMI355 {
public:
    explicit RsmStereoMI355(int hip_device = 0) : ctx_(0), status_(rsm_create(&ctx_, hip_device)) {}
    ~RsmStereoMI355() { rsm_destroy(ctx_); }
    bool Ok() const { return status_ == RSM_OK; }
    const char *LastError() const { return ctx_ ? rsm_last_error(ctx_) : "rsm_create failed (no MI355X / HIP runtime?)"; }

    // Keeps the last pair's fp64 disparity maps (the reference keeps them in a local, .cpp:22).
    std::vector<double> disparity[2];

    // `sm` is the reference's CStereoMatching (its members are public: CStereoMatching.h:38-45).
    template <class StereoMatchingT>
    bool MatchPair(StereoMatchingT &sm, int CamPair) {
        if (status_ != RSM_OK) return false;
        CManageData *data = sm.m_data;
        std::vector<camera> &cam = data->cam[CamPair];
        rsm_pair_in in;
        const int top = 1 << (data->m_PyrmNum - 1);
        in.width = data->m_LowestLevelSize.width * top; // largestSize, .cpp:120
        in.height = data->m_LowestLevelSize.height * top;
        in.pyr_levels = data->m_PyrmNum;
        in.radius = sm.MatchBlockRadius;
        in.ws = sm.m_ws;
        in.offset = sm.m_offset;
        in.origin_width = data->m_OriginSize.width;
        in.verbose = sm.Verbose;
        for (int v = 0; v < 2; v++) {
            const cv::Mat &img = cam[v].image, &msk = cam[v].mask; // what Rectify left (.cpp:154-158)
            if (img.empty() || msk.empty() || !img.isContinuous() || !msk.isContinuous() || img.type() != CV_8UC3 ||
                msk.type() != CV_8UC1 || img.cols != in.width || img.rows != in.height)
                return false; // the reference returns silently on unreadable images (.cpp:147-151)
            in.image[v] = img.data;
            in.mask[v] = msk.data;
        }
        for (int i = 0; i < 4; i++)
            for (int j = 0; j < 4; j++) in.Q[4 * i + j] = sm.Q.template at<double>(i, j);
        for (int i = 0; i < 3; i++) {
            for (int j = 0; j < 3; j++) in.R_final[3 * i + j] = sm.R_final.template at<double>(i, j);
            in.T_final[i] = sm.T_final.template at<double>(i, 0);
        }
        const size_t px = (size_t)in.width * in.height;
        xyz_.resize(3 * px);
        rsm_pair_out out;
        for (int v = 0; v < 2; v++) {
            disparity[v].resize(px);
            out.disparity[v] = disparity[v].data();
        }
        out.max_points = (int64_t)px;
        out.xyz = xyz_.data();
        out.bgr = 0;
        status_ = rsm_match_pair(ctx_, &in, &out);
        if (status_ != RSM_OK) return false; // RSM_E_DEGENERATE_MARGIN is the reference's exit(0), .cpp:827-830
        for (int v = 0; v < 2; v++) {
            Boundary &b = sm.margin[v];
            b.YL = out.margin[v].YL; b.YR = out.margin[v].YR; b.XL = out.margin[v].XL; b.XR = out.margin[v].XR;
            b.width = out.margin[v].width; b.height = out.margin[v].height;
            cam[v].bound = b; // .cpp:27-28
        }
#ifdef IS_PCL
        for (int64_t i = 0; i < out.n_points; i++) {
            cv::Mat point(3, 1, CV_64FC1, &xyz_[3 * i]); // same 3x1 CV_64F the reference builds at .cpp:749
            sm.m_CloudOptimization->InsertPoint(point);  // .cpp:751, row-major pixel order
        }
        sm.m_CloudOptimization->filter(CamPair); // .cpp:31
#endif
        return true;
    }

private:
    rsm_ctx *ctx_;
    int status_;
    std::vector<double> xyz_;
    RsmStereoMI355(const RsmStereoMI355 &);
    RsmStereoMI355 &operator=(const RsmStereoMI355 &);
};

#endif
