// CStereoMatchingMI355.hpp -- the cv::Mat-facing side of the C++ glue: accessor traits over the reference's own
// CStereoMatching / CManageData (it is compiled INSIDE the reference tree, after SharedInclude.h and
// CStereoMatching.h) for the plain-C++ adapter of rsm_stereo_adapter.hpp, which holds all the marshalling and is
// exercised for real by tests/cpp/mock_adapter.cpp.  This repository syntax-checks this shim against the reference's
// vendored headers (tests/test_abi.py).
//
//   #include "SharedInclude.h"
//   #include "CStereoMatching.h"
//   #include "CStereoMatchingMI355.hpp"
//   ...
//   void CStereoMatching::MatchAllLayer()         // the whole loop, .cpp:15-34, three pairs in flight on GPU 0
//   {
//       static RsmStereoMI355 gpu(0, 3);          // one per process
//       // -- or, on a node with several MI355X: three pairs in flight on EACH visible GPU, pair p on GPU p % n --
//       // static RsmStereoMI355 gpu(RsmStereoMI355::AllDevices(), 3);
//       // Rectify is private (CStereoMatching.h:49-50): a lambda written inside this member function may call it
//       RsmCvTraits::rectify() = [](CStereoMatching &s, int CamPair) { s.Rectify(CamPair, s.Q); };   // .cpp:20
//       std::vector<int> status(m_data->m_CampairNum);
//       if (gpu.MatchAll(*this, m_data->m_CampairNum, status.data()) != m_data->m_CampairNum) printf("rsm: %s\n", gpu.LastError());
//   }
// or, keeping the reference's loop and replacing only its body (.cpp:21-31), one pair at a time:
//   static RsmStereoMI355 gpu(0);
//   for (int CamPair = 0; CamPair < m_data->m_CampairNum; CamPair++) {
//       Rectify(CamPair, Q);                  // unchanged reference code (.cpp:20)
//       if (!gpu.MatchPair(*this, CamPair)) printf("rsm: %s\n", gpu.LastError());
//   }
#ifndef CSTEREOMATCHING_MI355_HPP
#define CSTEREOMATCHING_MI355_HPP

#include "rsm_stereo_adapter.hpp"

struct RsmCvTraits {
    typedef CStereoMatching Stereo; // its members are public: CStereoMatching.h:38-45
    static int pyr_levels(Stereo &s) { return s.m_data->m_PyrmNum; }
    static int lowest_width(Stereo &s) { return s.m_data->m_LowestLevelSize.width; }
    static int lowest_height(Stereo &s) { return s.m_data->m_LowestLevelSize.height; }
    static int origin_width(Stereo &s) { return s.m_data->m_OriginSize.width; }
    static int radius(Stereo &s) { return s.MatchBlockRadius; }
    static double ws(Stereo &s) { return s.m_ws; }
    static int offset(Stereo &s) { return s.m_offset; }
    static int verbose(Stereo &s) { return s.Verbose; }
    static bool isoutput(Stereo &s) { return s.m_data->isoutput != 0; }
    // MatchAll's per-pair preparation = CStereoMatching::Rectify(CamPair, Q) (.cpp:20, private): the member function that
    // owns the loop installs it (see the header comment); it fills cam[pair][v].image / .mask / .P and Q, R_final, T_final
    typedef void (*RectifyFn)(Stereo &, int);
    static RectifyFn &rectify() {
        static RectifyFn fn = 0;
        return fn;
    }
    static bool prepare(Stereo &s, int pair) {
        if (!rectify()) return false;
        rectify()(s, pair);
        return !s.m_data->cam[pair][0].image.empty(); // Rectify returns silently when an image cannot be read (.cpp:147-151)
    }
    static bool mat(const cv::Mat &m, int type, const unsigned char *&p, int &w, int &h) {
        if (m.empty() || !m.isContinuous() || m.type() != type) return false;
        p = m.data;
        w = m.cols;
        h = m.rows;
        return true;
    }
    static bool image(Stereo &s, int pair, int v, const unsigned char *&p, int &w, int &h) {
        return mat(s.m_data->cam[pair][v].image, CV_8UC3, p, w, h); // what Rectify left, .cpp:154
    }
    static bool mask(Stereo &s, int pair, int v, const unsigned char *&p, int &w, int &h) {
        return mat(s.m_data->cam[pair][v].mask, CV_8UC1, p, w, h); // .cpp:156-158
    }
    static double Q(Stereo &s, int i, int j) { return s.Q.at<double>(i, j); }
    static double R_final(Stereo &s, int i, int j) { return s.R_final.at<double>(i, j); }
    static double T_final(Stereo &s, int i) { return s.T_final.at<double>(i, 0); }
    static void set_margin(Stereo &s, int pair, int v, const rsm_boundary &m) {
        Boundary &b = s.margin[v];
        b.YL = m.YL; b.YR = m.YR; b.XL = m.XL; b.XR = m.XR; b.width = m.width; b.height = m.height;
        s.m_data->cam[pair][v].bound = b; // .cpp:27-28
    }
    static void insert_point(Stereo &s, const double xyz[3]) {
#ifdef IS_PCL
        cv::Mat point(3, 1, CV_64FC1, const_cast<double *>(xyz)); // the 3x1 CV_64F the reference builds at .cpp:749
        s.m_CloudOptimization->InsertPoint(point);                // .cpp:751
#else
        (void)s; (void)xyz;
#endif
    }
    static void filter(Stereo &s, int pair) {
#ifdef IS_PCL
        s.m_CloudOptimization->filter(pair); // .cpp:31
#else
        (void)s; (void)pair;
#endif
    }
    // MatchAllFiltered (the first half of CCloudOptimization::filter on the GPU): the viewpoint the normals are turned toward,
    // cam[pair][0].CamCenter (CV_32FC1 3x1, CManageData.cpp:61-62; CCloudOptimization.cpp:51,114-121) ...
    static void cam_center(Stereo &s, int pair, float c[3]) {
        const cv::Mat &m = s.m_data->cam[pair][0].CamCenter;
        for (int i = 0; i < 3; i++) c[i] = (m.type() == CV_32FC1 && m.total() >= 3) ? m.at<float>(i) : 0.0f;
    }
    // ... and where the pair's filtered cloud goes: the reference's `cloud_normal` (CCloudOptimization.cpp:110-121) is a local
    // of filter(); a pipeline that adopts the GPU filter installs the function that takes it over from there (:123 onwards)
    typedef void (*FilteredFn)(Stereo &, int pair, const rsm_point16 *points, const float *normals4, int64_t n_kept, int64_t n_raw);
    static FilteredFn &filtered_sink() {
        static FilteredFn fn = 0;
        return fn;
    }
    static void filtered_cloud(Stereo &s, int pair, const rsm_point16 *points, const float *normals4, int64_t n_kept, int64_t n_raw) {
        if (filtered_sink()) filtered_sink()(s, pair, points, normals4, n_kept, n_raw);
    }
};

typedef RsmStereoAdapter<RsmCvTraits> RsmStereoMI355;

#endif
