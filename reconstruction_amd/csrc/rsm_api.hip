// rsm_api.hip -- host side of librsm_mi355.so: context, per-pair pipeline and the C ABI of include/rsm.h.
//
// Pipeline = CStereoMatching::MatchAllLayer loop body (reconstruction/CStereoMatching.cpp:21-29) from the
// rectified top-level images: ConstructPyrm (:1040-1053) -> MatchOneLayer x PyrmNum (:36-113, stage order
// kept exactly) -> DisparityToCloud<double> (:682-761).  Everything stays resident in HBM between the one
// upload and the one download; both matching directions run in the same launches (gridDim.z).
#include "../../include/rsm.h"
#include "rectify_host.h"
#include "rsm_dev.h"

#include <algorithm>

#include <limits.h>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

// ------------------------------------------------------------------------------------------------
enum Stage {
    ST_PYRAMID = 0,
    ST_MARGIN,
    ST_BOXSUM,
    ST_INITIAL_MATCH,
    ST_SMOOTH,
    ST_ORDER,
    ST_UNIQ16,
    ST_REMATCH,
    ST_MEDIAN,
    ST_REFINE_INIT,
    ST_REFINE_SWEEP,     // all levels below the top
    ST_REFINE_SWEEP_TOP, // the top level's sweeps (light + worklist kernels)
    ST_REFINE_LIGHT_TOP, // only the k_refine_sweep<1> launches of the top level
    ST_REFINE_SKEW_TOP,  // only the k_refine_skew<T,1> launches of the top level (T sweeps each; the dominant kernel)
    ST_UNIQ64,
    ST_CLOUD,
    ST_COUNT
};
static const char *kStageNames[ST_COUNT] = {"pyramid", "margin", "boxsum", "initial_match", "smooth", "order",
                                            "uniqueness_s16", "rematch", "median", "refine_init",
                                            "refine_sweep", "refine_sweep_top", "refine_light_top", "refine_skew_top", "uniqueness_f64", "cloud"};

struct EvPair {
    hipEvent_t a, b;
    int stage;
};

struct rsm_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t stream2 = nullptr; // side stream: per-level BGRX copies and window sums (they depend on the images only)
    hipEvent_t ev_pyr = nullptr, ev_prep[RSM_MAX_LEVELS]{};
    std::string err;

    // resident pair
    bool have_pair = false, have_result = false;
    rsm_pair_in in{};
    int N = 0;
    int Wk[RSM_MAX_LEVELS]{}, Hk[RSM_MAX_LEVELS]{};
    uint8_t *img[RSM_MAX_LEVELS][2]{}, *msk[RSM_MAX_LEVELS][2]{};
    Mg mg[RSM_MAX_LEVELS][2]{};
    int h_margin_init[RSM_MAX_LEVELS * 2 * 4]{}; // host source of the margins' initial values (async copy)

    // workspace (sized for the top level)
    size_t cap_px = 0;
    std::vector<void *> allocs;
    int *d_margins = nullptr; // N*2*4 ints
    int32_t *S1[RSM_MAX_LEVELS][2]{}, *S2[RSM_MAX_LEVELS][2]{}, *tmp1 = nullptr, *tmp2 = nullptr; // per level and view
    uint32_t *img4[RSM_MAX_LEVELS][2]{};
    int16_t *d16a[2]{}, *BL[2]{}, *BR[2]{}; // d16a: scratch (Rectify's mask temp)
    uint8_t *cloud_flags = nullptr;         // k_cloud's per-pixel "emitted a point" flags of the last run: rsm_filter_last_cloud's lattice reads them
                                            // long after rsm_run_pair has returned, so they have a buffer nothing else borrows
    int16_t *d16i[RSM_MAX_LEVELS][2]{}, *d16s[RSM_MAX_LEVELS][2]{}, *d16m[RSM_MAX_LEVELS][2]{}; // per level: initial-match / constraint-stage / median maps, pre-filled NOMATCH
    double *f64[3][2]{};
    int32_t *nv[2]{};
    uint32_t *rf_key[2]{};
    int32_t *rf_cnt = nullptr; // NCC wide-pixel counter
    int32_t *wrow = nullptr;   // NCC: wide pixels per (direction, row) of the level at hand
    uint32_t *rf_list = nullptr;
    uint32_t *tie_list = nullptr; // NCC tie pixels (k_ncc_exact)
    int32_t *tie_cnt = nullptr;   // [2 * level + (Rematch ? 1 : 0)]
    double2 *rf_ent[2]{};
    RfUpd *upd_list = nullptr; // k_refine_skew's cache updates
    int32_t *upd_cnt = nullptr;
    int upd_cap = 0;
    int32_t *prefix = nullptr;
    int *d_j1 = nullptr, *d_j2 = nullptr;   // structuring-element spans: Rectify's mask erosion
    int *d_cj1 = nullptr, *d_cj2 = nullptr; // ... and DisparityToCloud's (uploaded with the pair)
    uint8_t *blk = nullptr;                 // coarse bad-block map of the top-level mask (cloud erosion)
    hipEvent_t ev_cloudprep = nullptr;
    int ordinal = 0;               // n-th context created on its device
    int opt_cu_share = 0;          // > 1: the context's streams are confined to one of that many equal shares of the compute units
    hipEvent_t ev_heavy = nullptr; // end of this context's last bandwidth-bound section (heavy_begin / heavy_end)
    hipStream_t stream_filter = nullptr; // rsm_filter_last_cloud's stream: the lowest priority the device offers (filter_stream())
    hipEvent_t ev_filter = nullptr;      // orders the filter behind whatever `stream` still holds
    hipEvent_t ev_heavy2 = nullptr; // ... of its last issue-bound (time-skewed) section: lane 1
    hipEvent_t ev_fork = nullptr, ev_join = nullptr; // the two directions of a time-skewed section on two streams (refine_sweeps)
    RfUpd *upd_list2 = nullptr;     // the second direction's update list / counters while the two run as separate launch chains
    int32_t *upd_cnt2 = nullptr;
    int opt_refine_split = 1;       // a pair that has the GPU to itself runs the two directions of its time-skewed sections on two streams
    int opt_shared_gpu = 0;         // the caller's hint that other contexts use this GPU (a pool of pairs in flight): never split, whatever g_running says at the moment
    int pool_shared = 0;            // the same, derived per call by rsm_run_pairs / rsm_match_pairs from their pool (the caller's option stays as set)
    bool shared_now() const { return opt_shared_gpu || pool_shared; }
    std::atomic<int> in_run{0};     // inside rsm_run_pair (options that replace the streams refuse to act then)
    int32_t *row_count = nullptr;
    int64_t *row_offset = nullptr;
    int64_t *d_npoints = nullptr;
    unsigned long long *d_vtop = nullptr;
    double *d_q = nullptr, *d_R = nullptr, *d_T = nullptr;
    double *xyz = nullptr;
    uint8_t *bgr = nullptr;
    rsm_point16 *pack16 = nullptr; // the cloud as 16-byte records / the filter's output, staged for a host download (on first use)
    float *pack_nrm = nullptr;     // ... and the filter's normals
    FilterArena *filt_arena = nullptr; // the cloud filter's scratch (created on first use, grows with the cloud)
    int opt_filter_wg_max = 2048;      // rsm_filter_last_cloud: the wave passes' workgroup form while at most this many queries are left (0: never)
    int opt_filter_normals_window = 8; // rsm_filter_last_cloud: the normals' radius search on the pixel lattice while no point needs a wider window than this (0: grid)
    int filt_normals[2]{};             // last rsm_filter_last_cloud: the window the normals used (0: the grid), the widest a point needed (-1: not asked)
    int opt_filter_list = 23;          // ... and the 24-pixel window a thread each for what the tile pass leaves over
    int filt_memo_radius = 0, filt_memo_k = 0, filt_memo_w = 0, filt_memo_h = 0, filt_memo_uses = 0; // rsm_filter_last_cloud: the last probe's choice
    int opt_filter_low_priority = 1;   // rsm_filter_last_cloud on a stream of the lowest priority (1) or on the context's own (0)
    int opt_filter_window = 1;         // rsm_filter_last_cloud: the pixel-window k-nearest pass in front of the grid ladder (1: radius from a sparse probe; 0: off; else the radius)
    int64_t filt_tile_left = 0;        // ... queries the tile pass alone left over
    int64_t filt_info[4]{};            // last rsm_filter_last_cloud: window pass used, queries it left to the ladder, points in, points kept

    // results
    double *res_disp[2]{};
    int64_t n_points = 0;
    int64_t v_top = 0;

    // options (rsm_set_option)
    int opt_ncc_bytes = 0;
    int opt_no_exact = 0;
    int opt_no_rowgemm = 0;
    int opt_ncc_mid = 0;         // rows that hold many (RG_MIN) pixels with intervals longer than this go to a row kernel; 0 = by window size
                                 // (measured crossover against the band kernel: 11x11 from ~40 candidates on, 5x5 beyond 160)
    int opt_ncc_slide_max = 512; // rows whose widest interval has at most this many candidates take the sliding-sums kernel, the others the int8 row GEMM
    int opt_heavy_from_sweep = 1;  // ... from this sweep of the level on
    int opt_heavy_min_px = 400000; // ... from this many margin pixels on (smaller levels are launch-bound themselves)
    int opt_heavy_lanes = 2;     // 2: the single-sweep part and the time-skewed part of a level's refine take turns separately (lanes 0 / 1)
    int opt_heavy_exclusive = 1; // refine sections of contexts sharing a GPU take turns (heavy_begin): 1 = the top level's, 2 = every large level's, 0 = none
    int opt_refine_skew_from = 22; // first sweep of a level that may run in the time-skewed kernel (k_refine_skew; 0: never): before that too many pixels still miss the data-term cache for its lane-serial miss service
    int opt_refine_prefill = 1;    // k_refine_first also fills the second cache way with the neighbour iMatch its update points to
    int opt_refine_skew_T = 4;     // sweeps per time-skewed launch (2..4)
    int opt_refine_skew_min_px = 1000000; // ... at levels with at least this many margin pixels per direction (smaller levels: the 4T-step pipeline fill of a chunk eats the gain)
    int opt_refine_skew_waves = 2560;    // workgroups a time-skewed launch aims at (sets the rows per chunk): 5 per CU are resident, so two rounds --
                                         // workgroups at different points of their chunks share a CU better than 1 280 in lockstep (measured: 0.28 against 0.34 ms)
    int opt_refine_skew_waves_alone = 3840; // ... when no other context of the device is inside rsm_run_pair (0: the same)
    int opt_refine_skew_rows = 0;        // > 0: rows per chunk, overrides refine_skew_waves (tests)
    int opt_refine_skew_prio = 0;        // the time-skewed kernel's waves rotate their issue priority every 2^this shader clocks (0: never)
    int opt_refine_skew_uw = 0;          // columns a strip owns; 0: 66 - 2T, all its last level can compute (an even number <= that: A/B)

    // profiling
    bool profile = false;
    bool profile_stages = false;
    std::vector<EvPair> evpool;
    size_t ev_used = 0;
    double prof_ms[ST_COUNT]{};
    int64_t prof_launches[ST_COUNT]{};
    double prof_bytes[ST_COUNT]{};
};


// ---- one bandwidth-bound section at a time per GPU ------------------------------------------------------------
// Contexts that share a GPU (rsm_run_pairs / rsm_match_pairs: several pairs in flight) overlap usefully only where
// one of them leaves the chip idle: the launch-latency-bound small levels, the row-serial stages, host syncs.  Two
// pairs' large-level Jacobi sweeps gain nothing from running side by side -- each streams at the fabric's rate
// alone -- and would only halve each other's speed.  So those sections take turns, ordered on the GPU by events (the host
// never blocks for it): a section waits for the event that ends the previous context's section; the short mutex
// only makes "enqueue the section, publish its end event" atomic.
// MEASURED (C2, two pairs in flight): no turns 214-220 Mdisp/s with every top-level sweep launch twice as long (236 us);
// turns for every large level 210 with isolated-like launches (125 us); turns for the TOP level only (the default) 227-231:
// the other pair's second-largest level and full-size non-refine stages then run beside the top-level sweeps and fill
// their launch tails, at the price of 158 us per top-level launch.
#define RSM_MAX_DEVICES 64
static std::atomic<int> g_running[RSM_MAX_DEVICES]; // contexts inside rsm_run_pair, per device
struct RunningGuard {
    int dev;
    std::atomic<int> *own;
    RunningGuard(int d, std::atomic<int> *in_run) : dev(d), own(in_run) {
        if (dev < RSM_MAX_DEVICES) g_running[dev].fetch_add(1);
        own->store(1);
    }
    ~RunningGuard() {
        own->store(0);
        if (dev < RSM_MAX_DEVICES) g_running[dev].fetch_sub(1);
    }
};
static std::mutex g_heavy_mu[RSM_MAX_DEVICES][2];
static hipEvent_t g_heavy_last[RSM_MAX_DEVICES][2];
static rsm_ctx *g_heavy_owner[RSM_MAX_DEVICES][2];

// lane 0: sections bound by the fabric (one sweep per launch); lane 1 (option heavy_lanes = 2): the time-skewed sections, bound
// by fp64 issue -- a section of one kind runs well beside a section of the other kind of another pair
static bool heavy_begin(rsm_ctx *c, double pixels, bool top, int lane = 0) {
    if (!c->opt_heavy_exclusive || pixels < (double)c->opt_heavy_min_px || c->device >= RSM_MAX_DEVICES) return false;
    if (c->opt_heavy_exclusive == 1 && !top) return false; // 1: only the top level's sweeps take turns
    g_heavy_mu[c->device][lane].lock();
    if (g_heavy_last[c->device][lane] && g_heavy_owner[c->device][lane] != c) (void)hipStreamWaitEvent(c->stream, g_heavy_last[c->device][lane], 0);
    return true;
}
static void heavy_end(rsm_ctx *c, bool held, int lane = 0) {
    if (!held) return;
    hipEvent_t ev = lane ? c->ev_heavy2 : c->ev_heavy;
    (void)hipEventRecord(ev, c->stream);
    g_heavy_last[c->device][lane] = ev;
    g_heavy_owner[c->device][lane] = c;
    g_heavy_mu[c->device][lane].unlock();
}
static void heavy_forget(rsm_ctx *c) { // the context goes away: nobody may wait on its event any more
    if (c->device >= RSM_MAX_DEVICES) return;
    for (int lane = 0; lane < 2; lane++) {
        std::lock_guard<std::mutex> g(g_heavy_mu[c->device][lane]);
        if (g_heavy_owner[c->device][lane] == c) {
            g_heavy_owner[c->device][lane] = nullptr;
            g_heavy_last[c->device][lane] = nullptr;
        }
    }
}

static int set_err(rsm_ctx *c, int code, const char *fmt, ...) {
    if (c) {
        char buf[512];
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(buf, sizeof buf, fmt, ap);
        va_end(ap);
        c->err = buf;
    }
    return code;
}

#define HIPCHK(c, call)                                                                               \
    do {                                                                                              \
        hipError_t e__ = (call);                                                                      \
        if (e__ != hipSuccess)                                                                        \
            return set_err((c), RSM_E_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e__),    \
                           __FILE__, __LINE__);                                                       \
    } while (0)

static Mg to_mg(const rsm_boundary &b) { return Mg{b.YL, b.YR, b.XL, b.XR}; }
static rsm_boundary to_boundary(const Mg &m) {
    return rsm_boundary{m.YL, m.YR, m.XL, m.XR, m.XR - m.XL + 1, m.YR - m.YL + 1};
}

// ---- cv::getStructuringElement(MORPH_ELLIPSE) row spans (OpenCV 2.4; see oracle + DESIGN.md) ----
static void ellipse_spans(int k, std::vector<int> &j1, std::vector<int> &j2) {
    j1.assign(k, 0);
    j2.assign(k, 0);
    const int r = k / 2, c = k / 2;
    const double inv_r2 = r ? 1.0 / ((double)r * r) : 0.0;
    for (int i = 0; i < k; i++) {
        const int dy = i - r;
        if (abs(dy) <= r) {
            const int dx = (int)lrint(c * sqrt((r * r - dy * dy) * inv_r2));
            j1[i] = (c - dx > 0) ? c - dx : 0;
            j2[i] = (c + dx + 1 < k) ? c + dx + 1 : k;
        }
    }
}

// ------------------------------------------------------------------------------------------------
extern "C" const char *rsm_version(void) { return "rsm-mi355 0.3 (gfx950)"; }
extern "C" int rsm_abi_version(void) { return RSM_ABI_VERSION; }
extern "C" int rsm_device_count(void) {
    int n = 0;
    return hipGetDeviceCount(&n) == hipSuccess && n > 0 ? n : 0;
}

extern "C" const char *rsm_last_error(const rsm_ctx *ctx) { return ctx ? ctx->err.c_str() : "null ctx"; }

extern "C" int rsm_create(rsm_ctx **out, int hip_device) {
    if (!out) return RSM_E_INVALID;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return RSM_E_HIP;
    if (hip_device < 0 || hip_device >= ndev) return RSM_E_INVALID;
    rsm_ctx *c = new rsm_ctx();
    c->device = hip_device;
    {
        static std::atomic<int> g_ordinal[RSM_MAX_DEVICES];
        c->ordinal = hip_device < RSM_MAX_DEVICES ? g_ordinal[hip_device].fetch_add(1) : 0; // which share of the CUs option cu_share gives it
    }
    // (stream priorities were measured and dropped: a high-priority main stream next to low-priority sweep streams ran
    //  two pairs in flight at 164 Mdisp/s instead of 210)
    if (hipSetDevice(hip_device) != hipSuccess || hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess ||
        hipStreamCreateWithFlags(&c->stream2, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_pyr, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_cloudprep, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_heavy, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_heavy2, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming) != hipSuccess) {
        delete c;
        return RSM_E_HIP;
    }
    if (const char *e = getenv("RSM_FILTER_LOW_PRIORITY")) c->opt_filter_low_priority = atoi(e) != 0; // (A/B of the adapter loop, whose contexts take no options)
    for (int k = 0; k < RSM_MAX_LEVELS; k++)
        if (hipEventCreateWithFlags(&c->ev_prep[k], hipEventDisableTiming) != hipSuccess) {
            delete c;
            return RSM_E_HIP;
        }
    *out = c;
    return RSM_OK;
}

static void free_workspace(rsm_ctx *c) {
    if (c->stream2) (void)hipStreamSynchronize(c->stream2);
    for (void *p : c->allocs) (void)hipFree(p);
    c->allocs.clear();
    c->pack16 = nullptr;
    c->pack_nrm = nullptr;
    c->cap_px = 0;
    c->have_pair = c->have_result = false;
}

extern "C" void rsm_destroy(rsm_ctx *c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    free_workspace(c);
    filter_arena_destroy(c->filt_arena);
    for (auto &e : c->evpool) {
        (void)hipEventDestroy(e.a);
        (void)hipEventDestroy(e.b);
    }
    (void)hipEventDestroy(c->ev_pyr);
    (void)hipEventDestroy(c->ev_cloudprep);
    heavy_forget(c);
    (void)hipEventDestroy(c->ev_heavy);
    (void)hipEventDestroy(c->ev_heavy2);
    (void)hipEventDestroy(c->ev_fork);
    (void)hipEventDestroy(c->ev_join);
    for (int k = 0; k < RSM_MAX_LEVELS; k++) (void)hipEventDestroy(c->ev_prep[k]);
    if (c->ev_filter) (void)hipEventDestroy(c->ev_filter);
    if (c->stream_filter) (void)hipStreamDestroy(c->stream_filter);
    (void)hipStreamDestroy(c->stream2);
    (void)hipStreamDestroy(c->stream);
    delete c;
}

template <typename T>
static int dalloc(rsm_ctx *c, T **p, size_t n) {
    void *q = nullptr;
    hipError_t e = hipMalloc(&q, n * sizeof(T) + 64);
    if (e != hipSuccess) return set_err(c, RSM_E_NOMEM, "hipMalloc(%zu) failed: %s", n * sizeof(T), hipGetErrorString(e));
    c->allocs.push_back(q);
    *p = (T *)q;
    return RSM_OK;
}
#define DALLOC(c, p, n)                         \
    do {                                        \
        int s__ = dalloc((c), &(p), (size_t)(n)); \
        if (s__ != RSM_OK) return s__;          \
    } while (0)

static int validate(rsm_ctx *c, const rsm_pair_in *in) {
    if (!c || !in) return RSM_E_INVALID;
    if (in->pyr_levels < 1 || in->pyr_levels > RSM_MAX_LEVELS) return set_err(c, RSM_E_INVALID, "pyr_levels");
    if (in->radius < 1 || in->radius > 15) return set_err(c, RSM_E_INVALID, "radius");
    // 16384: the widest margin row k_order can hold in a CU's 160 KB of LDS (6.2 B per column), and int16 columns
    if (in->width <= 0 || in->height <= 0 || in->width > 16384 || in->height > 16384)
        return set_err(c, RSM_E_INVALID, "size (1..16384)");
    const int top = 1 << (in->pyr_levels - 1);
    if ((in->width % top) || (in->height % top)) return set_err(c, RSM_E_INVALID, "size not divisible by 2^(levels-1)");
    if ((in->width / top) <= 2 * in->radius + 2 || (in->height / top) <= 2 * in->radius + 2)
        return set_err(c, RSM_E_INVALID, "lowest level smaller than the match window");
    if (!in->image[0] || !in->image[1] || !in->mask[0] || !in->mask[1]) return set_err(c, RSM_E_INVALID, "null image/mask");
    if (in->origin_width <= 0) return set_err(c, RSM_E_INVALID, "origin_width");
    return RSM_OK;
}

static int ensure_workspace(rsm_ctx *c, const rsm_pair_in *in) {
    const size_t px = (size_t)in->width * in->height;
    const bool same = c->N > 0 && c->cap_px == px && c->N == in->pyr_levels && c->Wk[c->N - 1] == in->width;
    c->in = *in;
    c->in.image[0] = c->in.image[1] = c->in.mask[0] = c->in.mask[1] = nullptr;
    if (same) return RSM_OK;
    free_workspace(c);
    const int N = in->pyr_levels;
    c->N = N;
    for (int k = N - 1; k >= 0; k--) {
        c->Wk[k] = in->width >> (N - 1 - k);
        c->Hk[k] = in->height >> (N - 1 - k);
        for (int v = 0; v < 2; v++) {
            DALLOC(c, c->img[k][v], (size_t)c->Wk[k] * c->Hk[k] * 3);
            DALLOC(c, c->msk[k][v], (size_t)c->Wk[k] * c->Hk[k]);
            DALLOC(c, c->S1[k][v], (size_t)c->Wk[k] * c->Hk[k]);
            DALLOC(c, c->S2[k][v], (size_t)c->Wk[k] * c->Hk[k]);
            DALLOC(c, c->img4[k][v], (size_t)c->Wk[k] * c->Hk[k]);
            DALLOC(c, c->d16i[k][v], (size_t)c->Wk[k] * c->Hk[k]);
            DALLOC(c, c->d16m[k][v], (size_t)c->Wk[k] * c->Hk[k]);
            DALLOC(c, c->d16s[k][v], (size_t)c->Wk[k] * c->Hk[k]);
        }
    }
    DALLOC(c, c->d_margins, RSM_MAX_LEVELS * 2 * 4);
    DALLOC(c, c->tmp1, px);
    DALLOC(c, c->tmp2, px);
    for (int v = 0; v < 2; v++) {
        DALLOC(c, c->d16a[v], px);
        if (v == 0) DALLOC(c, c->cloud_flags, px);
        DALLOC(c, c->BL[v], px);
        DALLOC(c, c->BR[v], px);
        for (int i = 0; i < 3; i++) DALLOC(c, c->f64[i][v], px);
        DALLOC(c, c->nv[v], px / 4 + 16);
        DALLOC(c, c->rf_key[v], px);     // both ways' keys of a pixel in one dword
        DALLOC(c, c->rf_ent[v], 2 * px); // a (pwp, delta) record per way
    }
    DALLOC(c, c->rf_cnt, 32 + 2 * (size_t)in->height); // level k uses rf_cnt + k: [0] wide-pixel count, [16 + dir * H + y] Rematch pixels of a row
    DALLOC(c, c->rf_list, std::max(2 * px + 64, 2 * SETB_SCRATCH(in->width)));
    DALLOC(c, c->tie_list, 2 * px + 64);
    DALLOC(c, c->wrow, NCC_WROW_INTS((size_t)in->height)); // per (direction, row) counters and the row kernels' row lists (rsm_dev.h)
    c->upd_cap = (int)std::min<size_t>(65536, std::max<size_t>(1024, px / 8));
    DALLOC(c, c->upd_list, (size_t)RF_UPD_SHARDS * c->upd_cap);
    DALLOC(c, c->upd_list2, (size_t)RF_UPD_SHARDS * c->upd_cap);
    DALLOC(c, c->upd_cnt2, 2 * RF_UPD_SHARDS);
    DALLOC(c, c->upd_cnt, 2 * RF_UPD_SHARDS);
    DALLOC(c, c->tie_cnt, 2 * RSM_MAX_LEVELS);
    DALLOC(c, c->prefix, (size_t)(in->width + 1) * in->height);
    DALLOC(c, c->blk, CLOUD_BLOCKS(in->width, in->height));
    DALLOC(c, c->d_j1, 4096);
    DALLOC(c, c->d_j2, 4096);
    DALLOC(c, c->d_cj1, 4096);
    DALLOC(c, c->d_cj2, 4096);
    DALLOC(c, c->row_count, (size_t)in->height);
    DALLOC(c, c->row_offset, (size_t)in->height);
    DALLOC(c, c->d_npoints, 2);
    DALLOC(c, c->d_vtop, 2);
    DALLOC(c, c->d_q, 16);
    DALLOC(c, c->d_R, 9);
    DALLOC(c, c->d_T, 3);
    DALLOC(c, c->xyz, px * 3);
    DALLOC(c, c->bgr, px * 3);
    c->cap_px = px;
    return RSM_OK;
}

// The per-pair constants of DisparityToCloud travel with the pair: ellipse spans for ceil(0.02 * rows) (.cpp:703),
// Q with its last column scaled (.cpp:692,698), R_final, T_final.  Synchronous (the sources are temporaries).
static int upload_cloud_params(rsm_ctx *c) {
    const int k = c->N - 1, H = c->Hk[k];
    const int ksize = (int)ceil(0.02 * H);
    if (ksize > 4096) return set_err(c, RSM_E_INVALID, "erode size");
    std::vector<int> j1, j2;
    ellipse_spans(ksize, j1, j2);
    hipStream_t st = c->stream;
    HIPCHK(c, hipMemcpyAsync(c->d_cj1, j1.data(), sizeof(int) * ksize, hipMemcpyHostToDevice, st));
    HIPCHK(c, hipMemcpyAsync(c->d_cj2, j2.data(), sizeof(int) * ksize, hipMemcpyHostToDevice, st));
    double q[16];
    memcpy(q, c->in.Q, sizeof q);
    const double scale = (double)c->Wk[0] / c->in.origin_width * (1 << k); // .cpp:692
    for (int i = 0; i < 4; i++) q[i * 4 + 3] *= scale;                     // .cpp:698
    HIPCHK(c, hipMemcpyAsync(c->d_q, q, sizeof q, hipMemcpyHostToDevice, st));
    HIPCHK(c, hipMemcpyAsync(c->d_R, c->in.R_final, sizeof(double) * 9, hipMemcpyHostToDevice, st));
    HIPCHK(c, hipMemcpyAsync(c->d_T, c->in.T_final, sizeof(double) * 3, hipMemcpyHostToDevice, st));
    HIPCHK(c, hipStreamSynchronize(st));
    return RSM_OK;
}

static int upload_common(rsm_ctx *c, const rsm_pair_in *in, hipMemcpyKind kind) {
    int s = validate(c, in);
    if (s != RSM_OK) return s;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream2)); // nothing of a previous (possibly failed) run may still be in flight
    s = ensure_workspace(c, in);
    if (s != RSM_OK) return s;
    const int top = c->N - 1;
    const size_t px = (size_t)in->width * in->height;
    for (int v = 0; v < 2; v++) {
        HIPCHK(c, hipMemcpyAsync(c->img[top][v], in->image[v], px * 3, kind, c->stream));
        HIPCHK(c, hipMemcpyAsync(c->msk[top][v], in->mask[v], px, kind, c->stream));
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    s = upload_cloud_params(c);
    if (s != RSM_OK) return s;
    c->have_pair = true;
    c->have_result = false;
    return RSM_OK;
}

extern "C" int rsm_upload_pair(rsm_ctx *c, const rsm_pair_in *in) { return upload_common(c, in, hipMemcpyHostToDevice); }
extern "C" int rsm_upload_pair_device(rsm_ctx *c, const rsm_pair_in *in) {
    return upload_common(c, in, hipMemcpyDeviceToDevice);
}

// ---- profiling helpers --------------------------------------------------------------------------
static int prof_slot(rsm_ctx *c, int stage) { // next event pair of the pool
    if (c->ev_used == c->evpool.size()) {
        EvPair e{};
        (void)hipEventCreate(&e.a);
        (void)hipEventCreate(&e.b);
        c->evpool.push_back(e);
    }
    c->evpool[c->ev_used].stage = stage;
    return (int)c->ev_used++;
}
static int prof_begin(rsm_ctx *c, int stage) {
    if (!c->profile || !c->profile_stages) return -1;
    const int s = prof_slot(c, stage);
    (void)hipEventRecord(c->evpool[s].a, c->stream);
    return s;
}
static void prof_end(rsm_ctx *c, int slot, int stage, int launches, double bytes) {
    if (slot < 0) return;
    (void)hipEventRecord(c->evpool[slot].b, c->stream);
    c->prof_launches[stage] += launches;
    c->prof_bytes[stage] += bytes;
}

extern "C" int rsm_set_option(rsm_ctx *c, const char *name, long long value) {
    if (!c || !name) return RSM_E_INVALID;
    if (!strncmp(name, "filter_", 7)) c->filt_memo_radius = 0; // (the cloud filter's own options: it probes its window radius afresh)
    if (!strcmp(name, "ncc_bytes")) c->opt_ncc_bytes = value != 0;
    else if (!strcmp(name, "heavy_from_sweep")) c->opt_heavy_from_sweep = (int)std::max(1LL, std::min(value, 100000LL));
    else if (!strcmp(name, "heavy_min_px")) c->opt_heavy_min_px = (int)std::max(0LL, std::min(value, 2000000000LL));
    else if (!strcmp(name, "heavy_lanes")) c->opt_heavy_lanes = (int)std::max(1LL, std::min(value, 2LL));
    else if (!strcmp(name, "filter_wg_max")) c->opt_filter_wg_max = (int)std::max(0LL, std::min(value, 1LL << 30));
    else if (!strcmp(name, "filter_normals_window")) c->opt_filter_normals_window = (int)std::max(0LL, std::min(value, 40LL));
    else if (!strcmp(name, "filter_low_priority")) c->opt_filter_low_priority = value != 0;
    else if (!strcmp(name, "heavy_exclusive")) c->opt_heavy_exclusive = (int)std::max(0LL, std::min(value, 2LL));
    else if (!strcmp(name, "no_rowgemm")) c->opt_no_rowgemm = value != 0 ? 1 : 0;
    else if (!strcmp(name, "wide_rows")) c->opt_no_rowgemm = (value >= 0 && value <= 3) ? (int)value : 0;
    else if (!strcmp(name, "ncc_mid")) c->opt_ncc_mid = value <= 0 ? 0 : (int)std::max(8LL, std::min(value, 160LL));
    else if (!strcmp(name, "ncc_slide_max")) c->opt_ncc_slide_max = (int)std::max(0LL, std::min(value, 1000000LL));
    else if (!strcmp(name, "no_exact")) c->opt_no_exact = value != 0;
    else if (!strcmp(name, "refine_skew_from")) c->opt_refine_skew_from = (int)std::max(0LL, std::min(value, 100000LL));
    else if (!strcmp(name, "refine_prefill")) c->opt_refine_prefill = value != 0;
    else if (!strcmp(name, "refine_split")) c->opt_refine_split = (int)std::max(0LL, std::min(value, 2LL)); // 2: also with pairs in flight (A/B)
    else if (!strcmp(name, "refine_skew_T")) c->opt_refine_skew_T = (int)std::max(2LL, std::min(value, 4LL));
    else if (!strcmp(name, "refine_skew_min_px")) c->opt_refine_skew_min_px = (int)std::max(0LL, std::min(value, 2000000000LL));
    else if (!strcmp(name, "refine_skew_waves")) c->opt_refine_skew_waves = (int)std::max(1LL, std::min(value, 1000000LL));
    else if (!strcmp(name, "refine_skew_prio")) c->opt_refine_skew_prio = (int)std::max(0LL, std::min(value, 40LL));
    else if (!strcmp(name, "refine_skew_uw")) { // an even number of columns (the state row's 16-byte pieces start on even columns)
        if (value < 0 || value > 62 || (value & 1)) return set_err(c, RSM_E_INVALID, "refine_skew_uw %lld: 0 (= 66 - 2T) or an even number <= 66 - 2T", value);
        c->opt_refine_skew_uw = (int)value;
    } else if (!strcmp(name, "shared_gpu")) c->opt_shared_gpu = value != 0;
    else if (!strcmp(name, "filter_list")) c->opt_filter_list = (int)std::max(0LL, std::min(value, 31LL)); // bit 0: the 24-pixel list pass, bit 1: the 40-pixel one, bit 2: the wave passes (80, 160, ... pixels) for what they leave, bits 3 / 4: the 24- / 40-pixel pass in the wave form too
    else if (!strcmp(name, "filter_window")) c->opt_filter_window = (int)std::max(0LL, std::min(value, 24LL)); // 0 off, 1 default, else the radius
    else if (!strcmp(name, "refine_skew_waves_alone")) c->opt_refine_skew_waves_alone = (int)std::max(0LL, std::min(value, 1000000LL));
    else if (!strcmp(name, "refine_skew_rows")) c->opt_refine_skew_rows = (int)std::max(0LL, std::min(value, 1000000LL));
    else if (!strcmp(name, "cu_share")) {
        // Contexts that share a GPU each on their own share of the compute units (the `ordinal % n`-th of n equal ranges of the
        // CU mask, hipExtStreamCreateWithCUMask) instead of all of them on the whole chip; 0 / 1 = the whole chip.
        const int n = (int)std::max(0LL, std::min(value, 64LL));
        if (c->in_run.load()) return set_err(c, RSM_E_STATE, "cu_share: the context is inside rsm_run_pair (its streams are in use)");
        if (hipSetDevice(c->device) != hipSuccess) return set_err(c, RSM_E_HIP, "hipSetDevice");
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, c->device) != hipSuccess) return set_err(c, RSM_E_HIP, "hipGetDeviceProperties");
        const int ncu = prop.multiProcessorCount;
        (void)hipStreamSynchronize(c->stream);
        (void)hipStreamSynchronize(c->stream2);
        hipStream_t s1 = nullptr, s2 = nullptr;
        hipError_t e1, e2;
        if (n > 1) {
            std::vector<uint32_t> mask((size_t)(ncu + 31) / 32, 0u);
            const int idx = c->ordinal % n, lo = (int)((long long)idx * ncu / n), hi = (int)((long long)(idx + 1) * ncu / n);
            for (int i = lo; i < hi; i++) mask[(size_t)i / 32] |= 1u << (i & 31);
            e1 = hipExtStreamCreateWithCUMask(&s1, (uint32_t)mask.size(), mask.data());
            e2 = hipExtStreamCreateWithCUMask(&s2, (uint32_t)mask.size(), mask.data());
        } else {
            e1 = hipStreamCreateWithFlags(&s1, hipStreamNonBlocking);
            e2 = hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
        }
        if (e1 != hipSuccess || e2 != hipSuccess) {
            if (s1) (void)hipStreamDestroy(s1);
            if (s2) (void)hipStreamDestroy(s2);
            return set_err(c, RSM_E_HIP, "cu_share: stream creation failed");
        }
        (void)hipStreamDestroy(c->stream);
        (void)hipStreamDestroy(c->stream2);
        c->stream = s1;
        c->stream2 = s2;
        c->opt_cu_share = n;
    } else return set_err(c, RSM_E_INVALID, "unknown option %s", name);
    return RSM_OK;
}

extern "C" int rsm_profile_enable(rsm_ctx *c, int on) {
    if (!c) return RSM_E_INVALID;
    c->profile = on != 0;
    c->profile_stages = on == 1; // 2: only the dominant kernel's launches are bracketed
    for (int i = 0; i < ST_COUNT; i++) { // the counters accumulate over the runs that follow
        c->prof_ms[i] = 0;
        c->prof_launches[i] = 0;
        c->prof_bytes[i] = 0;
    }
    return RSM_OK;
}
extern "C" int rsm_profile_stage_count(void) { return ST_COUNT; }
extern "C" const char *rsm_profile_stage_name(int s) { return (s >= 0 && s < ST_COUNT) ? kStageNames[s] : ""; }
extern "C" int rsm_profile_get(rsm_ctx *c, double *ms, int64_t *launches, double *bytes) {
    if (!c) return RSM_E_INVALID;
    for (int i = 0; i < ST_COUNT; i++) {
        if (ms) ms[i] = c->prof_ms[i];
        if (launches) launches[i] = c->prof_launches[i];
        if (bytes) bytes[i] = c->prof_bytes[i];
    }
    return RSM_OK;
}

// ---- stage-argument assembly ---------------------------------------------------------------------
static StageArgs level_args(rsm_ctx *c, int k) {
    StageArgs a{};
    a.ndir = 2;
    a.W = c->Wk[k];
    a.H = c->Hk[k];
    a.Wp = k > 0 ? c->Wk[k - 1] : 0;
    a.Hp = k > 0 ? c->Hk[k - 1] : 0;
    a.r = c->in.radius;
    a.offset = c->in.offset;
    a.ws = c->in.ws;
    a.rf_list = c->rf_list;
    a.ncc_cnt = c->rf_cnt + k; // a counter per level, zeroed together at the start of the run
    a.tie_list = c->tie_list;
    a.tie_cnt = c->tie_cnt + 2 * k;
    a.wrow = c->wrow;
    a.opt_no_rowgemm = c->opt_no_rowgemm;
    a.ncc_mid = c->opt_ncc_mid;
    a.ncc_slide_max = c->opt_ncc_slide_max;
    a.upd_list = c->upd_list;
    a.upd_cnt = c->upd_cnt;
    a.upd_cap = c->upd_cap;
    a.rf_stride = c->cap_px;
    a.opt_ncc_bytes = c->opt_ncc_bytes;
    a.opt_no_exact = c->opt_no_exact;
    for (int v = 0; v < 2; v++) {
        DirArgs &d = a.d[v];
        const int o = 1 - v;
        d.img_own = c->img[k][v];
        d.img_oth = c->img[k][o];
        d.img4_own = c->img4[k][v];
        d.img4_oth = c->img4[k][o];
        d.mask_own = c->msk[k][v];
        d.mask_oth = c->msk[k][o];
        d.S1_own = c->S1[k][v];
        d.S2_own = c->S2[k][v];
        d.S1_oth = c->S1[k][o];
        d.S2_oth = c->S2[k][o];
        d.own = c->mg[k][v];
        d.oth = c->mg[k][o];
        d.BL = c->BL[v];
        d.BR = c->BR[v];
        d.parent_nv = c->nv[v];
        d.rf_key = c->rf_key[v];
        d.rf_ent = c->rf_ent[v];
    }
    return a;
}

static bool degenerate(const Mg &m) { return m.YL >= m.YR || m.XL >= m.XR; } // .cpp:827

// DisparityRefine's Jacobi sweeps (.cpp:590-678): sweep t reads A (t even) or B (t odd) and writes the other.
// Sweep 0 is k_refine_first (every pixel computes its data term); then one launch per sweep (k_refine_sweep) while the
// data-term cache fills, and from sweep `refine_skew_from` on the large levels run T sweeps per launch, time-skewed
// (k_refine_skew.hip), with the leftover sweeps single again.  Returns the number of sweeps launched; *final_in_B tells
// which buffer holds the result.
static int refine_sweeps(rsm_ctx *c, StageArgs &a, double *const bufA[2], double *const bufB[2], int iters,
                         hipStream_t st, bool top, bool *final_in_B, double heavy_px = 0.0) {
    int launches = 0;
    bool curB = false; // the current values are in bufA
    auto bind = [&](int t) {
        for (int v = 0; v < a.ndir; v++) {
            a.d[v].f64_a = curB ? bufB[v] : bufA[v];
            a.d[v].f64_b = curB ? bufA[v] : bufB[v];
        }
        a.flag2 = t;
    };
    auto level_bytes = [&]() { // 16 B per interior pixel (fp64 in + out)
        double bytes = 0;
        for (int v = 0; v < a.ndir; v++) {
            const int r0 = a.d[v].own.YL + 1, r1 = a.d[v].own.YR;
            if (r1 > r0) bytes += 16.0 * (r1 - r0) * (a.d[v].own.XR - a.d[v].own.XL + 1);
        }
        return bytes;
    };
    int nlaunch = 0;
    bool split_now = false; // inside a time-skewed section whose two directions run on two streams (below)
    auto launch = [&](int t, int skewT) { // sweeps t .. t + max(skewT, 1) - 1; skewT > 0: one k_refine_skew launch
        bind(t);
        const bool timed = c && c->profile && top && (nlaunch++ & 7) == 4;
        hipEvent_t e0 = nullptr, e1 = nullptr;
        if (timed) { // every 8th sweep launch of the top level
            const int stg = skewT ? ST_REFINE_SKEW_TOP : ST_REFINE_LIGHT_TOP;
            const int es = prof_slot(c, stg);
            e0 = c->evpool[es].a;
            e1 = c->evpool[es].b;
            c->prof_launches[stg] += 1;
            c->prof_bytes[stg] += (skewT ? (double)skewT : 1.0) * level_bytes();
        }
        if (skewT && split_now) { // the two directions as two launch chains: one's tail runs beside the other's head
            StageArgs a0 = a, a1 = a;
            a0.ndir = a1.ndir = 1;
            a1.d[0] = a.d[1];
            a1.upd_list = c->upd_list2;
            a1.upd_cnt = c->upd_cnt2;
            launch_refine_skew(a0, skewT, st, e0, e1); // (e0 / e1: only under refine_split = 2, an A/B mode)
            launch_refine_skew(a1, skewT, c->stream2, nullptr, nullptr);
        } else if (skewT) launch_refine_skew(a, skewT, st, e0, e1);
        else launch_refine_sweep(a, st, e0, e1);
        launches += skewT ? skewT : 1;
        curB = !curB;
    };
    *final_in_B = false;
    if (iters <= 0) return 0;
    double px = 0;
    for (int v = 0; v < a.ndir; v++) px += (double)(a.d[v].own.XR - a.d[v].own.XL + 1) * (a.d[v].own.YR - a.d[v].own.YL + 1);
    bind(0);
    a.flag3 = (c && !c->opt_refine_prefill) ? 2 : 0;
    launch_refine_sweep(a, st); // k_refine_first: whole interior
    a.flag3 = 0;
    launches++;
    curB = true;
    // settled sweeps of the large levels T per launch, time-skewed (k_refine_skew)
    const bool skew = c && c->opt_refine_skew_from > 0 && a.upd_list && px / a.ndir >= c->opt_refine_skew_min_px && refine_skew_fits(a);
    const int skewT = skew ? c->opt_refine_skew_T : 0;
    if (skew) {
        (void)hipMemsetAsync(a.upd_cnt, 0, sizeof(int32_t) * 2 * RF_UPD_SHARDS, st);
        a.skew_prio = c->opt_refine_skew_prio;
        a.skew_uw = (c->opt_refine_skew_uw > 0 && c->opt_refine_skew_uw <= 66 - 2 * skewT) ? c->opt_refine_skew_uw : 66 - 2 * skewT;
        int rows = 1, strips = 0;
        for (int v = 0; v < a.ndir; v++) {
            rows = std::max(rows, a.d[v].own.YR - a.d[v].own.YL - 1);
            strips += refine_skew_strips(a.d[v].own, skewT, a.skew_uw);
        }
        // A pair that has the GPU to itself takes twice as many, half as tall chunks: its launches end in a tail of one or two
        // waves per SIMD that nothing else fills, and a second round of workgroups shortens it; with pairs in flight the other
        // pairs' kernels fill the tail and the extra pipeline fill only costs.
        const bool lone = c->device < RSM_MAX_DEVICES && !c->shared_now() && g_running[c->device].load() == 1 && c->opt_refine_skew_waves_alone > 0;
        const int waves = lone ? c->opt_refine_skew_waves_alone : c->opt_refine_skew_waves;
        const int chunks = std::max(1, waves / std::max(1, strips));
        a.skew_rows = c->opt_refine_skew_rows > 0 ? c->opt_refine_skew_rows : std::max(4 * skewT, (rows + chunks - 1) / chunks);
    }
    int nskew = 0;
    // the section that takes turns with the other contexts of this GPU starts once the sweeps have settled into pure
    // streaming: the first ones are busy computing data terms (VALU) and run well beside another pair's streaming sweeps
    bool held = false;
    int lane = 0;
    const int turn_from = (c && heavy_px > 0.0) ? std::min(std::max(c->opt_heavy_from_sweep, 1), iters) : iters + 1;
    // A pair that has the GPU to itself (no other context inside rsm_run_pair on this device, no per-launch timing) runs the
    // two directions of a time-skewed section as separate launch chains on its two streams: every skewed launch ends in a
    // tail of one or two waves per SIMD, and the other direction's launch fills it.  With pairs in flight the other pairs'
    // kernels do that already, and the split costs throughput (measured): not used there.
    const bool may_split = skew && a.ndir == 2 && c->opt_refine_split && (!c->profile || c->opt_refine_split == 2) && c->stream2 != st && c->upd_list2 &&
                           c->device < RSM_MAX_DEVICES && (c->opt_refine_split == 2 || (!c->shared_now() && g_running[c->device].load() == 1));
    auto join = [&]() {
        if (!split_now) return;
        (void)hipEventRecord(c->ev_join, c->stream2);
        (void)hipStreamWaitEvent(st, c->ev_join, 0);
        split_now = false;
    };
    for (int t = 1; t < iters;) {
        const bool skew_now = skew && t >= c->opt_refine_skew_from && t + skewT <= iters;
        if (!skew_now) join(); // (before the lane is handed on: the section's end event must cover the second stream's launches)
        if (c && c->opt_heavy_lanes == 2 && (int)skew_now != lane && t >= turn_from) { // the section changes kind: hand its lane on
            heavy_end(c, held, lane);
            held = false;
            lane = (int)skew_now;
        }
        if (t >= turn_from && !held) held = heavy_begin(c, heavy_px, top, lane);
        if (skew_now && may_split && !split_now) { // fork: the second stream continues from here -- AFTER heavy_begin's wait has
                                                   // been enqueued on `st`, so the second direction takes turns like the first
            (void)hipMemsetAsync(c->upd_cnt2, 0, sizeof(int32_t) * 2 * RF_UPD_SHARDS, st);
            (void)hipEventRecord(c->ev_fork, st);
            (void)hipStreamWaitEvent(c->stream2, c->ev_fork, 0);
            split_now = true;
        }
        if (skew_now) {
            a.flag3 = nskew++;
            launch(t, skewT);
            t += skewT;
        } else {
            launch(t, 0);
            t += 1;
        }
    }
    join();
    heavy_end(c, held, lane);
    *final_in_B = curB;
    return launches;
}

extern "C" int rsm_run_pair(rsm_ctx *c) {
    if (!c) return RSM_E_INVALID;
    if (!c->have_pair) return set_err(c, RSM_E_STATE, "rsm_run_pair before rsm_upload_pair");
    HIPCHK(c, hipSetDevice(c->device));
    RunningGuard running(c->device, &c->in_run);
    hipStream_t st = c->stream;
    const int N = c->N, r = c->in.radius;
    c->have_result = false;
    c->ev_used = 0;
    const double P_top_full = (double)c->Wk[N - 1] * c->Hk[N - 1];

    // ConstructPyrm, .cpp:1040-1053 (top level = uploaded images).  The main stream needs the masks (margins) first;
    // everything that depends on the images or the top mask only goes to the side stream and is made while the small
    // levels (launch-latency bound, the GPU mostly idle) are matched: the image pyramid, every level's BGRX copy and
    // NCC window sums (level k waits for ev_prep[k]), and the cloud's bad-pixel prefix + block map (ev_cloudprep).
    HIPCHK(c, hipEventRecord(c->ev_pyr, st)); // everything enqueued before this run
    HIPCHK(c, hipStreamWaitEvent(c->stream2, c->ev_pyr, 0));
    for (int k = N - 2; k >= 0; k--)
        for (int v = 0; v < 2; v++) launch_pyr_down(c->img[k + 1][v], c->Wk[k + 1], c->Hk[k + 1], 3, c->img[k][v], c->stream2);
    for (int k = 0; k < N; k++) {
        for (int v = 0; v < 2; v++) {
            launch_bgr_to_bgrx(c->img[k][v], c->Wk[k], c->Hk[k], c->img4[k][v], c->stream2);
            launch_box_sums(c->img4[k][v], c->Wk[k], c->Hk[k], r, c->tmp1, c->tmp2, c->S1[k][v], c->S2[k][v], c->stream2);
            // the maps the initial match and the median filter write into start as NOMATCH everywhere (.cpp:772)
            launch_fill_i16(c->d16i[k][v], (size_t)c->Wk[k] * c->Hk[k], (int16_t)NOMATCH, c->stream2);
            launch_fill_i16(c->d16m[k][v], (size_t)c->Wk[k] * c->Hk[k], (int16_t)NOMATCH, c->stream2);
            launch_fill_i16(c->d16s[k][v], (size_t)c->Wk[k] * c->Hk[k], (int16_t)NOMATCH, c->stream2);
        }
        HIPCHK(c, hipEventRecord(c->ev_prep[k], c->stream2));
    }
    launch_bad_prefix(c->msk[N - 1][0], c->Wk[N - 1], c->Hk[N - 1], c->prefix, c->stream2);
    launch_bad_blocks(c->prefix, c->Wk[N - 1], c->Hk[N - 1], c->blk, c->stream2);
    HIPCHK(c, hipEventRecord(c->ev_cloudprep, c->stream2));
    const int ps1 = prof_begin(c, ST_PYRAMID);
    for (int k = N - 2; k >= 0; k--)
        for (int v = 0; v < 2; v++) launch_pyr_down(c->msk[k + 1][v], c->Wk[k + 1], c->Hk[k + 1], 1, c->msk[k][v], st);
    prof_end(c, ps1, ST_PYRAMID, 2 * (N - 1), 14.0 * P_top_full);

    // FindMargin for every level and view (.cpp:51-52): depends on the masks only
    const int ps2 = prof_begin(c, ST_MARGIN);
    {
        const uint8_t *masks[RSM_MAX_LEVELS * 2];
        int Ws[RSM_MAX_LEVELS * 2], Hs[RSM_MAX_LEVELS * 2];
        for (int k = 0; k < N; k++)
            for (int v = 0; v < 2; v++) {
                masks[k * 2 + v] = c->msk[k][v];
                Ws[k * 2 + v] = c->Wk[k];
                Hs[k * 2 + v] = c->Hk[k];
            }
        launch_find_margin_batch(2 * N, masks, Ws, Hs, r, c->d_margins, c->h_margin_init, st);
        HIPCHK(c, hipMemsetAsync(c->rf_cnt, 0, sizeof(int) * 16, st)); // the levels' wide-pixel counters
        HIPCHK(c, hipMemsetAsync(c->tie_cnt, 0, sizeof(int) * 2 * RSM_MAX_LEVELS, st));
    }
    prof_end(c, ps2, ST_MARGIN, 1, 0);
    int hm[RSM_MAX_LEVELS * 2 * 4];
    HIPCHK(c, hipMemcpyAsync(hm, c->d_margins, sizeof(int) * N * 2 * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipStreamSynchronize(st));
    for (int k = 0; k < N; k++)
        for (int v = 0; v < 2; v++) {
            const int *m = hm + (k * 2 + v) * 4;
            c->mg[k][v] = Mg{m[2], m[3], m[0], m[1]};
        }
    // Rematch -> SetBoundary_smooth exits on a degenerate margin (.cpp:827-830).  Every level's margin is known now:
    // decide before any level is enqueued, and leave no side-stream work behind that the next call could collide with
    for (int k = 0; k < N; k++)
        if (degenerate(c->mg[k][0]) || degenerate(c->mg[k][1])) {
            (void)hipStreamSynchronize(c->stream2);
            return set_err(c, RSM_E_DEGENERATE_MARGIN, "level %d: YL>=YR || XL>=XR", k);
        }
    launch_count_masked(c->msk[N - 1][0], c->Wk[N - 1], c->Hk[N - 1], c->mg[N - 1][0], c->d_vtop, st);

    int par = 0; // index of the fp64 buffer holding the previous level's disparity (this is `disparity[]`)
    for (int k = 0; k < N; k++) {
        const int W = c->Wk[k], H = c->Hk[k];
        StageArgs a = level_args(c, k);
        const double Pk = 0.5 * ((double)(a.d[0].own.XR - a.d[0].own.XL + 1) * (a.d[0].own.YR - a.d[0].own.YL + 1) +
                                 (double)(a.d[1].own.XR - a.d[1].own.XL + 1) * (a.d[1].own.YR - a.d[1].own.YL + 1));

        const int ps3 = prof_begin(c, ST_BOXSUM); // what the main stream still has to wait for
        HIPCHK(c, hipStreamWaitEvent(st, c->ev_prep[k], 0));
        prof_end(c, ps3, ST_BOXSUM, 6, 0);

        // ---- initial match (.cpp:53-62) -> d16i[k] (NOMATCH-filled by the side stream)
        const int ps4 = prof_begin(c, ST_INITIAL_MATCH);
        for (int v = 0; v < 2; v++) {
            a.d[v].d16_in = c->d16i[k][v];
            a.d[v].d16_out = c->d16i[k][v];
        }
        HIPCHK(c, hipMemsetAsync(c->wrow, 0, sizeof(int32_t) * 2 * (size_t)H, st)); // wide pixels per row of this level
        if (k == 0) {
            launch_ncc_argmax(a, 0, st);
        } else {
            for (int v = 0; v < 2; v++) a.d[v].parent = c->f64[par][v];
            launch_next_valid2(c->f64[par][0], c->f64[par][1], a.Wp, a.Hp, c->nv[0], c->nv[1], st);
            launch_hl_interval(a, st);
            launch_ncc_argmax(a, 1, st);
        }
        prof_end(c, ps4, ST_INITIAL_MATCH, k == 0 ? 3 : 6, 24.0 * Pk);

        // ---- SmoothConstraint (.cpp:66-67): d16i[k] -> d16s[k] (both NOMATCH outside the margins: no copy)
        const int ps5 = prof_begin(c, ST_SMOOTH);
        for (int v = 0; v < 2; v++) {
            a.d[v].d16_in = c->d16i[k][v];
            a.d[v].d16_out = c->d16s[k][v];
        }
        launch_smooth(a, st, false);
        prof_end(c, ps5, ST_SMOOTH, 1, 8.0 * Pk);

        // ---- OrderConstraint (.cpp:71-72): d16s[k] in place
        const int ps6 = prof_begin(c, ST_ORDER);
        for (int v = 0; v < 2; v++) a.d[v].d16_in = a.d[v].d16_out = c->d16s[k][v];
        launch_order(a, st);
        prof_end(c, ps6, ST_ORDER, 1, 8.0 * Pk);

        // ---- UniquenessContraint<short> (.cpp:75)
        const int ps7 = prof_begin(c, ST_UNIQ16);
        launch_uniq_s16(c->d16s[k][0], c->d16s[k][1], W, H, c->mg[k][0], c->mg[k][1], st);
        launch_uniq_s16(c->d16s[k][1], c->d16s[k][0], W, H, c->mg[k][1], c->mg[k][0], st);
        launch_uniq_s16(c->d16s[k][0], c->d16s[k][1], W, H, c->mg[k][0], c->mg[k][1], st);
        prof_end(c, ps7, ST_UNIQ16, 3, 18.0 * Pk);

        // ---- Rematch (.cpp:80-81): SetBoundary_smooth + NCC on still-unmatched pixels, in place
        const int ps8 = prof_begin(c, ST_REMATCH);
        launch_set_boundary(a, st, true);
        launch_ncc_argmax(a, 2, st);
        prof_end(c, ps8, ST_REMATCH, 3, 46.0 * Pk);

        // ---- UniquenessContraint<short> (.cpp:86)
        const int ps9 = prof_begin(c, ST_UNIQ16);
        launch_uniq_s16(c->d16s[k][0], c->d16s[k][1], W, H, c->mg[k][0], c->mg[k][1], st);
        launch_uniq_s16(c->d16s[k][1], c->d16s[k][0], W, H, c->mg[k][1], c->mg[k][0], st);
        launch_uniq_s16(c->d16s[k][0], c->d16s[k][1], W, H, c->mg[k][0], c->mg[k][1], st);
        prof_end(c, ps9, ST_UNIQ16, 3, 18.0 * Pk);

        // ---- MedianFilter (.cpp:89-90): d16s[k] -> d16m[k] (pre-filled NOMATCH, .cpp:772)
        const int ps10 = prof_begin(c, ST_MEDIAN);
        for (int v = 0; v < 2; v++) {
            a.d[v].d16_in = c->d16s[k][v];
            a.d[v].d16_out = c->d16m[k][v];
        }
        launch_median(a, st);
        prof_end(c, ps10, ST_MEDIAN, 3, 10.0 * Pk);

        // ---- DisparityRefine (.cpp:95-98): int16 d16m[k] -> fp64, 30 + 30k Jacobi sweeps
        const int iters = 30 + k * 30;
        const int ia = (par + 1) % 3, ib = (par + 2) % 3;
        const int ps11 = prof_begin(c, ST_REFINE_INIT);
        for (int v = 0; v < 2; v++) {
            a.d[v].d16_in = c->d16m[k][v];
            a.d[v].f64_a = c->f64[ia][v];
            a.d[v].f64_b = c->f64[ib][v];
        }
        launch_refine_init(a, st);
        prof_end(c, ps11, ST_REFINE_INIT, 1, 12.0 * Pk);
        const int st_sweep = (k == N - 1) ? ST_REFINE_SWEEP_TOP : ST_REFINE_SWEEP;
        const int ps12 = prof_begin(c, st_sweep);
        a.flag = (k == N - 1);
        double *bufA[2] = {c->f64[ia][0], c->f64[ia][1]}, *bufB[2] = {c->f64[ib][0], c->f64[ib][1]};
        bool inB = false;
        const int nlaunch = refine_sweeps(c, a, bufA, bufB, iters, st, k == N - 1, &inB, Pk);
        const int cur = inB ? ib : ia;
        prof_end(c, ps12, st_sweep, nlaunch, 32.0 * Pk * iters);

        // ---- UniquenessContraint<double> (.cpp:109) on the refined maps (now in f64[cur])
        const int ps13 = prof_begin(c, ST_UNIQ64);
        launch_uniq_f64(c->f64[cur][0], c->f64[cur][1], W, H, c->mg[k][0], c->mg[k][1], st);
        launch_uniq_f64(c->f64[cur][1], c->f64[cur][0], W, H, c->mg[k][1], c->mg[k][0], st);
        launch_uniq_f64(c->f64[cur][0], c->f64[cur][1], W, H, c->mg[k][0], c->mg[k][1], st);
        prof_end(c, ps13, ST_UNIQ64, 3, 72.0 * Pk);
        par = cur;
        { // a launch that could not start (e.g. an LDS request the input drove too high) must not go unnoticed
            const hipError_t le = hipGetLastError();
            if (le != hipSuccess) {
                (void)hipStreamSynchronize(c->stream2);
                (void)hipStreamSynchronize(st);
                return set_err(c, RSM_E_HIP, "level %d: kernel launch failed: %s", k, hipGetErrorString(le));
            }
        }
    }

    // ---- DisparityToCloud<double>(disparity[0], maskPyrm[top][0], Q, top, true) (.cpp:29)
    {
        const int k = N - 1, W = c->Wk[k], H = c->Hk[k];
        const int ps14 = prof_begin(c, ST_CLOUD);
        const int ksize = (int)ceil(0.02 * H); // .cpp:703; spans, Q, R, T: upload_cloud_params
        HIPCHK(c, hipStreamWaitEvent(st, c->ev_cloudprep, 0));
        launch_cloud(c->f64[par][0], c->prefix, c->img[k][0], W, H, ksize, c->d_cj1, c->d_cj2, c->d_q, c->d_R, c->d_T,
                     c->mg[k][0], c->cloud_flags, c->blk, c->row_count, c->row_offset, c->d_npoints, c->xyz, c->bgr, (int64_t)c->cap_px, st);
        const Mg &m = c->mg[k][0];
        prof_end(c, ps14, ST_CLOUD, 4, 28.0 * (double)(m.XR - m.XL + 1) * (m.YR - m.YL + 1));
    }
    int64_t np = 0;
    unsigned long long vt = 0;
    HIPCHK(c, hipMemcpyAsync(&np, c->d_npoints, sizeof np, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipMemcpyAsync(&vt, c->d_vtop, sizeof vt, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipStreamSynchronize(st));
    HIPCHK(c, hipGetLastError());
    c->n_points = np;
    c->v_top = (int64_t)vt;
    c->res_disp[0] = c->f64[par][0];
    c->res_disp[1] = c->f64[par][1];
    c->have_result = true;
    if (c->profile) {
        for (size_t i = 0; i < c->ev_used; i++) {
            float ms = 0;
            if (hipEventElapsedTime(&ms, c->evpool[i].a, c->evpool[i].b) == hipSuccess) c->prof_ms[c->evpool[i].stage] += ms;
        }
    }
    return RSM_OK;
}

extern "C" int rsm_download_pair(rsm_ctx *c, rsm_pair_out *out) {
    if (!c || !out) return RSM_E_INVALID;
    if (!c->have_result) return set_err(c, RSM_E_STATE, "no result to download");
    HIPCHK(c, hipSetDevice(c->device));
    const int k = c->N - 1;
    const size_t px = (size_t)c->Wk[k] * c->Hk[k];
    for (int v = 0; v < 2; v++) {
        out->margin[v] = to_boundary(c->mg[k][v]); // .cpp:27-28
        if (out->disparity[v])
            HIPCHK(c, hipMemcpyAsync(out->disparity[v], c->res_disp[v], px * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    }
    out->n_points = c->n_points;
    out->v_top = c->v_top;
    int64_t n = c->n_points;
    if (n > out->max_points) n = out->max_points;
    if (n > (int64_t)c->cap_px) n = (int64_t)c->cap_px;
    if (n > 0 && out->xyz) HIPCHK(c, hipMemcpyAsync(out->xyz, c->xyz, (size_t)n * 3 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    if (n > 0 && out->bgr) HIPCHK(c, hipMemcpyAsync(out->bgr, c->bgr, (size_t)n * 3, hipMemcpyDeviceToHost, c->stream));
    if (n > 0 && out->points16) { // InsertPoint's float cast + the colour packed on the GPU: 16 instead of 27 bytes per point cross PCIe
        if (!c->pack16) {
            const int s = dalloc(c, &c->pack16, c->cap_px);
            if (s != RSM_OK) return s;
        }
        launch_pack_cloud16(c->xyz, c->bgr, n, c->pack16, c->stream);
        HIPCHK(c, hipMemcpyAsync(out->points16, c->pack16, (size_t)n * sizeof(rsm_point16), hipMemcpyDeviceToHost, c->stream));
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return RSM_OK;
}

// Page-locked host memory for the buffers that cross the boundary.  The reference's pipeline keeps its images and results
// in pageable cv::Mat memory (CManageData.cpp:75-78, CStereoMatching.cpp:682-761); a download into such memory goes through
// the runtime's staging copy at ~10 GB/s (34 ms for a C2 pair's two fp64 maps and cloud), into page-locked memory it is one
// DMA at the link's rate.  A cv::Mat can wrap memory from rsm_host_alloc (its user-data constructor), or memory the
// caller already owns can be page-locked in place (rsm_host_register).
extern "C" void *rsm_host_alloc(size_t bytes) {
    void *p = nullptr;
    if (bytes == 0 || hipHostMalloc(&p, bytes, hipHostMallocPortable) != hipSuccess) return nullptr; // (every GPU of the node may DMA into it)
    return p;
}
extern "C" void rsm_host_free(void *p) {
    if (p) (void)hipHostFree(p);
}
extern "C" int rsm_host_register(void *p, size_t bytes) {
    if (!p || bytes == 0) return RSM_E_INVALID;
    return hipHostRegister(p, bytes, hipHostRegisterDefault) == hipSuccess ? RSM_OK : RSM_E_HIP;
}
extern "C" int rsm_host_unregister(void *p) {
    if (!p) return RSM_E_INVALID;
    return hipHostUnregister(p) == hipSuccess ? RSM_OK : RSM_E_HIP;
}

extern "C" int rsm_result_device(rsm_ctx *c, const double **d0, const double **d1, int64_t *n_points,
                                 const double **xyz, const uint8_t **bgr) {
    if (!c) return RSM_E_INVALID;
    if (!c->have_result) return set_err(c, RSM_E_STATE, "no result");
    if (d0) *d0 = c->res_disp[0];
    if (d1) *d1 = c->res_disp[1];
    if (n_points) *n_points = c->n_points;
    if (xyz) *xyz = c->xyz;
    if (bgr) *bgr = c->bgr;
    return RSM_OK;
}

extern "C" int rsm_export_cloud_device(rsm_ctx *c, double *d_xyz, uint8_t *d_bgr, int64_t max_points) {
    if (!c) return RSM_E_INVALID;
    if (!c->have_result) return set_err(c, RSM_E_STATE, "no result");
    HIPCHK(c, hipSetDevice(c->device));
    int64_t n = c->n_points < max_points ? c->n_points : max_points;
    if (n > (int64_t)c->cap_px) n = (int64_t)c->cap_px;
    if (n > 0 && d_xyz) HIPCHK(c, hipMemcpyAsync(d_xyz, c->xyz, (size_t)n * 3 * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
    if (n > 0 && d_bgr) HIPCHK(c, hipMemcpyAsync(d_bgr, c->bgr, (size_t)n * 3, hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return RSM_OK;
}

extern "C" int rsm_pack_cloud16(rsm_ctx *c, rsm_point16 *d_dst, int64_t max_points, int64_t *n_points) {
    if (!c) return RSM_E_INVALID;
    if (!c->have_result) return set_err(c, RSM_E_STATE, "no result");
    HIPCHK(c, hipSetDevice(c->device));
    int64_t n = c->n_points < max_points ? c->n_points : max_points;
    if (n > (int64_t)c->cap_px) n = (int64_t)c->cap_px;
    if (n > 0 && !d_dst) return RSM_E_INVALID;
    launch_pack_cloud16(c->xyz, c->bgr, n, d_dst, c->stream);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (n_points) *n_points = n > 0 ? n : 0;
    return RSM_OK;
}

extern "C" int rsm_match_pair(rsm_ctx *c, const rsm_pair_in *in, rsm_pair_out *out) {
    int s = rsm_upload_pair(c, in);
    if (s != RSM_OK) return s;
    s = rsm_run_pair(c);
    if (s != RSM_OK) return s;
    return rsm_download_pair(c, out);
}


// =================================================================================================
// Several pairs at once: the pair loop of CStereoMatching::MatchAllLayer (.cpp:17-33) has no cross-pair data flow,
// so pairs can be in flight together -- on one GPU (a pair's launch-latency-bound small levels and host syncs hide
// under another pair's top-level sweeps) and across the GPUs of a node (contexts on different devices).
// One host thread per context; every context keeps its own streams and workspace.
// =================================================================================================
extern "C" int rsm_run_pairs(rsm_ctx *const *ctxs, int n) { return rsm_run_pairs_repeat(ctxs, n, 1); }

extern "C" int rsm_run_pairs_repeat(rsm_ctx *const *ctxs, int n, int repeats) {
    if (!ctxs || n <= 0 || repeats < 1) return RSM_E_INVALID;
    for (int i = 0; i < n; i++) {
        if (!ctxs[i]) return RSM_E_INVALID;
        for (int j = 0; j < i; j++)
            if (ctxs[j] == ctxs[i]) return RSM_E_INVALID; // a context is not re-entrant
    }
    for (int i = 0; i < n; i++) { // contexts that share a device with another one of the pool: pairs in flight, never the lone-pair split
        bool shared = false;
        for (int j = 0; j < n; j++) shared |= j != i && ctxs[j]->device == ctxs[i]->device;
        ctxs[i]->pool_shared = shared;
    }
    std::vector<int> st((size_t)n, RSM_OK);
    auto loop = [&](int i) { // every context runs its resident pair `repeats` times, free of the others
        for (int k = 0; k < repeats && st[(size_t)i] == RSM_OK; k++) st[(size_t)i] = rsm_run_pair(ctxs[i]);
    };
    std::vector<std::thread> th;
    th.reserve((size_t)n - 1);
    for (int i = 1; i < n; i++) th.emplace_back(loop, i);
    loop(0);
    for (auto &t : th) t.join();
    for (int i = 0; i < n; i++) ctxs[i]->pool_shared = 0;
    for (int i = 0; i < n; i++)
        if (st[(size_t)i] != RSM_OK) return st[(size_t)i];
    return RSM_OK;
}

extern "C" int rsm_match_pairs(rsm_ctx *const *ctxs, int n_ctx, const rsm_pair_in *in, rsm_pair_out *out, int n_pairs,
                               int *status) {
    if (!ctxs || n_ctx <= 0 || n_pairs < 0 || (n_pairs > 0 && (!in || !out))) return RSM_E_INVALID;
    for (int i = 0; i < n_ctx; i++) {
        if (!ctxs[i]) return RSM_E_INVALID;
        for (int j = 0; j < i; j++)
            if (ctxs[j] == ctxs[i]) return RSM_E_INVALID;
    }
    for (int i = 0; i < n_ctx; i++) { // (see rsm_run_pairs_repeat)
        bool shared = false;
        for (int j = 0; j < n_ctx; j++) shared |= j != i && ctxs[j]->device == ctxs[i]->device;
        ctxs[i]->pool_shared = shared;
    }
    std::atomic<int> next(0);
    std::vector<int> st((size_t)n_pairs, RSM_OK);
    // Work queue: a context takes the next pair when it is free, so that upload / download of one pair (PCIe)
    // overlaps another pair's kernels and uneven pairs balance themselves.  A failed pair (e.g.
    // RSM_E_DEGENERATE_MARGIN, the reference's exit(0)) does not stop the others.
    auto worker = [&](rsm_ctx *c) {
        for (int p = next.fetch_add(1); p < n_pairs; p = next.fetch_add(1)) st[(size_t)p] = rsm_match_pair(c, &in[p], &out[p]);
    };
    const int nt = std::min(n_ctx, std::max(n_pairs, 1));
    std::vector<std::thread> th;
    for (int i = 1; i < nt; i++) th.emplace_back(worker, ctxs[i]);
    worker(ctxs[0]);
    for (auto &t : th) t.join();
    for (int i = 0; i < n_ctx; i++) ctxs[i]->pool_shared = 0;
    int first = RSM_OK;
    for (int p = 0; p < n_pairs; p++) {
        if (status) status[p] = st[(size_t)p];
        if (first == RSM_OK && st[(size_t)p] != RSM_OK) first = st[(size_t)p];
    }
    return first;
}

extern "C" int rsm_match_pairs_multi_gpu(const rsm_pair_in *in, int n_pairs, int n_gpus, int pairs_in_flight,
                                         rsm_pair_out *out, int *status) {
    if (n_pairs < 0 || (n_pairs > 0 && (!in || !out)) || n_gpus < 0 || pairs_in_flight < 0) return RSM_E_INVALID;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return RSM_E_HIP;
    if (n_gpus == 0 || n_gpus > ndev) n_gpus = ndev;
    if (pairs_in_flight == 0) pairs_in_flight = 2;
    std::vector<rsm_ctx *> ctxs;
    int s = RSM_OK;
    // context i lives on GPU i % n_gpus: the queue hands consecutive pairs to different GPUs first
    for (int i = 0; i < n_gpus * pairs_in_flight && i < std::max(n_pairs, 1) && s == RSM_OK; i++) {
        rsm_ctx *c = nullptr;
        s = rsm_create(&c, i % n_gpus);
        if (s == RSM_OK) ctxs.push_back(c);
    }
    if (s == RSM_OK) s = rsm_match_pairs(ctxs.data(), (int)ctxs.size(), in, out, n_pairs, status);
    for (rsm_ctx *c : ctxs) rsm_destroy(c);
    return s;
}

// =================================================================================================
// Per-stage entry points (host buffers, one direction) for the parity tests.
// =================================================================================================
namespace {
struct Tmp { // scoped device allocations of one stage call
    rsm_ctx *c;
    std::vector<void *> ptrs;
    bool ok = true;
    explicit Tmp(rsm_ctx *c_) : c(c_) {}
    ~Tmp() {
        for (void *p : ptrs) (void)hipFree(p);
    }
    template <typename T>
    T *alloc(size_t n) {
        void *p = nullptr;
        if (hipMalloc(&p, n * sizeof(T) + 64) != hipSuccess) {
            ok = false;
            return nullptr;
        }
        ptrs.push_back(p);
        return (T *)p;
    }
    template <typename T>
    T *up(const T *h, size_t n) {
        T *d = alloc<T>(n);
        if (d && (hipMemcpyAsync(d, h, n * sizeof(T), hipMemcpyHostToDevice, c->stream) != hipSuccess ||
                  hipStreamSynchronize(c->stream) != hipSuccess))
            ok = false;
        return d;
    }
    template <typename T>
    void down(T *h, const T *d, size_t n) {
        if (hipMemcpyAsync(h, d, n * sizeof(T), hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
            hipStreamSynchronize(c->stream) != hipSuccess)
            ok = false;
    }
};
int finish(rsm_ctx *c, Tmp &t) {
    hipError_t e = hipStreamSynchronize(c->stream);
    if (e == hipSuccess) e = hipGetLastError();
    if (e != hipSuccess) return set_err(c, RSM_E_HIP, "stage failed: %s", hipGetErrorString(e));
    if (!t.ok) return set_err(c, RSM_E_HIP, "stage alloc/copy failed");
    return RSM_OK;
}
bool stage_ok(rsm_ctx *c, int W, int H) { return c && W > 0 && H > 0 && hipSetDevice(c->device) == hipSuccess; }
} // namespace

extern "C" int rsm_stage_find_margin(rsm_ctx *c, const uint8_t *mask, int W, int H, int r, rsm_boundary *m) {
    if (!stage_ok(c, W, H) || !mask || !m) return RSM_E_INVALID;
    Tmp t(c);
    uint8_t *dm = t.up(mask, (size_t)W * H);
    int *d4 = t.alloc<int>(4);
    if (!t.ok) return finish(c, t);
    launch_find_margin(dm, W, H, r, d4, c->stream);
    int h[4];
    t.down(h, d4, 4);
    *m = to_boundary(Mg{h[2], h[3], h[0], h[1]});
    return finish(c, t);
}

extern "C" int rsm_stage_pyr_down(rsm_ctx *c, const uint8_t *src, int W, int H, int ch, uint8_t *dst) {
    if (!stage_ok(c, W, H) || !src || !dst || (ch != 1 && ch != 3)) return RSM_E_INVALID;
    Tmp t(c);
    const size_t nd = (size_t)((W + 1) / 2) * ((H + 1) / 2) * ch;
    uint8_t *ds = t.up(src, (size_t)W * H * ch);
    uint8_t *dd = t.alloc<uint8_t>(nd);
    if (!t.ok) return finish(c, t);
    launch_pyr_down(ds, W, H, ch, dd, c->stream);
    t.down(dst, dd, nd);
    return finish(c, t);
}

extern "C" int rsm_stage_erode_ellipse(rsm_ctx *c, const uint8_t *mask, int W, int H, int ksize, uint8_t *dst) {
    if (!stage_ok(c, W, H) || !mask || !dst || ksize < 1 || ksize > 4096) return RSM_E_INVALID;
    Tmp t(c);
    std::vector<int> j1, j2;
    ellipse_spans(ksize, j1, j2);
    uint8_t *dm = t.up(mask, (size_t)W * H);
    int *d1 = t.up(j1.data(), (size_t)ksize), *d2 = t.up(j2.data(), (size_t)ksize);
    int32_t *pre = t.alloc<int32_t>((size_t)(W + 1) * H);
    uint8_t *dd = t.alloc<uint8_t>((size_t)W * H);
    if (!t.ok) return finish(c, t);
    launch_bad_prefix(dm, W, H, pre, c->stream);
    launch_erode_binary(pre, W, H, ksize, d1, d2, dd, c->stream);
    t.down(dst, dd, (size_t)W * H);
    return finish(c, t);
}

namespace {
// uploads the images/masks of one direction and builds the window-sum tables
struct MatchBufs {
    uint8_t *io, *it, *mo, *mt;
    uint32_t *i4o, *i4t;
    uint32_t *wl; // wide-pixel worklist + counter of the NCC kernels
    int32_t *wc;
    uint32_t *tl; // tie list + counters (k_ncc_exact)
    int32_t *tc;
    int32_t *wr;  // wide pixels per row
    int32_t *S1o, *S2o, *S1t, *S2t;
};
bool setup_match(rsm_ctx *c, Tmp &t, const uint8_t *img_own, const uint8_t *img_oth, const uint8_t *mask_own,
                 const uint8_t *mask_oth, int W, int H, int r, MatchBufs &b) {
    const size_t px = (size_t)W * H;
    b.io = t.up(img_own, px * 3);
    b.it = t.up(img_oth, px * 3);
    b.mo = t.up(mask_own, px);
    b.mt = t.up(mask_oth, px);
    b.wl = t.alloc<uint32_t>(std::max(px + 64, SETB_SCRATCH(W))); // NCC worklist / SetBoundary scratch
    b.wc = t.alloc<int32_t>(16 + 2 * (size_t)H);
    if (b.wc) (void)hipMemsetAsync(b.wc, 0, sizeof(int), c->stream); // wide-pixel counter of the NCC launch
    b.tl = t.alloc<uint32_t>(px + 64);
    b.tc = t.alloc<int32_t>(2);
    if (b.tc) (void)hipMemsetAsync(b.tc, 0, 2 * sizeof(int), c->stream);
    b.wr = t.alloc<int32_t>(NCC_WROW_INTS((size_t)H));
    if (b.wr) (void)hipMemsetAsync(b.wr, 0, sizeof(int32_t) * NCC_WROW_INTS((size_t)H), c->stream);
    b.i4o = t.alloc<uint32_t>(px);
    b.i4t = t.alloc<uint32_t>(px);
    b.S1o = t.alloc<int32_t>(px);
    b.S2o = t.alloc<int32_t>(px);
    b.S1t = t.alloc<int32_t>(px);
    b.S2t = t.alloc<int32_t>(px);
    int32_t *t1 = t.alloc<int32_t>(px), *t2 = t.alloc<int32_t>(px);
    if (!t.ok) return false;
    launch_bgr_to_bgrx(b.io, W, H, b.i4o, c->stream);
    launch_bgr_to_bgrx(b.it, W, H, b.i4t, c->stream);
    launch_box_sums(b.i4o, W, H, r, t1, t2, b.S1o, b.S2o, c->stream);
    launch_box_sums(b.i4t, W, H, r, t1, t2, b.S1t, b.S2t, c->stream);
    return true;
}
StageArgs one_dir(rsm_ctx *c, int W, int H, int r, const rsm_boundary *own, const rsm_boundary *oth) {
    StageArgs a{};
    a.opt_ncc_bytes = c->opt_ncc_bytes;
    a.opt_no_exact = c->opt_no_exact;
    a.opt_no_rowgemm = c->opt_no_rowgemm;
    a.ncc_mid = c->opt_ncc_mid;
    a.ncc_slide_max = c->opt_ncc_slide_max;
    a.ndir = 1;
    a.W = W;
    a.H = H;
    a.r = r;
    a.d[0].own = to_mg(*own);
    if (oth) a.d[0].oth = to_mg(*oth);
    return a;
}
void bind_match(StageArgs &a, const MatchBufs &b) {
    a.rf_list = b.wl;
    a.ncc_cnt = b.wc;
    a.tie_list = b.tl;
    a.tie_cnt = b.tc;
    a.wrow = b.wr;
    DirArgs &d = a.d[0];
    d.img_own = b.io;
    d.img_oth = b.it;
    d.img4_own = b.i4o;
    d.img4_oth = b.i4t;
    d.mask_own = b.mo;
    d.mask_oth = b.mt;
    d.S1_own = b.S1o;
    d.S2_own = b.S2o;
    d.S1_oth = b.S1t;
    d.S2_oth = b.S2t;
}
} // namespace

extern "C" int rsm_stage_initial_match(rsm_ctx *c, const uint8_t *img_own, const uint8_t *img_oth,
                                       const uint8_t *mask_own, const uint8_t *mask_oth, int W, int H, int r,
                                       int offset, const rsm_boundary *own, const rsm_boundary *oth,
                                       const double *parent, int Wp, int Hp, int16_t *disp) {
    if (!stage_ok(c, W, H) || !img_own || !img_oth || !mask_own || !mask_oth || !own || !oth || !disp) return RSM_E_INVALID;
    Tmp t(c);
    const size_t px = (size_t)W * H;
    MatchBufs b{};
    if (!setup_match(c, t, img_own, img_oth, mask_own, mask_oth, W, H, r, b)) return finish(c, t);
    StageArgs a = one_dir(c, W, H, r, own, oth);
    bind_match(a, b);
    a.offset = offset;
    a.Wp = Wp;
    a.Hp = Hp;
    int16_t *dd = t.alloc<int16_t>(px);
    a.d[0].BL = t.alloc<int16_t>(px);
    a.d[0].BR = t.alloc<int16_t>(px);
    if (!t.ok) return finish(c, t);
    launch_fill_i16(dd, px, (int16_t)NOMATCH, c->stream);
    a.d[0].d16_in = a.d[0].d16_out = dd;
    if (!parent) {
        launch_ncc_argmax(a, 0, c->stream);
    } else {
        double *dp = t.up(parent, (size_t)Wp * Hp);
        int32_t *nv = t.alloc<int32_t>((size_t)Wp * Hp);
        if (!t.ok) return finish(c, t);
        a.d[0].parent = dp;
        a.d[0].parent_nv = nv;
        launch_next_valid(dp, Wp, Hp, nv, c->stream);
        launch_hl_interval(a, c->stream);
        launch_ncc_argmax(a, 1, c->stream);
    }
    t.down(disp, dd, px);
    return finish(c, t);
}

extern "C" int rsm_stage_smooth(rsm_ctx *c, int16_t *disp, int W, int H, const rsm_boundary *own) {
    if (!stage_ok(c, W, H) || !disp || !own) return RSM_E_INVALID;
    Tmp t(c);
    const size_t px = (size_t)W * H;
    StageArgs a = one_dir(c, W, H, 0, own, nullptr);
    a.d[0].d16_in = t.up(disp, px);
    a.d[0].d16_out = t.alloc<int16_t>(px);
    if (!t.ok) return finish(c, t);
    launch_smooth(a, c->stream);
    t.down(disp, a.d[0].d16_out, px);
    return finish(c, t);
}

extern "C" int rsm_stage_order(rsm_ctx *c, int16_t *disp, int W, int H, const rsm_boundary *own) {
    if (!stage_ok(c, W, H) || !disp || !own) return RSM_E_INVALID;
    Tmp t(c);
    const size_t px = (size_t)W * H;
    StageArgs a = one_dir(c, W, H, 0, own, nullptr);
    a.d[0].d16_in = a.d[0].d16_out = t.up(disp, px);
    if (!t.ok) return finish(c, t);
    launch_order(a, c->stream);
    t.down(disp, a.d[0].d16_in, px);
    return finish(c, t);
}

extern "C" int rsm_stage_uniqueness_pass_s16(rsm_ctx *c, int16_t *p, const int16_t *q, int W, int H,
                                             const rsm_boundary *own, const rsm_boundary *oth) {
    if (!stage_ok(c, W, H) || !p || !q || !own || !oth) return RSM_E_INVALID;
    Tmp t(c);
    const size_t px = (size_t)W * H;
    int16_t *dp = t.up(p, px);
    int16_t *dq = t.up(q, px);
    if (!t.ok) return finish(c, t);
    launch_uniq_s16(dp, dq, W, H, to_mg(*own), to_mg(*oth), c->stream);
    t.down(p, dp, px);
    return finish(c, t);
}

extern "C" int rsm_stage_uniqueness_pass_f64(rsm_ctx *c, double *p, const double *q, int W, int H,
                                             const rsm_boundary *own, const rsm_boundary *oth) {
    if (!stage_ok(c, W, H) || !p || !q || !own || !oth) return RSM_E_INVALID;
    Tmp t(c);
    const size_t px = (size_t)W * H;
    double *dp = t.up(p, px);
    double *dq = t.up(q, px);
    if (!t.ok) return finish(c, t);
    launch_uniq_f64(dp, dq, W, H, to_mg(*own), to_mg(*oth), c->stream);
    t.down(p, dp, px);
    return finish(c, t);
}

extern "C" int rsm_stage_set_boundary(rsm_ctx *c, const int16_t *disp, const uint8_t *mask_own, int W, int H,
                                      const rsm_boundary *own, const rsm_boundary *oth, int16_t *BL, int16_t *BR) {
    if (!stage_ok(c, W, H) || !disp || !mask_own || !own || !oth || !BL || !BR) return RSM_E_INVALID;
    if (degenerate(to_mg(*own))) return set_err(c, RSM_E_DEGENERATE_MARGIN, "YL>=YR || XL>=XR");
    Tmp t(c);
    const size_t px = (size_t)W * H;
    StageArgs a = one_dir(c, W, H, 0, own, oth);
    a.d[0].d16_in = t.up(disp, px);
    a.d[0].mask_own = t.up(mask_own, px);
    a.d[0].BL = t.alloc<int16_t>(px);
    a.d[0].BR = t.alloc<int16_t>(px);
    a.rf_list = t.alloc<uint32_t>(SETB_SCRATCH(W));
    if (!t.ok) return finish(c, t);
    launch_fill_i16(a.d[0].BL, px, (int16_t)-10000, c->stream);
    launch_fill_i16(a.d[0].BR, px, (int16_t)10000, c->stream);
    launch_set_boundary(a, c->stream);
    t.down(BL, a.d[0].BL, px);
    t.down(BR, a.d[0].BR, px);
    return finish(c, t);
}

extern "C" int rsm_stage_rematch(rsm_ctx *c, const uint8_t *img_own, const uint8_t *img_oth, const uint8_t *mask_own,
                                 const uint8_t *mask_oth, int W, int H, int r, const rsm_boundary *own,
                                 const rsm_boundary *oth, int16_t *disp) {
    if (!stage_ok(c, W, H) || !img_own || !img_oth || !mask_own || !mask_oth || !own || !oth || !disp) return RSM_E_INVALID;
    if (degenerate(to_mg(*own))) return set_err(c, RSM_E_DEGENERATE_MARGIN, "YL>=YR || XL>=XR");
    Tmp t(c);
    const size_t px = (size_t)W * H;
    MatchBufs b{};
    if (!setup_match(c, t, img_own, img_oth, mask_own, mask_oth, W, H, r, b)) return finish(c, t);
    StageArgs a = one_dir(c, W, H, r, own, oth);
    bind_match(a, b);
    a.d[0].d16_in = a.d[0].d16_out = t.up(disp, px);
    a.d[0].BL = t.alloc<int16_t>(px);
    a.d[0].BR = t.alloc<int16_t>(px);
    if (!t.ok) return finish(c, t);
    launch_set_boundary(a, c->stream, true);
    launch_ncc_argmax(a, 2, c->stream);
    t.down(disp, a.d[0].d16_in, px);
    return finish(c, t);
}

extern "C" int rsm_stage_median(rsm_ctx *c, int16_t *disp, const uint8_t *mask_own, int W, int H,
                                const rsm_boundary *own) {
    if (!stage_ok(c, W, H) || !disp || !mask_own || !own) return RSM_E_INVALID;
    Tmp t(c);
    const size_t px = (size_t)W * H;
    StageArgs a = one_dir(c, W, H, 0, own, nullptr);
    a.d[0].d16_in = t.up(disp, px);
    a.d[0].mask_own = t.up(mask_own, px);
    a.d[0].d16_out = t.alloc<int16_t>(px);
    if (!t.ok) return finish(c, t);
    launch_fill_i16(a.d[0].d16_out, px, (int16_t)NOMATCH, c->stream);
    launch_median(a, c->stream);
    t.down(disp, a.d[0].d16_out, px);
    return finish(c, t);
}

// The specified exp(-t) of DisparityRefine's smoothness weights (k_refine.hip: exp_neg) on an array: lets the parity
// tests hold the device evaluation to the oracle's bit for bit over the whole argument range.
static int stage_exp_neg(rsm_ctx *c, const double *t_in, int64_t n, double *out, int small_form) {
    if (!c || !t_in || !out || n < 0) return RSM_E_INVALID;
    if (n == 0) return RSM_OK;
    if (hipSetDevice(c->device) != hipSuccess) return set_err(c, RSM_E_HIP, "hipSetDevice");
    Tmp t(c);
    const double *dt = t.up(t_in, (size_t)n);
    double *dout = t.alloc<double>((size_t)n);
    if (!t.ok) return finish(c, t);
    launch_exp_neg(dt, dout, (long long)n, c->stream, small_form);
    t.down(out, dout, (size_t)n);
    return finish(c, t);
}
extern "C" int rsm_stage_exp_neg(rsm_ctx *c, const double *t_in, int64_t n, double *out) { return stage_exp_neg(c, t_in, n, out, 0); }
// the form the time-skewed kernel's common path uses (no special-case code) on every argument below 512, the general one elsewhere
extern "C" int rsm_stage_exp_neg_small(rsm_ctx *c, const double *t_in, int64_t n, double *out) { return stage_exp_neg(c, t_in, n, out, 1); }

// k_refine_skew's unscaled division beside the compiler's (k_refine.hip: div_unscaled) on arrays of operands
extern "C" int rsm_stage_div_unscaled(rsm_ctx *c, const double *a_in, const double *b_in, int64_t n, double *q_fast, double *q_ieee) {
    if (!c || !a_in || !b_in || !q_fast || !q_ieee || n < 0) return RSM_E_INVALID;
    if (n == 0) return RSM_OK;
    if (hipSetDevice(c->device) != hipSuccess) return set_err(c, RSM_E_HIP, "hipSetDevice");
    Tmp t(c);
    const double *da = t.up(a_in, (size_t)n), *db = t.up(b_in, (size_t)n);
    double *df = t.alloc<double>((size_t)n), *di = t.alloc<double>((size_t)n);
    if (!t.ok) return finish(c, t);
    launch_div_unscaled(da, db, df, di, (long long)n, c->stream);
    t.down(q_fast, df, (size_t)n);
    t.down(q_ieee, di, (size_t)n);
    return finish(c, t);
}

// the cloud filter's trimmed sqrtf (k_filter.hip: sqrtf_rn) against the compiler's on the floats with bit patterns first .. first + n - 1
extern "C" int rsm_stage_sqrt_check(rsm_ctx *c, uint32_t first_bits, int64_t n, int64_t *mismatches) {
    if (!c || !mismatches || n < 0 || (uint64_t)first_bits + (uint64_t)n > (1ull << 32)) return RSM_E_INVALID;
    *mismatches = 0;
    if (n == 0) return RSM_OK;
    if (hipSetDevice(c->device) != hipSuccess) return set_err(c, RSM_E_HIP, "hipSetDevice");
    Tmp t(c);
    unsigned long long *d = t.alloc<unsigned long long>(1);
    if (!t.ok) return finish(c, t);
    if (hipMemsetAsync(d, 0, sizeof(unsigned long long), c->stream) != hipSuccess) return set_err(c, RSM_E_HIP, "hipMemsetAsync");
    launch_sqrt_check(first_bits, (long long)n, d, c->stream);
    unsigned long long h = 0;
    t.down(&h, d, 1);
    const int rc = finish(c, t);
    *mismatches = (int64_t)h;
    return rc;
}

extern "C" int rsm_stage_refine(rsm_ctx *c, const int16_t *disp_in, const uint8_t *img_own, const uint8_t *img_oth,
                                int W, int H, int iterations, double ws, const rsm_boundary *own, double *disp_out) {
    if (!stage_ok(c, W, H) || !disp_in || !img_own || !img_oth || !own || !disp_out || iterations < 0) return RSM_E_INVALID;
    Tmp t(c);
    const size_t px = (size_t)W * H;
    StageArgs a = one_dir(c, W, H, 0, own, nullptr);
    a.ws = ws;
    DirArgs &d = a.d[0];
    d.d16_in = t.up(disp_in, px);
    d.img_own = t.up(img_own, px * 3);
    d.img_oth = t.up(img_oth, px * 3);
    uint32_t *i4o = t.alloc<uint32_t>(px), *i4t = t.alloc<uint32_t>(px);
    double *A = t.alloc<double>(px), *B = t.alloc<double>(px);
    d.rf_key = t.alloc<uint32_t>(px);
    d.rf_ent = t.alloc<double2>(2 * px);
    a.rf_stride = px;
    a.upd_cap = 4096;
    a.upd_list = t.alloc<RfUpd>((size_t)RF_UPD_SHARDS * a.upd_cap);
    a.upd_cnt = t.alloc<int32_t>(2 * RF_UPD_SHARDS);
    if (!t.ok) return finish(c, t);
    if (iterations > RF_MAX_SWEEPS) return set_err(c, RSM_E_INVALID, "iterations");
    d.f64_a = A;
    d.f64_b = B;
    launch_bgr_to_bgrx(d.img_own, W, H, i4o, c->stream);
    launch_bgr_to_bgrx(d.img_oth, W, H, i4t, c->stream);
    d.img4_own = i4o;
    d.img4_oth = i4t;
    launch_refine_init(a, c->stream);
    double *bufA[2] = {A, nullptr}, *bufB[2] = {B, nullptr};
    bool inB = false;
    refine_sweeps(c, a, bufA, bufB, iterations, c->stream, false, &inB);
    d.f64_a = inB ? B : A; // the buffer the last sweep wrote
    t.down(disp_out, (const double *)d.f64_a, px);
    return finish(c, t);
}

// DisparityRefine's matching costs xi (CStereoMatching.cpp:624-629) as the device restatements of the data term compute them:
// lets the parity tests hold them to the compiled reference's own values (tests/golden: xi_table_*), bit for bit.
extern "C" int rsm_stage_refine_xi(rsm_ctx *c, const uint8_t *img_own, const uint8_t *img_oth, int W, int H, int form, double *out) {
    if (!stage_ok(c, W, H) || !img_own || !img_oth || !out || W < 3 || H < 3 || form < 0 || form > 2) return RSM_E_INVALID;
    Tmp t(c);
    const size_t px = (size_t)W * H, n = (size_t)(H - 2) * (W - 2) * (W - 2);
    const uint8_t *io = t.up(img_own, px * 3), *it = t.up(img_oth, px * 3);
    uint32_t *i4o = t.alloc<uint32_t>(px), *i4t = t.alloc<uint32_t>(px);
    double *dout = t.alloc<double>(3 * n);
    if (!t.ok) return finish(c, t);
    launch_bgr_to_bgrx(io, W, H, i4o, c->stream);
    launch_bgr_to_bgrx(it, W, H, i4t, c->stream);
    launch_refine_xi(i4o, i4t, W, H, form, dout, c->stream);
    t.down(out, (const double *)dout, 3 * n);
    return finish(c, t);
}

extern "C" int rsm_stage_cloud(rsm_ctx *c, const double *disp, const uint8_t *mask_org, const uint8_t *img_own, int W,
                               int H, const double *Q, double scale, const double *R_final, const double *T_final,
                               const rsm_boundary *own, double *xyz, uint8_t *bgr, int64_t max_points,
                               int64_t *n_points) {
    if (!stage_ok(c, W, H) || !disp || !mask_org || !img_own || !Q || !R_final || !T_final || !own || !n_points)
        return RSM_E_INVALID;
    Tmp t(c);
    const size_t px = (size_t)W * H;
    const int ksize = (int)ceil(0.02 * H);
    if (ksize < 1 || ksize > 4096) return RSM_E_INVALID;
    std::vector<int> j1, j2;
    ellipse_spans(ksize, j1, j2);
    double q[16];
    memcpy(q, Q, sizeof q);
    for (int i = 0; i < 4; i++) q[i * 4 + 3] *= scale;
    double *dd = t.up(disp, px);
    uint8_t *dm = t.up(mask_org, px);
    uint8_t *di = t.up(img_own, px * 3);
    int *d1 = t.up(j1.data(), (size_t)ksize), *d2 = t.up(j2.data(), (size_t)ksize);
    double *dq = t.up(q, 16), *dR = t.up(R_final, 9), *dT = t.up(T_final, 3);
    int32_t *pre = t.alloc<int32_t>((size_t)(W + 1) * H);
    int32_t *rc = t.alloc<int32_t>((size_t)H);
    int64_t *ro = t.alloc<int64_t>((size_t)H);
    int64_t *dn = t.alloc<int64_t>(1);
    const int64_t cap = max_points > 0 ? max_points : 0;
    double *dx = (xyz && cap) ? t.alloc<double>((size_t)cap * 3) : nullptr;
    uint8_t *db = (bgr && cap) ? t.alloc<uint8_t>((size_t)cap * 3) : nullptr;
    if (!t.ok) return finish(c, t);
    launch_bad_prefix(dm, W, H, pre, c->stream);
    uint8_t *fl = t.alloc<uint8_t>(px + CLOUD_BLOCKS(W, H));
    if (!t.ok) return finish(c, t);
    launch_bad_blocks(pre, W, H, fl + px, c->stream);
    launch_cloud(dd, pre, di, W, H, ksize, d1, d2, dq, dR, dT, to_mg(*own), fl, fl + px, rc, ro, dn, dx, db, cap, c->stream);
    int64_t n = 0;
    t.down(&n, (const int64_t *)dn, 1);
    *n_points = n;
    const int64_t m = n < cap ? n : cap;
    if (m > 0 && dx) t.down(xyz, (const double *)dx, (size_t)m * 3);
    if (m > 0 && db) t.down(bgr, (const uint8_t *)db, (size_t)m * 3);
    return finish(c, t);
}


// =================================================================================================
// Rectify (CStereoMatching.cpp:117-168): host fp64 plan + device maps / remap / grey erosion.
// =================================================================================================
extern "C" int rsm_stereo_rectify(const double *K1, const double *K2, int nx, int ny, const double *R, const double *T,
                                  double *R1, double *R2, double *P1, double *P2, double *Q) {
    if (!K1 || !K2 || !R || !T || !R1 || !R2 || !P1 || !P2 || !Q || nx <= 0 || ny <= 0) return RSM_E_INVALID;
    stereo_rectify_host(K1, K2, nx, ny, R, T, R1, R2, P1, P2, Q);
    return RSM_OK;
}

extern "C" int rsm_rectify_pair(rsm_ctx *c, const rsm_rectify_in *in, int radius, double ws, int offset, int verbose,
                                rsm_rectify_out *out) {
    if (!c || !in || !out) return RSM_E_INVALID;
    if (in->origin_width <= 0 || in->origin_height <= 0 || in->lowest_width <= 0 || in->lowest_height <= 0 ||
        in->pyr_levels < 1 || in->pyr_levels > RSM_MAX_LEVELS)
        return set_err(c, RSM_E_INVALID, "rectify: sizes");
    for (int v = 0; v < 2; v++)
        if (!in->image[v] || !in->mask[v]) return set_err(c, RSM_E_INVALID, "read image error"); // .cpp:147-151
    HIPCHK(c, hipSetDevice(c->device));
    RectifyPlan plan;
    rectify_plan(in->K[0], in->K[1], in->E[0], in->E[1], in->origin_width, in->origin_height, in->lowest_width,
                 in->lowest_height, in->pyr_levels, &plan);
    rsm_pair_in pin{};
    pin.width = plan.W;
    pin.height = plan.H;
    pin.pyr_levels = in->pyr_levels;
    pin.radius = radius;
    pin.ws = ws;
    pin.offset = offset;
    pin.origin_width = in->origin_width;
    pin.verbose = verbose;
    memcpy(pin.Q, plan.Q, sizeof pin.Q);
    memcpy(pin.R_final, plan.R_final, sizeof pin.R_final);
    memcpy(pin.T_final, plan.T_final, sizeof pin.T_final);
    for (int v = 0; v < 2; v++) { // only to satisfy validate(); the rectified images are produced on the device
        pin.image[v] = in->image[v];
        pin.mask[v] = in->mask[v];
    }
    int s = validate(c, &pin);
    if (s != RSM_OK) return s;
    HIPCHK(c, hipStreamSynchronize(c->stream2)); // the maps below reuse the side stream's scratch (tmp1 / tmp2)
    s = ensure_workspace(c, &pin);
    if (s != RSM_OK) return s;
    hipStream_t st = c->stream;
    const int top = c->N - 1, W = plan.W, H = plan.H;
    const size_t opx = (size_t)in->origin_width * in->origin_height, px = (size_t)W * H;
    Tmp t(c);
    uint8_t *raw_img = t.alloc<uint8_t>(opx * 3), *raw_msk = t.alloc<uint8_t>(opx);
    if (!t.ok) return set_err(c, RSM_E_NOMEM, "rectify: raw buffers");
    std::vector<int> j1, j2;
    if (plan.ksize > 4096) return set_err(c, RSM_E_INVALID, "erode size");
    ellipse_spans(plan.ksize, j1, j2);
    HIPCHK(c, hipMemcpyAsync(c->d_j1, j1.data(), sizeof(int) * plan.ksize, hipMemcpyHostToDevice, st));
    HIPCHK(c, hipMemcpyAsync(c->d_j2, j2.data(), sizeof(int) * plan.ksize, hipMemcpyHostToDevice, st));
    int16_t *map1 = (int16_t *)c->tmp1;   // 4 B / pixel
    uint16_t *map2 = (uint16_t *)c->tmp2; // 2 B / pixel
    uint8_t *mtmp = (uint8_t *)c->d16a[0];
    uint8_t *stbuf = (uint8_t *)c->f64[0][0]; // sparse-table planes: <= 8 B / pixel
    int levels = 1;
    while ((1 << levels) <= plan.ksize) levels++;
    if (levels > 8) return set_err(c, RSM_E_INVALID, "erode size");
    for (int v = 0; v < 2; v++) {
        HIPCHK(c, hipMemcpyAsync(raw_img, in->image[v], opx * 3, hipMemcpyHostToDevice, st));
        HIPCHK(c, hipMemcpyAsync(raw_msk, in->mask[v], opx, hipMemcpyHostToDevice, st));
        const double *K = in->K[v];
        launch_rect_map(plan.ir[v], K[0], K[4], K[2], K[5], W, H, map1, map2, st);                 // :144
        launch_remap(raw_img, in->origin_width, in->origin_height, 3, map1, map2, W, H, c->img[top][v], st); // :154
        launch_remap(raw_msk, in->origin_width, in->origin_height, 1, map1, map2, W, H, mtmp, st);          // :156
        launch_erode_gray(mtmp, W, H, plan.ksize, c->d_j1, c->d_j2, stbuf, c->msk[top][v], st);             // :157-158
        HIPCHK(c, hipStreamSynchronize(st)); // raw buffers are reused by the next view
    }
    HIPCHK(c, hipGetLastError());
    memcpy(out->Q, plan.Q, sizeof plan.Q);
    memcpy(out->R_final, plan.R_final, sizeof plan.R_final);
    memcpy(out->T_final, plan.T_final, sizeof plan.T_final);
    for (int v = 0; v < 2; v++) {
        memcpy(out->P[v], plan.Pext[v], sizeof plan.Pext[v]);
        if (out->image[v]) HIPCHK(c, hipMemcpyAsync(out->image[v], c->img[top][v], px * 3, hipMemcpyDeviceToHost, st));
        if (out->mask[v]) HIPCHK(c, hipMemcpyAsync(out->mask[v], c->msk[top][v], px, hipMemcpyDeviceToHost, st));
    }
    out->width = W;
    out->height = H;
    HIPCHK(c, hipStreamSynchronize(st));
    s = upload_cloud_params(c);
    if (s != RSM_OK) return s;
    c->have_pair = true;
    c->have_result = false;
    return RSM_OK;
}

extern "C" int rsm_stage_rect_map(rsm_ctx *c, const double *A, const double *R, const double *newA, int W, int H,
                                  int16_t *map1, uint16_t *map2) {
    if (!stage_ok(c, W, H) || !A || !R || !newA || !map1 || !map2) return RSM_E_INVALID;
    // (newA * R)^-1 through the same host routine the pipeline uses
    double AR[9], ir[9];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) AR[3 * i + j] = newA[3 * i] * R[j] + newA[3 * i + 1] * R[3 + j] + newA[3 * i + 2] * R[6 + j];
    rectify_inv3(AR, ir);
    Tmp t(c);
    const size_t px = (size_t)W * H;
    int16_t *d1 = t.alloc<int16_t>(px * 2);
    uint16_t *d2 = t.alloc<uint16_t>(px);
    if (!t.ok) return finish(c, t);
    launch_rect_map(ir, A[0], A[4], A[2], A[5], W, H, d1, d2, c->stream);
    t.down(map1, (const int16_t *)d1, px * 2);
    t.down(map2, (const uint16_t *)d2, px);
    return finish(c, t);
}

extern "C" int rsm_stage_remap(rsm_ctx *c, const uint8_t *src, int Ws, int Hs, int ch, const int16_t *map1,
                               const uint16_t *map2, int W, int H, uint8_t *dst) {
    if (!stage_ok(c, W, H) || !src || !map1 || !map2 || !dst || Ws <= 0 || Hs <= 0 || (ch != 1 && ch != 3)) return RSM_E_INVALID;
    Tmp t(c);
    const size_t px = (size_t)W * H;
    uint8_t *ds = t.up(src, (size_t)Ws * Hs * ch);
    int16_t *d1 = t.up(map1, px * 2);
    uint16_t *d2 = t.up(map2, px);
    uint8_t *dd = t.alloc<uint8_t>(px * ch);
    if (!t.ok) return finish(c, t);
    launch_remap(ds, Ws, Hs, ch, d1, d2, W, H, dd, c->stream);
    t.down(dst, (const uint8_t *)dd, px * ch);
    return finish(c, t);
}

extern "C" int rsm_stage_erode_gray(rsm_ctx *c, const uint8_t *src, int W, int H, int ksize, uint8_t *dst) {
    if (!stage_ok(c, W, H) || !src || !dst || ksize < 1 || ksize > 255) return RSM_E_INVALID;
    Tmp t(c);
    const size_t px = (size_t)W * H;
    std::vector<int> j1, j2;
    ellipse_spans(ksize, j1, j2);
    uint8_t *ds = t.up(src, px);
    int *d1 = t.up(j1.data(), (size_t)ksize), *d2 = t.up(j2.data(), (size_t)ksize);
    uint8_t *stb = t.alloc<uint8_t>(px * 9), *dd = t.alloc<uint8_t>(px);
    if (!t.ok) return finish(c, t);
    launch_erode_gray(ds, W, H, ksize, d1, d2, stb, dd, c->stream);
    t.down(dst, (const uint8_t *)dd, px);
    return finish(c, t);
}

// ---- per-pair cloud filter (CloudOptimization/CCloudOptimization.cpp:82-121) -------------------------------------
static int filter_params_ok(const rsm_filter_params *p) {
    return p && p->sor_mean_k >= 1 && p->sor_mean_k <= 100000 && p->normal_radius > 0.0 && p->sor_std_mul == p->sor_std_mul;
}

// the filter's buffers come from the context's arena: (re)sized for the cloud at hand, then the caller-visible ones first
static int filter_buffers(rsm_ctx *c, int64_t n, bool want_normals, float **dx, int32_t **dk, float **df, float4 **dn, size_t extra = 0) {
    if (!c->filt_arena) c->filt_arena = filter_arena_create();
    const size_t own = (size_t)n * (12 + 4 + 12 + 16) + 4096 + extra;
    if (filter_arena_reserve(c->filt_arena, own + filter_arena_bytes(n)) != RSM_OK)
        return set_err(c, RSM_E_NOMEM, "cloud filter: no device memory for %lld points", (long long)n);
    *dx = (float *)filter_arena_alloc(c->filt_arena, sizeof(float) * 3 * (size_t)n);
    *dk = (int32_t *)filter_arena_alloc(c->filt_arena, sizeof(int32_t) * (size_t)n);
    *df = (float *)filter_arena_alloc(c->filt_arena, sizeof(float) * 3 * (size_t)n);
    *dn = want_normals ? (float4 *)filter_arena_alloc(c->filt_arena, sizeof(float4) * (size_t)n) : nullptr;
    if (!*dx || !*dk || !*df || (want_normals && !*dn)) return set_err(c, RSM_E_NOMEM, "cloud filter: arena too small");
    return RSM_OK;
}

extern "C" int rsm_filter_cloud(rsm_ctx *c, const float *xyz, int64_t n, const rsm_filter_params *prm, int32_t *kept_index,
                                float *normals, int64_t *n_kept, double *stats) {
    if (!c || n < 0 || (n > 0 && (!xyz || !kept_index)) || !n_kept || !filter_params_ok(prm)) return RSM_E_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    *n_kept = 0;
    if (n == 0) return RSM_OK;
    Tmp t(c);
    float *dx, *df;
    int32_t *dk;
    float4 *dn;
    const int sb = filter_buffers(c, n, normals != nullptr, &dx, &dk, &df, &dn);
    if (sb != RSM_OK) return sb;
    HIPCHK(c, hipMemcpyAsync(dx, xyz, sizeof(float) * 3 * (size_t)n, hipMemcpyHostToDevice, c->stream));
    const int s = filter_cloud_device(c->filt_arena, dx, n, prm->sor_mean_k, prm->sor_std_mul, prm->normal_radius, prm->cam_center, dk, df, dn,
                                      n_kept, stats, c->stream);
    if (s != RSM_OK) return set_err(c, s, "cloud filter failed");
    if (*n_kept > 0) {
        t.down(kept_index, (const int32_t *)dk, (size_t)*n_kept);
        if (normals) t.down(normals, (const float *)dn, (size_t)4 * *n_kept);
    }
    return finish(c, t);
}

// The stream of the per-pair cloud filter.  With several pairs in flight on a GPU the filter of one pair runs beside the matching of
// the others, and its large kernels (21 000 workgroups of 75 KB LDS) took the compute units the matchers' dependent launches -- the
// top level's refine sweeps, the loop's critical path -- were waiting for: 29.5 ms per pair in the adapter's loop for 15.3 ms of
// matching + 9.5 ms of filter.  On a stream of the lowest priority the dispatcher hands compute units to the matchers first.
static hipStream_t filter_stream(rsm_ctx *c) {
    if (!c->opt_filter_low_priority) return c->stream;
    if (!c->stream_filter) {
        int least = 0, greatest = 0;
        if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess || hipStreamCreateWithPriority(&c->stream_filter, hipStreamNonBlocking, least) != hipSuccess ||
            hipEventCreateWithFlags(&c->ev_filter, hipEventDisableTiming) != hipSuccess) {
            (void)hipGetLastError();
            if (c->stream_filter) (void)hipStreamDestroy(c->stream_filter);
            c->stream_filter = nullptr;
            return c->stream;
        }
    }
    if (hipEventRecord(c->ev_filter, c->stream) != hipSuccess || hipStreamWaitEvent(c->stream_filter, c->ev_filter, 0) != hipSuccess) return c->stream;
    return c->stream_filter;
}

extern "C" int rsm_filter_last_cloud(rsm_ctx *c, const rsm_filter_params *prm, rsm_point16 *d_points, float *d_normals,
                                     int64_t max_points, int64_t *n_kept, double *stats) {
    if (!c || !n_kept || !filter_params_ok(prm)) return RSM_E_INVALID;
    if (!c->have_result) return set_err(c, RSM_E_STATE, "no result");
    HIPCHK(c, hipSetDevice(c->device));
    *n_kept = 0;
    const int64_t n = c->n_points;
    if (n == 0) return RSM_OK;
    Tmp t(c);
    float *dx, *df;
    int32_t *dk;
    float4 *dn;
    // The cloud is the depth map this context just made: its pixel lattice (k_cloud's flags and row offsets are still in place)
    // decides most k-nearest queries (k_filter.hip: k_sor_window).  Needs R_final to be a rotation (distances in the cloud =
    // distances in the camera frame); anything else takes the generic search.
    const int k = c->N - 1;
    const Mg &mg = c->mg[k][0];
    FilterLattice lat{};
    bool use_lat = c->opt_filter_window && mg.XR >= mg.XL && mg.YR >= mg.YL;
    if (use_lat) {
        const double *R = c->in.R_final;
        for (int i = 0; i < 3 && use_lat; i++)
            for (int j = 0; j < 3; j++) {
                double d = 0.0;
                for (int l = 0; l < 3; l++) d += R[3 * l + i] * R[3 * l + j];
                if (!(fabs(d - (i == j ? 1.0 : 0.0)) < 1e-9)) use_lat = false;
            }
        const double scale = (double)c->Wk[0] / c->in.origin_width * (1 << k); // .cpp:692
        lat.flags = c->cloud_flags;
        lat.row_offset = c->row_offset;
        lat.W = c->Wk[k];
        lat.XL = mg.XL, lat.XR = mg.XR, lat.YL = mg.YL, lat.YR = mg.YR;
        lat.xyz64 = c->xyz;
        lat.qz = c->in.Q[11] * scale;
        memcpy(lat.R, c->in.R_final, sizeof lat.R);
        memcpy(lat.T, c->in.T_final, sizeof lat.T);
        if (!(fabs(lat.qz) > 0.0) || !std::isfinite(lat.qz)) use_lat = false;
    }
    int left = -1, tile_left = -1;
    lat.undecided_out = &left;
    lat.tile_left_out = &tile_left;
    lat.list_pass = c->opt_filter_list;
    lat.wg_max = c->opt_filter_wg_max;
    lat.normals_wmax = std::min(c->opt_filter_normals_window, 40);
    c->filt_normals[0] = 0, c->filt_normals[1] = -1;
    lat.normals_out = c->filt_normals;
    int used_radius = 0;
    lat.radius_out = &used_radius;
    lat.radius = c->opt_filter_window <= 1 ? 0 : (c->opt_filter_window <= 7 ? 7 : (c->opt_filter_window <= 12 ? 12 : (c->opt_filter_window <= 16 ? 16 : (c->opt_filter_window <= 20 ? 20 : 24))));
    // The probed radius is a property of the rig (how thick its clouds are in pixel spacings): a context remembers what the probe
    // chose for its last cloud and skips the three probe launches and their host round trips (0.6 ms of C2's 14.8) while the
    // choice keeps deciding most queries; every 8th call, a different k, a different image size or a set_option probes again.
    const bool memo_ok = c->opt_filter_window == 1 && c->filt_memo_radius > 0 && c->filt_memo_k == prm->sor_mean_k && c->filt_memo_w == c->Wk[k] &&
                         c->filt_memo_h == c->Hk[k] && c->filt_memo_uses < 7;
    if (use_lat && memo_ok) lat.radius = c->filt_memo_radius;
    const int sb = filter_buffers(c, n, d_normals != nullptr, &dx, &dk, &df, &dn, use_lat ? cloud_lattice_bytes(mg.XL, mg.XR, mg.YL, mg.YR) + (size_t)n * 4 + 8192 : 0);
    if (sb != RSM_OK) return sb;
    const hipStream_t fs = filter_stream(c);
    launch_f64_to_f32x3(c->xyz, n, dx, fs); // InsertPoint's cast, CCloudOptimization.cpp:61
    int64_t m = 0;
    const int s = filter_cloud_device(c->filt_arena, dx, n, prm->sor_mean_k, prm->sor_std_mul, prm->normal_radius, prm->cam_center, dk, df, dn, &m, stats,
                                      fs, use_lat ? &lat : nullptr);
    if (s != RSM_OK) return set_err(c, s, "cloud filter failed");
    if (use_lat && c->opt_filter_window == 1) {
        if (memo_ok && tile_left >= 0 && (double)tile_left <= 0.3 * (double)n) c->filt_memo_uses++; // still a good choice
        else if (!memo_ok && used_radius > 0) { // a fresh probe's choice
            c->filt_memo_radius = used_radius;
            c->filt_memo_k = prm->sor_mean_k;
            c->filt_memo_w = c->Wk[k];
            c->filt_memo_h = c->Hk[k];
            c->filt_memo_uses = 0;
        } else c->filt_memo_radius = 0; // left too much over (or no window at all): probe next time
    }
    c->filt_info[0] = left >= 0 ? used_radius : 0;
    c->filt_info[1] = left >= 0 ? left : 0;
    c->filt_tile_left = tile_left >= 0 ? tile_left : 0;
    c->filt_info[2] = n;
    c->filt_info[3] = m;
    if (m > max_points) return set_err(c, RSM_E_INVALID, "rsm_filter_last_cloud: %lld points survive, capacity %lld", (long long)m, (long long)max_points);
    if (m > 0 && d_points) launch_pack_filtered16(c->xyz, c->bgr, dk, m, d_points, fs);
    if (m > 0 && d_normals) HIPCHK(c, hipMemcpyAsync(d_normals, dn, sizeof(float4) * (size_t)m, hipMemcpyDeviceToDevice, fs));
    *n_kept = m;
    if (fs != c->stream) HIPCHK(c, hipStreamSynchronize(fs));
    return finish(c, t);
}

extern "C" int rsm_filter_last_normals_info(rsm_ctx *c, int64_t info[2]) {
    if (!c || !info) return RSM_E_INVALID;
    info[0] = c->filt_normals[0];
    info[1] = c->filt_normals[1];
    return RSM_OK;
}

extern "C" int rsm_filter_last_info(rsm_ctx *c, int64_t info[4]) {
    if (!c || !info) return RSM_E_INVALID;
    memcpy(info, c->filt_info, sizeof c->filt_info);
    return RSM_OK;
}

extern "C" int rsm_filter_last_cloud_host(rsm_ctx *c, const rsm_filter_params *prm, rsm_point16 *h_points, float *h_normals,
                                          int64_t max_points, int64_t *n_kept, double *stats) {
    if (!c || !n_kept || !h_points || max_points < 0) return RSM_E_INVALID;
    if (!c->have_result) return set_err(c, RSM_E_STATE, "no result");
    HIPCHK(c, hipSetDevice(c->device));
    if (!c->pack16) {
        const int s = dalloc(c, &c->pack16, c->cap_px);
        if (s != RSM_OK) return s;
    }
    if (h_normals && !c->pack_nrm) {
        const int s = dalloc(c, &c->pack_nrm, c->cap_px * 4);
        if (s != RSM_OK) return s;
    }
    const int64_t cap = (int64_t)c->cap_px < max_points ? (int64_t)c->cap_px : max_points;
    const int s = rsm_filter_last_cloud(c, prm, c->pack16, h_normals ? c->pack_nrm : nullptr, cap, n_kept, stats);
    if (s != RSM_OK) return s;
    const size_t m = (size_t)*n_kept;
    if (m > 0) {
        HIPCHK(c, hipMemcpyAsync(h_points, c->pack16, m * sizeof(rsm_point16), hipMemcpyDeviceToHost, c->stream));
        if (h_normals) HIPCHK(c, hipMemcpyAsync(h_normals, c->pack_nrm, m * 4 * sizeof(float), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    return RSM_OK;
}

// ---- PLY writer (CStereoMatching.cpp:723-729, 754-756) ----------------------------------------------
extern "C" int rsm_write_ply(const char *path, const double *xyz, const uint8_t *bgr, int64_t n) {
    if (!path || n < 0 || (n > 0 && (!xyz || !bgr))) return RSM_E_INVALID;
    FILE *fp = fopen(path, "wb");
    if (!fp) return RSM_E_INVALID;
    fprintf(fp, "ply\n");
    fprintf(fp, "format binary_little_endian 1.0\n");
    fprintf(fp, "element vertex %d\n", (int)n);
    fprintf(fp, "property float x\nproperty float y\nproperty float z\nproperty uchar blue\nproperty uchar green\nproperty uchar red\n");
    fprintf(fp, "end_header\n");
    for (int64_t i = 0; i < n; i++) {
        const float p[3] = {(float)xyz[3 * i], (float)xyz[3 * i + 1], (float)xyz[3 * i + 2]}; // convertTo CV_32F, .cpp:754
        fwrite(p, sizeof(float), 3, fp);
        fwrite(bgr + 3 * i, 1, 3, fp);
    }
    const int ok = ferror(fp) == 0;
    fclose(fp);
    return ok ? RSM_OK : RSM_E_INVALID;
}

// the same file from the 16-byte records (float xyz + BGR are exactly a PLY vertex of this header)
extern "C" int rsm_write_ply16(const char *path, const rsm_point16 *points, int64_t n) {
    if (!path || n < 0 || (n > 0 && !points)) return RSM_E_INVALID;
    FILE *fp = fopen(path, "wb");
    if (!fp) return RSM_E_INVALID;
    fprintf(fp, "ply\n");
    fprintf(fp, "format binary_little_endian 1.0\n");
    fprintf(fp, "element vertex %d\n", (int)n);
    fprintf(fp, "property float x\nproperty float y\nproperty float z\nproperty uchar blue\nproperty uchar green\nproperty uchar red\n");
    fprintf(fp, "end_header\n");
    for (int64_t i = 0; i < n; i++) fwrite(&points[i], 1, 15, fp); // x, y, z, b, g, r (the pad byte stays behind)
    const int ok = ferror(fp) == 0;
    fclose(fp);
    return ok ? RSM_OK : RSM_E_INVALID;
}

// ---- NCC kernel microbenchmark --------------------------------------------------------------------
extern "C" int rsm_bench_ncc(rsm_ctx *c, int W, int H, int r, int cands, int iters, double *ms_per_launch) {
    if (!stage_ok(c, W, H) || r < 1 || r > 15 || cands < 1 || iters < 1 || !ms_per_launch) return RSM_E_INVALID;
    if (W <= 2 * r + cands + 2 || H <= 2 * r + 2) return RSM_E_INVALID;
    Tmp t(c);
    const size_t px = (size_t)W * H;
    std::vector<uint8_t> hi(px * 3), hm(px, 255);
    uint32_t s = 12345u;
    for (auto &v : hi) {
        s = s * 1664525u + 1013904223u;
        v = (uint8_t)(s >> 24);
    }
    MatchBufs b{};
    if (!setup_match(c, t, hi.data(), hi.data(), hm.data(), hm.data(), W, H, r, b)) return finish(c, t);
    rsm_boundary m{r, H - 1 - r, r, W - 1 - r, W - 2 * r, H - 2 * r};
    StageArgs a = one_dir(c, W, H, r, &m, &m);
    bind_match(a, b);
    std::vector<int16_t> hl(px), hr(px);
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            int L = x - cands / 2, R = L + cands - 1;
            if (L < r) { L = r; R = L + cands - 1; }
            if (R > W - 1 - r) { R = W - 1 - r; L = R - cands + 1; }
            hl[(size_t)y * W + x] = (int16_t)L;
            hr[(size_t)y * W + x] = (int16_t)R;
        }
    a.d[0].BL = t.up(hl.data(), px);
    a.d[0].BR = t.up(hr.data(), px);
    int16_t *dd = t.alloc<int16_t>(px);
    if (!t.ok) return finish(c, t);
    a.d[0].d16_in = a.d[0].d16_out = dd;
    launch_fill_i16(dd, px, (int16_t)NOMATCH, c->stream);
    launch_ncc_argmax(a, 1, c->stream); // warm-up (setup_match zeroed the counter)
    hipEvent_t e0, e1;
    HIPCHK(c, hipEventCreate(&e0));
    HIPCHK(c, hipEventCreate(&e1));
    HIPCHK(c, hipEventRecord(e0, c->stream));
    for (int i = 0; i < iters; i++) {
        (void)hipMemsetAsync(a.ncc_cnt, 0, sizeof(int), c->stream); // fresh wide-pixel counter per launch
        (void)hipMemsetAsync(a.tie_cnt, 0, 2 * sizeof(int), c->stream);
        (void)hipMemsetAsync(a.wrow, 0, sizeof(int32_t) * (size_t)H, c->stream);
        launch_ncc_argmax(a, 1, c->stream);
    }
    HIPCHK(c, hipEventRecord(e1, c->stream));
    HIPCHK(c, hipEventSynchronize(e1));
    float ms = 0;
    HIPCHK(c, hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    *ms_per_launch = (double)ms / iters;
    return finish(c, t);
}
