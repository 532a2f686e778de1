// rsm_dev.h -- internal declarations shared by the HIP translation units of librsm_mi355.so.
// gfx950 (MI355X) only: wave = 64 lanes, no portability layer.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#define NOMATCH (-10000) // reconstruction/CStereoMatching.h:9
#define WAVE 64
#define RF_MAX_SWEEPS 1024
#define RF_NOKEY ((int16_t)-32768)
#define RF_NOKEY2 0x80008000u // both ways of DirArgs::rf_key empty
#define RF_PPT 4      // pixels per thread of the refine sweep kernel
#define SBV_S 32 // SetBoundary: row segments per column of the vertical sweeps
// int32 scratch of launch_set_boundary (in rf_list), per direction
#define SETB_SCRATCH(W) ((size_t)(SBV_S * 6 + 2) * (size_t)(W))

// Margin of one view at one level (struct Boundary, CManageData.h:10-14, without width/height).
struct Mg {
    int YL, YR, XL, XR;
};

// Everything one direction of one stage may need. Direction v: own view = v, other = 1-v
// (IsZeroOne = (v == 0) in the reference's signatures).
struct DirArgs {
    const uint8_t *img_own, *img_oth;   // BGR interleaved, stride 3*W
    const uint32_t *img4_own, *img4_oth; // BGRX copies (one dword per pixel), stride W
    const uint8_t *mask_own, *mask_oth; // stride W
    const int32_t *S1_own, *S2_own;     // (2r+1)^2*3 window sums of bytes / squared bytes, centre-indexed
    const int32_t *S1_oth, *S2_oth;
    Mg own, oth;
    int16_t *d16_in, *d16_out; // int16 disparity maps (in may alias out when the stage allows it)
    int16_t *BL, *BR;          // candidate intervals (absolute columns)
    const double *parent;      // fp64 disparity of level k-1 (this direction)
    const int32_t *parent_nv;  // next-valid-column table of `parent`
    double *f64_a, *f64_b;     // fp64 disparity ping-pong / uniqueness maps
    uint32_t *rf_key;          // refine cache, per pixel: iMatch - x = int(d - 1.5) of the entry in way 0 (low half) | way 1 (high half), RF_NOKEY = empty
    double2 *rf_ent;           // refine cache, [way * rf_stride + pixel]: the data term (pwp, delta = pdp - dCenter) of that iMatch
};

// A data-term cache entry computed inside a k_refine_skew launch, applied to the cache by k_refine_apply after it.
struct RfUpd {
    uint32_t pix; // pixel index | direction << 31
    int32_t rel;  // iMatch - x of the entry (its parity selects the way)
    double pwp, delta;
};
#define RF_UPD_SHARDS 32 // append counters (one counter serialises at ~88 appends per microsecond)

// ints of StageArgs::wrow for a level of H rows: [0, 2H) wide pixels per (direction, row) appended by k_ncc_dot4; then two row lists of
// 2 (H + 1) ints (k_ncc_rowgemm's, k_ncc_slide's); then [2H] pixels with an interval longer than ncc_mid and [2H] the widest interval, per
// (direction, row), from k_ncc_rowstat
#define NCC_WROW_INTS(H) (10 * (H) + 32)
#define NCC_WROW_LIST(H, which) (2 * (H) + (which) * (2 * (H) + 2))
#define NCC_WROW_MID(H) (6 * (H) + 4)
#define NCC_WROW_MAX(H) (8 * (H) + 4)

struct StageArgs {
    DirArgs d[2];
    int ndir;    // 1 or 2 (gridDim.z)
    int W, H;    // this level
    int Wp, Hp;  // parent level
    int r;       // MatchBlockRadius
    int offset;  // m_offset
    double ws;   // m_ws
    int flag;    // stage-specific
    int flag2;   // refine: sweep index
    size_t rf_stride;  // refine: elements between the two cache ways (>= W*H)
    RfUpd *upd_list;   // refine (k_refine_skew): RF_UPD_SHARDS regions of upd_cap records
    int32_t *upd_cnt;  // [2][RF_UPD_SHARDS]: append counters, the set in use alternates per launch (flag3 & 1)
    int upd_cap;
    int flag3;         // refine: k_refine_first: bit 1 = no prefill of the second way; k_refine_skew: launch index (counter set)
    int skew_rows;     // refine (k_refine_skew): rows per chunk
    int skew_uw;       // refine (k_refine_skew): columns a strip owns (<= 66 - 2T)
    int skew_prio;     // refine (k_refine_skew): issue priority rotates every 2^skew_prio shader clocks (0: never)
    int opt_ncc_bytes;              // force the generic byte-wise NCC kernel (A/B validation)
    int opt_no_exact;               // skip k_ncc_exact (timing A/B only: ties then follow the integer form)
    uint32_t *rf_list; // NCC: worklist of wide pixels (dir << 31 | pixel index); SetBoundary: segment-map scratch
    uint32_t *tie_list; // NCC: pixels (dir << 31 | pixel index) whose scan saw a (near) tie -> k_ncc_exact
    int32_t *tie_cnt;   // NCC: [0] entries of tie_list after an initial-match launch, [1] after a Rematch launch (zero on entry)
    int32_t *wrow;     // NCC: [dir * H + y] wide pixels of a row after k_ncc_dot4 (zero on entry); rows with many go to k_ncc_rowgemm
    int opt_no_rowgemm; // A/B: every wide pixel through k_ncc_wide
    int ncc_mid;        // NCC: in a row with many pixels of long intervals, every interval longer than this joins the row kernels (<= NCC_WIDE = 160, the band kernel's limit)
    int ncc_slide_max;  // NCC: rows whose widest interval is at most this take k_ncc_slide, the others k_ncc_rowgemm
    int32_t *ncc_cnt;  // NCC: [0] number of wide pixels in rf_list (zero before an initial-match launch), [16 + dir * H + y] Rematch pixels of a row
};

// ---- launchers (each enqueues on `st`, no sync) ----------------------------------------------
void launch_fill_i16(int16_t *p, size_t n, int16_t v, hipStream_t st);
void launch_fill_f64(double *p, size_t n, double v, hipStream_t st);
void launch_fill_i32(int32_t *p, size_t n, int32_t v, hipStream_t st);
void launch_pyr_down(const uint8_t *src, int W, int H, int C, uint8_t *dst, hipStream_t st);
// out4 = {XL, XR, YL, YR} initialised by the kernel launcher (inverted defaults, .cpp:1014-1017)
void launch_find_margin(const uint8_t *mask, int W, int H, int r, int *out4, hipStream_t st);
void launch_find_margin_batch(int n, const uint8_t *const *masks, const int *Ws, const int *Hs, int r, int *out4, int *h_init,
                              hipStream_t st);
void launch_bgr_to_bgrx(const uint8_t *img, int W, int H, uint32_t *out, hipStream_t st);
void launch_box_sums(const uint32_t *img4, int W, int H, int r, int32_t *tmp1, int32_t *tmp2,
                     int32_t *S1, int32_t *S2, hipStream_t st);

void launch_next_valid(const double *parent, int Wp, int Hp, int32_t *nv, hipStream_t st);
void launch_next_valid2(const double *parent0, const double *parent1, int Wp, int Hp, int32_t *nv0, int32_t *nv1, hipStream_t st);
void launch_hl_interval(const StageArgs &a, hipStream_t st);
// mode 0: lowest level (interval = other margin), 1: interval arrays BL/BR into NOMATCH-filled out,
// 2: rematch (only pixels whose d16_in is NOMATCH; in place; the pixels come from launch_set_boundary(.., true))
void launch_ncc_argmax(const StageArgs &a, int mode, hipStream_t st);

void launch_smooth(const StageArgs &a, hipStream_t st, bool copy_outside = true); // d16_in -> d16_out
void launch_order(const StageArgs &a, hipStream_t st);         // d16_in in place
void launch_uniq_s16(int16_t *p, const int16_t *q, int W, int H, Mg own, Mg oth, hipStream_t st);
void launch_uniq_f64(double *p, const double *q, int W, int H, Mg own, Mg oth, hipStream_t st);
// d16_in, mask_own -> BL, BR; emit_list: also fill rf_list / ncc_cnt with the pixels Rematch has to evaluate
void launch_set_boundary(const StageArgs &a, hipStream_t st, bool emit_list = false);
void launch_median(const StageArgs &a, hipStream_t st);        // d16_in -> d16_out (pre-filled NOMATCH)
void launch_exp_neg(const double *t, double *out, long long n, hipStream_t st, int small_form); // the specified exp(-t) (tests)
// k_refine_skew's division without operand scaling beside the compiler's a / b (tests)
void launch_refine_xi(const uint32_t *A, const uint32_t *B, int W, int H, int form, double *out, hipStream_t st); // the data term's matching costs (tests)
void launch_sqrt_check(unsigned int first, long long n, unsigned long long *mismatches, hipStream_t st); // k_filter.hip
void launch_div_unscaled(const double *a, const double *b, double *q_fast, double *q_ieee, long long n, hipStream_t st);
void launch_refine_init(const StageArgs &a, hipStream_t st);   // d16_in -> f64_a, f64_b, cache reset
// f64_a -> f64_b; ev0/ev1 (optional) are recorded right around the light sweep kernel
void launch_refine_sweep(const StageArgs &a, hipStream_t st, hipEvent_t ev0 = nullptr, hipEvent_t ev1 = nullptr);
// T (2..4) sweeps f64_a -> f64_b in one time-skewed launch (a.flag3 = launch index, a.skew_rows, a.skew_uw) + the cache-update launch
void launch_refine_apply(const StageArgs &a, hipStream_t st); // k_refine.hip
void launch_refine_skew(const StageArgs &a, int T, hipStream_t st, hipEvent_t ev0 = nullptr, hipEvent_t ev1 = nullptr); // k_refine_skew.hip
int refine_skew_strips(const Mg &m, int T, int uw); // strips of 64 lanes a direction's interior takes (k_refine_skew.hip)
bool refine_skew_fits(const StageArgs &a);          // the kernel's flat over-reads stay inside the level

// cloud: returns nothing; *d_npoints (device int64) receives the point count; `flags` = W*H bytes of scratch,
// `blk` = launch_bad_blocks' map (CLOUD_BLOCKS(W,H) bytes)
#define CLOUD_BLOCKS(W, H) ((size_t)(((W) + 31) / 32) * (size_t)(((H) + 31) / 32))
void launch_bad_prefix(const uint8_t *mask, int W, int H, int32_t *prefix, hipStream_t st);
void launch_erode_binary(const int32_t *prefix, int W, int H, int ksize, const int *d_j1, const int *d_j2,
                         uint8_t *dst255, hipStream_t st);
void launch_bad_blocks(const int32_t *prefix, int W, int H, uint8_t *blk, hipStream_t st);
void launch_cloud(const double *disp, const int32_t *bad_prefix, const uint8_t *img, int W, int H, int ksize,
                  const int *d_j1, const int *d_j2, const double *q16_scaled, const double *R,
                  const double *T, Mg own, uint8_t *flags, const uint8_t *blk, int32_t *row_count, int64_t *row_offset,
                  int64_t *d_npoints, double *xyz, uint8_t *bgr, int64_t max_points, hipStream_t st);
void launch_pack_cloud16(const double *xyz, const uint8_t *bgr, int64_t n, void *dst16, hipStream_t st);
void launch_count_masked(const uint8_t *mask, int W, int H, Mg m, unsigned long long *d_count, hipStream_t st);

// Rectify (k_rectify.hip)
void launch_rect_map(const double *ir, double fx, double fy, double u0, double v0, int W, int H, int16_t *map1,
                     uint16_t *map2, hipStream_t st);
void launch_remap(const uint8_t *src, int Ws, int Hs, int C, const int16_t *map1, const uint16_t *map2, int W, int H,
                  uint8_t *dst, hipStream_t st);
// st: ceil(log2(ksize))+1 planes of W*H bytes of scratch
void launch_erode_gray(const uint8_t *src, int W, int H, int ksize, const int *d_j1, const int *d_j2, uint8_t *st,
                       uint8_t *dst, hipStream_t st_);

// cloud filter (k_filter.hip): SOR + radius normals on a device cloud of n float xyz points
// FilterArena: grow-only device scratch owned by the context (one hipMalloc per cloud size instead of dozens per call)
struct FilterArena;
FilterArena *filter_arena_create();
void filter_arena_destroy(FilterArena *a);
size_t filter_arena_bytes(int64_t n);                      // scratch one filter call needs for n points
int filter_arena_reserve(FilterArena *a, size_t bytes);    // makes room, rewinds the arena
void *filter_arena_alloc(FilterArena *a, size_t bytes);    // caller buffers that live across the call (nullptr: full)
// The cloud as the depth map it is (rsm_filter_last_cloud): which pixels of the top level's margin box emitted a point (k_cloud's
// flags + per-row offsets: point index = compaction order), the fp64 points, and the geometry that bounds how far a point outside
// a pixel window can be (k_filter.hip: k_sor_window).  R_final must be a rotation (the caller checks).
struct FilterLattice {
    const uint8_t *flags;
    const int64_t *row_offset;
    int W, XL, XR, YL, YR;
    const double *xyz64;
    double qz;          // Q[2][3] * scale (CStereoMatching.cpp:698)
    double R[9], T[3];  // R_final, T_final (host copies)
    int radius;         // window radius in pixels: 7, 12, 16, 20 or 24; 0 = chosen from a sparse probe (7 / 12 / 16, or no window pass)
    int *radius_out;    // optional: the radius used (0: the probe found no window worth its pass)
    int list_pass;      // 1: what the tile pass leaves over gets a 24-pixel window a thread each before the grid ladder
    int *undecided_out; // optional: queries the window passes left to the grid ladder
    int *tile_left_out; // optional: queries the tile pass alone left over
    int wg_max;         // the wave passes take four waves per query (k_sor_window_wg) while at most this many queries are left (default 2048)
    int normals_wmax;   // the normals' radius search on the lattice copy while no point needs a window wider than this (0: always on a grid of radius-cells)
    int *normals_out;   // optional, 2 ints: the window the normals used (0: the grid), the widest window a point needed
};
size_t cloud_lattice_bytes(int XL, int XR, int YL, int YR);
int filter_cloud_device(FilterArena *a, const float *d_xyz, int64_t n, int mean_k, double std_mul, double normal_radius,
                        const float cam_center[3], int32_t *d_kept_index, float *d_fxyz, float4 *d_normals, int64_t *n_kept,
                        double stats[4], hipStream_t st, const FilterLattice *pre = nullptr);
void launch_f64_to_f32x3(const double *src, int64_t n, float *dst, hipStream_t st);
void launch_pack_filtered16(const double *xyz, const uint8_t *bgr, const int32_t *kept, int64_t m, void *dst16, hipStream_t st);
