/*
 * rsm.h -- C ABI of librsm_mi355.so: the MI355X-native (gfx950 / HIP) drop-in for the
 * CStereoMatching pyramidal dense-stereo path of seed93/reconstruction.
 *
 * The reference has no FFI layer; the seam this library replaces is the C++ class
 * surface the orchestrator uses (reference file:line, relative to the reference root):
 *   CStereoMatching::Init(...)            reconstruction/CStereoMatching.h:47, .cpp:5-13
 *   CStereoMatching::MatchAllLayer()      reconstruction/CStereoMatching.h:48, .cpp:15-34
 *   public fields Q,R_final,T_final,margin[2],MatchBlockRadius,m_ws,m_offset,Verbose
 *                                         reconstruction/CStereoMatching.h:38-45
 *   downstream contract: per emitted point CCloudOptimization::InsertPoint(3x1 CV_64F) in
 *   row-major pixel order (.cpp:749-751), cam[pair][v].bound = margin[v] (.cpp:27-28).
 * include/CStereoMatchingMI355.hpp is a header-only C++ adapter with exactly that surface
 * built on these entry points; INTEGRATION.md shows the patch a maintainer would apply.
 *
 * All entry points return 0 on success or a negative rsm_status; the library never calls
 * exit() (the reference does at CStereoMatching.cpp:827-830 -> RSM_E_DEGENERATE_MARGIN).
 * Plain pointers and sizes only; no C++ or torch types cross this boundary.
 * One rsm_ctx per GPU; a ctx is not re-entrant (like CStereoMatching), different ctxs
 * may be driven from different threads.
 */
#ifndef RSM_H
#define RSM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RSM_NOMATCH (-10000) /* NOMATCH, reconstruction/CStereoMatching.h:9 */
#define RSM_MAX_LEVELS 12

typedef enum rsm_status {
    RSM_OK = 0,
    RSM_E_INVALID = -1,           /* bad argument */
    RSM_E_DEGENERATE_MARGIN = -2, /* YL>=YR || XL>=XR (reference: exit(0), .cpp:827-830) */
    RSM_E_HIP = -3,               /* HIP runtime error (rsm_last_error() has the text) */
    RSM_E_NOMEM = -4,
    RSM_E_STATE = -5,             /* call order (e.g. run before upload) */
    RSM_E_COMM = -6               /* RCCL missing or failed (rsm_comm_last_error() has the text) */
} rsm_status;

/* struct Boundary, reconstruction/CManageData.h:10-14 (same field order) */
typedef struct rsm_boundary {
    int YL, YR, XL, XR;
    int width, height;
} rsm_boundary;

/* Inputs of one stereo pair = what MatchAllLayer reads after Rectify()
 * (.cpp:20: cam[pair][v].image/.mask, Q, R_final, T_final) + the Init() parameters. */
typedef struct rsm_pair_in {
    const uint8_t *image[2]; /* rectified top-level BGR 8UC3, row-major, stride 3*width  */
    const uint8_t *mask[2];  /* rectified+eroded top-level mask 8UC1, stride width       */
    int width, height;       /* top level = m_LowestLevelSize * 2^(m_PyrmNum-1) (.cpp:120) */
    int pyr_levels;          /* m_PyrmNum                                                */
    int radius;              /* MatchBlockRadius (Init radii; CReconstruction.cpp:17: 2)  */
    double ws;               /* m_ws (CReconstruction.cpp:17: 0.03)                       */
    int offset;              /* m_offset (CStereoMatching.h:47: 2)                        */
    int origin_width;        /* m_OriginSize.width (scale of .cpp:692)                    */
    double Q[16];            /* 4x4 row-major, after the sign flip of .cpp:138            */
    double R_final[9];       /* 3x3 row-major (.cpp:132)                                  */
    double T_final[3];       /* (.cpp:133)                                                */
    int verbose;             /* Verbose (.cpp:12)                                         */
} rsm_pair_in;

/* Outputs of one pair. Buffers are caller-allocated; NULL skips that output.
 * ZERO-INITIALISE the struct (memset / = {0}) before filling it in: the library reads EVERY pointer member, and members are
 * appended as the ABI grows -- `points16` came with ABI 2 (RSM_ABI_VERSION / rsm_abi_version(); a caller compiled against
 * the ABI 1 header, whose struct ends at v_top, must be rebuilt: an ABI 2 library reads 8 bytes beyond its struct). */
#define RSM_ABI_VERSION 2
typedef struct rsm_pair_out {
    double *disparity[2];   /* width*height fp64 each: disparity[0|1] after the last level  */
    rsm_boundary margin[2]; /* margin[0|1] at the top level (-> cam[pair][v].bound)         */
    int64_t n_points;       /* points DisparityToCloud emits (.cpp:732-759)                  */
    int64_t max_points;     /* capacity of xyz / bgr in points                               */
    double *xyz;            /* 3*max_points fp64: R_final*X + T_final, InsertPoint order     */
    uint8_t *bgr;           /* 3*max_points: colour of imagePyrm[top][0] (.cpp:756)          */
    int64_t v_top;          /* masked view-0 pixels inside margin[0] at the top level        */
    struct rsm_point16 *points16; /* max_points 16-byte records {float x, y, z; u8 b, g, r, pad}: the cloud as
                             * CCloudOptimization::InsertPoint keeps it (the fp64 point cast to float,
                             * CloudOptimization/CCloudOptimization.cpp:61) + the colour, packed on the GPU --
                             * 16 instead of 27 bytes per point over PCIe; NULL skips it (xyz / bgr likewise)  */
} rsm_pair_out;

typedef struct rsm_ctx rsm_ctx;

/* ---- lifetime --------------------------------------------------------------------------- */
int rsm_create(rsm_ctx **ctx, int hip_device);
void rsm_destroy(rsm_ctx *ctx);
const char *rsm_last_error(const rsm_ctx *ctx);
const char *rsm_version(void);
/* the ABI the library was built with (struct layouts of this header): compare with RSM_ABI_VERSION after loading it */
int rsm_abi_version(void);
/* GPUs visible to this process (hipGetDeviceCount; 0 when there is none or no HIP runtime) */
int rsm_device_count(void);

/* ---- one pair: host buffers in, host buffers out (replaces the loop body .cpp:20-31) ----- */
int rsm_match_pair(rsm_ctx *ctx, const rsm_pair_in *in, rsm_pair_out *out);

/* ---- the same, split so inputs can stay resident in HBM --------------------------------- */
/* H2D of the two rectified images + masks; (re)sizes the ctx workspace. */
int rsm_upload_pair(rsm_ctx *ctx, const rsm_pair_in *in);
/* Same, but image[]/mask[] are DEVICE pointers on this ctx's GPU (copied device-to-device). */
int rsm_upload_pair_device(rsm_ctx *ctx, const rsm_pair_in *in);
/* ConstructPyrm + all MatchOneLayer levels + DisparityToCloud on the resident pair.
 * Synchronous with respect to the host on return. */
int rsm_run_pair(rsm_ctx *ctx);
/* D2H of the results of the last rsm_run_pair. */
int rsm_download_pair(rsm_ctx *ctx, rsm_pair_out *out);
/* Page-locked host memory for the buffers of rsm_pair_in / rsm_pair_out (the reference keeps them in pageable cv::Mat
 * memory, CManageData.cpp:75-78 / CStereoMatching.cpp:21-31: a cv::Mat can wrap memory from rsm_host_alloc, or the memory
 * it already owns can be page-locked in place).  Downloads into page-locked memory run at the link's rate instead of through
 * the runtime's staging copy (a C2 pair's two fp64 maps + cloud: ~7 instead of 34 ms).  NULL / RSM_E_HIP on failure. */
void *rsm_host_alloc(size_t bytes);
void rsm_host_free(void *p);
int rsm_host_register(void *p, size_t bytes);
int rsm_host_unregister(void *p);
/* Device pointers of the last results (valid until the next upload/run/destroy):
 * fp64 disparity maps, n_points, packed cloud xyz (fp64 x3) and bgr (u8 x3). */
int rsm_result_device(rsm_ctx *ctx, const double **disparity0, const double **disparity1,
                      int64_t *n_points, const double **xyz, const uint8_t **bgr);

/* Copies the cloud of the last run device-to-device into caller-owned device buffers (e.g. the
 * buffers an RCCL gather sends from): up to max_points points, xyz fp64 x3 and/or bgr u8 x3. */
int rsm_export_cloud_device(rsm_ctx *ctx, double *d_xyz, uint8_t *d_bgr, int64_t max_points);

/* ---- several pairs at once (the pair loop of MatchAllLayer, .cpp:17-33, has no cross-pair data flow) -------- */
/* rsm_run_pair on n DIFFERENT contexts concurrently (one host thread each; contexts may share a GPU or sit on
 * different ones).  On one GPU two pairs in flight hide each other's launch-latency-bound small levels and host
 * syncs.  Returns the first non-zero status. */
int rsm_run_pairs(rsm_ctx *const *ctxs, int n);
/* Same, every context running its resident pair `repeats` times back to back without meeting the others in between
 * (steady-state throughput of a stream of pairs; bench.py). */
int rsm_run_pairs_repeat(rsm_ctx *const *ctxs, int n, int repeats);
/* The whole loop body for n_pairs pairs over a pool of contexts: a context takes the next pair from a queue as soon as
 * it is free (upload, run, download -- one pair's PCIe copies overlap another pair's kernels).  out[p] is filled as
 * by rsm_match_pair, in pair order, so the caller can replay InsertPoint / filter(CamPair) sequentially afterwards.
 * status (optional, n_pairs ints) receives each pair's status; a failed pair does not stop the others. */
int rsm_match_pairs(rsm_ctx *const *ctxs, int n_ctx, const rsm_pair_in *in, rsm_pair_out *out, int n_pairs, int *status);
/* Same with library-owned contexts: pairs sharded over n_gpus devices of this node (0 = all visible),
 * pairs_in_flight contexts per device (0 = 2) -- the single-process form of SURVEY 8(e); the one-process-per-GPU form
 * is rsm_gather_clouds below. */
int rsm_match_pairs_multi_gpu(const rsm_pair_in *in, int n_pairs, int n_gpus, int pairs_in_flight, rsm_pair_out *out,
                              int *status);

/* ---- multi-GPU, one process per GPU: RCCL gather of the per-pair clouds -------------------------- */
/* The record that travels: what CCloudOptimization::InsertPoint keeps of a point (CloudOptimization/
 * CCloudOptimization.cpp:61: the fp64 point cast to float) plus the colour of imagePyrm[top][0] (.cpp:756). */
typedef struct rsm_point16 {
    float x, y, z;
    uint8_t b, g, r, pad;
} rsm_point16;
/* Packs the cloud of the last rsm_run_pair into 16-byte records in a caller-owned DEVICE buffer (capacity
 * max_points); *n_points receives the number written.  This is the send buffer of rsm_gather_clouds. */
int rsm_pack_cloud16(rsm_ctx *ctx, rsm_point16 *d_dst, int64_t max_points, int64_t *n_points);

#define RSM_COMM_ID_BYTES 128   /* = NCCL_UNIQUE_ID_BYTES */
#define RSM_COMM_MAX_PAIRS 4096 /* pairs per gather */
typedef struct rsm_comm rsm_comm;
/* Rank 0 makes the id (ncclGetUniqueId) and hands its bytes to the other ranks by whatever launcher started them
 * (MPI, a file, torch.distributed's store, ...); then every rank creates its communicator on its GPU. */
int rsm_comm_unique_id(char id[RSM_COMM_ID_BYTES]);
int rsm_comm_create(rsm_comm **comm, const char id[RSM_COMM_ID_BYTES], int rank, int world, int hip_device);
/* The gather protocol is written against this small transport table; rsm_comm_create fills it with RCCL.  A pipeline
 * that already owns a transport (MPI, its own RCCL communicator) -- and the protocol tests, which run world 2/3/8 over
 * an in-process mock on host memory -- hand in theirs.  All functions return 0 or non-zero (failure); `buf` pointers
 * are whatever memory the caller of rsm_gather_clouds passes (device memory for RCCL).
 *   allreduce_sum_i64  in-place sum of n int64 in HOST memory over all ranks (collective)
 *   group_begin/_end   bracket the payload exchange; everything posted in between is complete when group_end returns
 *   send / recv        point-to-point; between one (sender, receiver) pair they match IN POSTING ORDER (RCCL's rule)
 *   copy               local copy (the root's own pairs) */
typedef struct rsm_transport {
    void *self;
    int (*allreduce_sum_i64)(void *self, int64_t *host_buf, int n);
    int (*group_begin)(void *self);
    int (*send)(void *self, const void *buf, uint64_t bytes, int peer);
    int (*recv)(void *self, void *buf, uint64_t bytes, int peer);
    int (*copy)(void *self, void *dst, const void *src, uint64_t bytes);
    int (*group_end)(void *self);
} rsm_transport;
int rsm_comm_create_transport(rsm_comm **comm, const rsm_transport *transport, int rank, int world);
void rsm_comm_destroy(rsm_comm *comm);
const char *rsm_comm_last_error(const rsm_comm *comm);
/* Fan-in of the clouds of all pairs to rank `root` -- the replacement of the global `cloud_in += cloud` accumulation
 * (CCloudOptimization.cpp:61,123) when pairs are sharded one process per GPU.  Every rank passes its n_local clouds
 * (device buffers of 16-byte records, their pair ids in [0, n_pairs_total) IN ANY ORDER, and point counts).  On the
 * root the clouds of ALL pairs arrive in d_out (device, capacity max_out records) in pair order; out_offsets (host,
 * n_pairs_total + 1) receives each pair's first record.  Collective: every rank of the communicator calls it, and
 * every rank returns the SAME status: a bad argument on one rank, a pair claimed by two ranks or a root buffer that
 * is too small make all ranks return RSM_E_INVALID before any payload moves (nobody is left waiting). */
int rsm_gather_clouds(rsm_comm *comm, int root, int n_local, const int *pair_ids, const rsm_point16 *const *d_clouds,
                      const int64_t *n_points, int n_pairs_total, rsm_point16 *d_out, int64_t max_out,
                      int64_t *out_offsets);
/* Counts only: every rank learns the point count of every pair (counts: n_pairs_total int64, host), e.g. to size the
 * root's buffer before rsm_gather_clouds.  Collective, same status on every rank, same checks as the gather. */
int rsm_gather_counts(rsm_comm *comm, int n_local, const int *pair_ids, const int64_t *n_points, int n_pairs_total,
                      int64_t *counts);
/* The transport-free core of rsm_gather_clouds, exposed for tests and for pipelines that post the transfers
 * themselves.  Step 1: every rank fills its contribution to the metadata vector (RSM_GATHER_META_WORDS(P) int64:
 * per pair the point count and the number of claims, the owner's rank, then an error count and the root's capacity);
 * the vectors are summed over the ranks.  Step 2: from the summed vector every rank derives the SAME verdict, the
 * pair offsets, and its own ordered list of transfers: a rank's sends ascend by pair id, and the root posts its
 * receives per peer in that same order -- the order in which point-to-point operations between two ranks match. */
#define RSM_GATHER_META_WORDS(P) (3 * (P) + 2)
typedef struct rsm_gather_op {
    int kind;          /* 0 = send to `peer`, 1 = receive from `peer`, 2 = local copy (root's own pair) */
    int peer;
    int pair;          /* pair id */
    int local_index;   /* index into the caller's local arrays (send / copy), -1 for a receive */
    int64_t offset;    /* first record of the pair in the root's output (receive / copy) */
    int64_t count;     /* records */
} rsm_gather_op;
int rsm_gather_meta_fill(int rank, int world, int root, int n_local, const int *pair_ids, const int64_t *n_points,
                         int n_pairs_total, int64_t max_out, int64_t *meta);
/* offsets: n_pairs_total + 1; ops: capacity max_ops (n_local + n_pairs_total always suffices), *n_ops written.
 * Returns RSM_OK, or RSM_E_INVALID when the summed metadata says the gather must not start (same on every rank). */
int rsm_gather_plan(int rank, int world, int root, int n_local, const int *pair_ids, const int64_t *n_points,
                    int n_pairs_total, const int64_t *meta_summed, int64_t *offsets, rsm_gather_op *ops, int max_ops,
                    int *n_ops);

/* Tuning / validation knobs.  None of them changes a result (every alternative path is held bit-identical by the tests):
 *   "ncc_bytes" = 1        the generic byte-wise NCC kernel instead of the dot4 one
 *   "wide_rows" = 0 | 1 | 2 | 3   rows of wide pixels: 0 (default) = chosen per row on the device (k_rg_rows: the sliding window
 *                          sums when the row's widest interval has at most ncc_slide_max candidates and the row holds enough
 *                          wide pixels, the int8 row GEMM on the matrix cores otherwise) / 1 = the one-workgroup-per-pixel kernel
 *                          only / 2 = every such row through the int8 row GEMM / 3 = every such row through the sliding sums;
 *                          "no_rowgemm" = 1 is wide_rows = 1
 *   "ncc_mid" / "ncc_slide_max"   which rows of long-interval pixels leave the band kernel (intervals longer than ncc_mid
 *                          candidates; 0 = by window size from the measured crossover) and which of them take the sliding sums
 *                          (widest interval <= ncc_slide_max, default 512) rather than the int8 row GEMM
 *   "no_exact" = 1         (timing A/B only) skip the reference-order re-evaluation of near-tie pixels
 *   "refine_skew_from" / "refine_skew_T" / "refine_skew_min_px" / "refine_skew_waves" / "refine_skew_rows"   time-skewed refine
 *                          sweeps (k_refine_skew): T (2..4, default 4) sweeps per launch from that sweep of a level on (default
 *                          22; 0 = never) at levels with at least min_px margin pixels per direction (default 1 M: the two
 *                          largest levels of a 12 MP pair; a level narrower than 80 columns never), aiming at `waves` workgroups
 *                          (default 2560: two rounds of the 1280 a chip holds) or `rows` rows per chunk
 *   "refine_skew_uw"       columns a strip of that kernel owns: 0 (default) = 66 - 2T, all that its last sweep can compute from 64
 *                          lanes; an even number below that (e.g. 56: every strip starts on a 128-byte line of the cache ways) for A/B
 *   "refine_skew_waves_alone"  the workgroups a time-skewed launch aims at while no other context of the device is inside
 *                          rsm_run_pair (default 3840; with pairs in flight `refine_skew_waves` applies); 0 = the same
 *   "refine_skew_prio"     p > 0: the time-skewed kernel's waves rotate their issue priority (s_setprio) every 2^p shader clocks, in step
 *                          over the whole chip, so that the workgroups of a CU -- which the hardware serves oldest first -- advance
 *                          alike; 0 (default) = off: it equalises them as designed and gains 7 % on a single round of workgroups,
 *                          nothing on the default two rounds
 *   "cu_share" = n         n > 1: the context's streams are confined to one of n equal shares of the compute units (the
 *                          context's creation ordinal on its device picks the share; measured slower than sharing the whole
 *                          chip in turns, profiles/LAB_NOTES.md 4); 0 / 1 = the whole chip.  The masked streams are BLOCKING streams
 *                          (hipExtStreamCreateWithCUMask has no non-blocking flag): legacy null-stream work of the process then
 *                          synchronises with them.  RSM_E_STATE while the context is inside rsm_run_pair
 *   "filter_list"          ... its list passes (a thread per query the tile pass left over, windows read from the lattice copy): bit 0
 *                          the 49 x 49 pass, bit 1 the 81 x 81 pass on what that leaves, bit 2 (needs bit 0 or a 24-pixel tile pass)
 *                          a wave per query over windows of 80, 160, 320 ... pixels for the few hundred those leave, then the
 *                          whole-chip search for the handful beyond -- no grid level at all; bits 3 / 4: the 49 x 49 / 81 x 81 pass in
 *                          that wave form too instead of a thread per query (coalesced reads of the lattice rows; on C2's cloud the
 *                          81 x 81 pass gains, 1.4 against 3.9 ms, the 49 x 49 pass does not); default 23, 0 = tile pass + grid ladder only
 *   "filter_wg_max"        ... the wave passes run four waves per query (the first pass over the window shared, a 4 096-entry selection list)
 *                          while at most this many queries are left: default 2048; 0 = always a wave per query (A/B; the same bits)
 *   "filter_normals_window" rsm_filter_last_cloud: the normals' radius search reads the pixel lattice (the k-nearest pass's copy, the removed
 *                          points blanked) while no point needs a window wider than this many pixels -- default 8, at most 40; beyond it, or
 *                          with 0, the filtered cloud is sorted into a grid of radius-cells as for a generic cloud (C2: 0.15 against 1.8 ms)
 *   "filter_low_priority"  1 (default): rsm_filter_last_cloud runs on a stream of the lowest priority the device offers -- with pairs in
 *                          flight the dispatcher then hands compute units to the other contexts' matching first (the adapter loop with
 *                          the filter inside: 36 C2 pairs, 6 in flight, 219 against 215 Mdisp/s; 5 in flight 210 against 196); 0: on the
 *                          context's own stream.  The environment variable RSM_FILTER_LOW_PRIORITY=0/1 sets the default of contexts
 *                          created afterwards (for callers that reach the library through the adapter only)
 *   "filter_window"        rsm_filter_last_cloud's pixel-window pass: 1 (default) radius from a sparse probe (remembered by the context:
 *                          probed again on every 8th call, for another k or image size, when it stops deciding 70 % of the queries and
 *                          after any "filter_*" option), 0 off (the generic grid
 *                          search decides every query), 7 / 12 / 16 / 20 / 24 that radius
 *   "shared_gpu" = 1       the caller's hint that other contexts use this context's GPU (pairs in flight): the lone-pair split
 *                          below is never used, whatever the library's own count says at the moment a level is enqueued;
 *                          rsm_run_pairs / rsm_match_pairs derive the same per call from their pool (the option stays as the caller
 *                          set it), RsmStereoAdapter sets it for its slots when it creates them
 *   "refine_prefill"       1 (default): the first sweep of a level also fills the second cache way (0: A/B)
 *   "refine_split"         1 (default): a pair that has the GPU to itself (no other context of the device inside rsm_run_pair,
 *                          no per-launch timing) runs the two directions of its time-skewed sections as separate launch chains
 *                          on its two streams (one's low-occupancy tail beside the other's head: one C2 pair 24.1 -> 23.0 ms);
 *                          not used with pairs in flight (measured slower there): the library counts the contexts of the device
 *                          that are inside rsm_run_pair when a level is enqueued, and "shared_gpu" rules it out altogether
 *   "heavy_exclusive" = 0 | 1 | 2   contexts sharing a GPU: no turns / the top level's refine sweeps take turns (default) /
 *                          every large level's; "heavy_min_px", "heavy_from_sweep" bound the sections that take turns;
 *                          "heavy_lanes" = 2 (default): a level's single-sweep part (fabric-bound) and its time-skewed part
 *                          (issue-bound) take turns separately, so one pair's may run beside the other kind of another pair's */
int rsm_set_option(rsm_ctx *ctx, const char *name, long long value);

/* ---- measurement ------------------------------------------------------------------------- */
/* Per-stage device time of the rsm_run_pair calls since the last rsm_profile_enable (which zeroes the counters),
 * measured with hipEvents on the ctx stream.
 * Enable before the run (on = 1: every stage and every 8th launch of the dominant kernel; on = 2: the latter only).
 * Stage names: rsm_profile_stage_name(i), i < rsm_profile_stage_count(). */
int rsm_profile_enable(rsm_ctx *ctx, int on);
int rsm_profile_stage_count(void);
const char *rsm_profile_stage_name(int stage);
/* ms[i] = summed device milliseconds of stage i, launches[i] = kernel launches in it,
 * bytes[i] = algorithmic bytes (SURVEY 8(d) model) those launches moved. */
int rsm_profile_get(rsm_ctx *ctx, double *ms, int64_t *launches, double *bytes);

/* ---- per-stage entry points (one direction; host buffers) for parity tests --------------- */
/* own/oth = margin[!IsZeroOne] / margin[IsZeroOne] of the reference functions. */
int rsm_stage_find_margin(rsm_ctx *ctx, const uint8_t *mask, int W, int H, int r, rsm_boundary *m);
int rsm_stage_pyr_down(rsm_ctx *ctx, const uint8_t *src, int W, int H, int channels, uint8_t *dst);
int rsm_stage_erode_ellipse(rsm_ctx *ctx, const uint8_t *mask, int W, int H, int ksize, uint8_t *dst255);
int rsm_stage_initial_match(rsm_ctx *ctx, const uint8_t *img_own, const uint8_t *img_oth,
                            const uint8_t *mask_own, const uint8_t *mask_oth, int W, int H, int r,
                            int offset, const rsm_boundary *own, const rsm_boundary *oth,
                            const double *parent /* NULL = lowest level */, int Wp, int Hp,
                            int16_t *disp);
int rsm_stage_smooth(rsm_ctx *ctx, int16_t *disp, int W, int H, const rsm_boundary *own);
int rsm_stage_order(rsm_ctx *ctx, int16_t *disp, int W, int H, const rsm_boundary *own);
int rsm_stage_uniqueness_pass_s16(rsm_ctx *ctx, int16_t *p, const int16_t *q, int W, int H,
                                  const rsm_boundary *own, const rsm_boundary *oth);
int rsm_stage_uniqueness_pass_f64(rsm_ctx *ctx, double *p, const double *q, int W, int H,
                                  const rsm_boundary *own, const rsm_boundary *oth);
int rsm_stage_set_boundary(rsm_ctx *ctx, const int16_t *disp, const uint8_t *mask_own, int W, int H,
                           const rsm_boundary *own, const rsm_boundary *oth, int16_t *BL, int16_t *BR);
int rsm_stage_rematch(rsm_ctx *ctx, const uint8_t *img_own, const uint8_t *img_oth,
                      const uint8_t *mask_own, const uint8_t *mask_oth, int W, int H, int r,
                      const rsm_boundary *own, const rsm_boundary *oth, int16_t *disp);
int rsm_stage_median(rsm_ctx *ctx, int16_t *disp, const uint8_t *mask_own, int W, int H,
                     const rsm_boundary *own);
int rsm_stage_refine(rsm_ctx *ctx, const int16_t *disp_in, const uint8_t *img_own,
                     const uint8_t *img_oth, int W, int H, int iterations, double ws,
                     const rsm_boundary *own, double *disp_out);
/* the specified exp(-t) of DisparityRefine's weights (CStereoMatching.cpp:665-666 call exp) on n values */
int rsm_stage_exp_neg(rsm_ctx *ctx, const double *t, int64_t n, double *out);
/* the same through the form the time-skewed refine kernel's common path evaluates (arguments below 512: no special-case code) */
int rsm_stage_exp_neg_small(rsm_ctx *ctx, const double *t, int64_t n, double *out);
/* DisparityRefine's two divisions (CStereoMatching.cpp:669,671) as the time-skewed kernel evaluates them on its common path --
 * the hardware's fp64 division sequence without its operand-scaling and fix-up steps -- beside the compiler's a / b, on n operand
 * pairs: the parity tests hold the two equal bit for bit over the operand range the kernel's guard admits */
int rsm_stage_div_unscaled(rsm_ctx *ctx, const double *a, const double *b, int64_t n, double *q_fast, double *q_ieee);
/* the cloud filter's square root (PCL's statistical outlier removal sums sqrt of float32 squared distances,
 * CCloudOptimization.cpp:25-61 via pcl::StatisticalOutlierRemoval) -- the compiler's correctly rounded sequence without its
 * denormal scaling -- against sqrtf on the n floats with bit patterns first_bits .. first_bits + n - 1: *mismatches = how many differ */
int rsm_stage_sqrt_check(rsm_ctx *ctx, uint32_t first_bits, int64_t n, int64_t *mismatches);
/* DisparityRefine's matching costs xi = (1 - arma::dot(vecL, vecR) / (normL * normR)) / 2 (CStereoMatching.cpp:624-629) of 3x3x3
 * windows, as the device restatements of the data term compute them (form 0: the first sweep's, 1: a lane per cache miss, 2: four
 * lanes per cache miss): out[c][((y-1) (W-2) + (x-1)) (W-2) + col] = xi(own column x, row y, other view's window left edge col + c),
 * c = 0..2, y in [1, H-1), x in [1, W-1), col in [0, W-3]; out holds 3 (H-2) (W-2)^2 doubles.  BGR images, W x H x 3. */
int rsm_stage_refine_xi(rsm_ctx *ctx, const uint8_t *img_own, const uint8_t *img_oth, int W, int H, int form, double *out);
int rsm_stage_cloud(rsm_ctx *ctx, const double *disp, const uint8_t *mask_org, const uint8_t *img_own,
                    int W, int H, const double *Q, double scale, const double *R_final,
                    const double *T_final, const rsm_boundary *own, double *xyz, uint8_t *bgr,
                    int64_t max_points, int64_t *n_points);

/* ---- Rectify (CStereoMatching::Rectify, reconstruction/CStereoMatching.cpp:117-168) -------- */
typedef struct rsm_rectify_in {
    double K[2][9];                   /* cam[pair][v].MatIntrinsics, row-major 3x3 (CManageData.cpp:59)   */
    double E[2][12];                  /* cam[pair][v].MatExtrinsics, row-major 3x4 (CManageData.cpp:60)   */
    int origin_width, origin_height;  /* m_OriginSize: size of the raw images (CManageData.cpp:68-69)     */
    int lowest_width, lowest_height;  /* m_LowestLevelSize                                                */
    int pyr_levels;                   /* m_PyrmNum                                                        */
    const uint8_t *image[2];          /* raw BGR images (cv::imread, .cpp:146), origin size, host memory  */
    const uint8_t *mask[2];           /* raw grey masks (.cpp:155), origin size, host memory              */
} rsm_rectify_in;

typedef struct rsm_rectify_out {
    double Q[16];                     /* after the sign flip of .cpp:138                                  */
    double R_final[9], T_final[3];    /* .cpp:132-133                                                     */
    double P[2][12];                  /* cam[pair][v].P after .cpp:143-145                                */
    int width, height;                /* largestSize (.cpp:120)                                           */
    uint8_t *image[2];                /* optional host copies of cam[pair][v].image, width*height*3       */
    uint8_t *mask[2];                 /* optional host copies of cam[pair][v].mask (eroded), width*height */
} rsm_rectify_out;

/* Rectifies one pair on the GPU (host fp64 stereoRectify, device maps / remap / mask erosion) and leaves the
 * rectified images resident exactly as rsm_upload_pair would, with Q / R_final / T_final set: rsm_run_pair
 * can follow directly.  radius / ws / offset / verbose are CStereoMatching::Init's parameters. */
int rsm_rectify_pair(rsm_ctx *ctx, const rsm_rectify_in *in, int radius, double ws, int offset, int verbose,
                     rsm_rectify_out *out);
/* cv::stereoRectify(K1, 0, K2, 0, (nx, ny), R, T, R1, R2, P1, P2, Q, flags = 0, alpha = -1) -- host only. */
int rsm_stereo_rectify(const double *K1, const double *K2, int nx, int ny, const double *R, const double *T,
                       double *R1, double *R2, double *P1, double *P2, double *Q);
/* stage entry points for the parity tests */
int rsm_stage_rect_map(rsm_ctx *ctx, const double *A, const double *R, const double *newA, int W, int H,
                       int16_t *map1, uint16_t *map2);
int rsm_stage_remap(rsm_ctx *ctx, const uint8_t *src, int Ws, int Hs, int channels, const int16_t *map1,
                    const uint16_t *map2, int W, int H, uint8_t *dst);
int rsm_stage_erode_gray(rsm_ctx *ctx, const uint8_t *src, int W, int H, int ksize, uint8_t *dst);

/* ---- cloud interchange ----------------------------------------------------- */
/* Writes the debug / interchange PLY of CStereoMatching::DisparityToCloud (.cpp:723-729 header, :754-756
 * records): binary_little_endian, per vertex float x,y,z (the fp64 point cast to float, .cpp:754) and uchar
 * blue,green,red.  Host-only (no GPU needed). Returns 0 or RSM_E_INVALID. */
int rsm_write_ply(const char *path, const double *xyz, const uint8_t *bgr, int64_t n_points);
/* the same file from 16-byte records (rsm_pair_out.points16: float xyz + BGR = one PLY vertex each) */
int rsm_write_ply16(const char *path, const struct rsm_point16 *points, int64_t n_points);

/* ---- per-pair cloud filter (CCloudOptimization::filter, CloudOptimization/CCloudOptimization.cpp:82-121) -- */
typedef struct rsm_filter_params {
    int sor_mean_k;        /* m_sor_meank  (CReconstruction.cpp:18: 100) */
    double sor_std_mul;    /* m_sor_stdThres (1) */
    double normal_radius;  /* m_mls_radius (2.5): NormalEstimation's search radius, .cpp:107 */
    float cam_center[3];   /* cam[pair][0].CamCenter (CManageData.cpp:61-62): the normals are turned toward it, .cpp:114-121 */
} rsm_filter_params;
/* StatisticalOutlierRemoval + radius-search PCA normals + the turn toward CamCenter on a cloud of n float points
 * (host buffers; PointXYZ order = InsertPoint order).  kept_index (capacity n) receives the indices of the points
 * that survive the outlier removal, in order; normals (capacity 4 * n floats) their (nx, ny, nz, curvature).
 * stats (optional, 4 doubles): mean, stddev, threshold of the mean-neighbour distances, points searched exhaustively. */
int rsm_filter_cloud(rsm_ctx *ctx, const float *xyz, int64_t n, const rsm_filter_params *params, int32_t *kept_index,
                     float *normals, int64_t *n_kept, double *stats);
/* The same on the cloud of the last rsm_run_pair, without leaving the GPU: the surviving points as 16-byte records
 * (the RCCL payload of rsm_gather_clouds, now without the outliers) and their normals (4 floats each, may be NULL)
 * in caller-owned DEVICE buffers of capacity max_points. */
int rsm_filter_last_cloud(rsm_ctx *ctx, const rsm_filter_params *params, rsm_point16 *d_points, float *d_normals,
                          int64_t max_points, int64_t *n_kept, double *stats);
/* What the last rsm_filter_last_cloud[_host] of this context did: info[0] = the radius (pixels) of the pixel-window k-nearest pass,
 * 0 when it did not run (the cloud of a matched pair is a depth map: the k nearest of most points lie within a small pixel window
 * around their own pixel, proven per point by a bound on the distance to every ray outside it -- csrc/k_filter.hip; the radius --
 * 7, 12 or 16 -- comes from a sparse probe), info[1] = queries it left to the generic grid search, info[2] = points in, info[3] =
 * points kept.  Option "filter_window" (rsm_set_option): 1 = probe (default), 0 = no window pass, 7 / 12 / 16 / 20 / 24 = that
 * radius (A/B; the results are the same bits either way). */
int rsm_filter_last_info(rsm_ctx *ctx, int64_t info[4]);
/* ... and its normals: info[0] = the pixel window (radius) their radius search ran over on the cloud's pixel lattice, 0 when it ran on a
 * grid of radius-cells over the filtered cloud instead; info[1] = the widest window any point needed (the search radius in pixel
 * spacings at the nearest point: a property of the rig), -1 when the lattice was not available.  Option "filter_normals_window". */
int rsm_filter_last_normals_info(rsm_ctx *ctx, int64_t info[2]);
/* The same with HOST output buffers (page-locked ones from rsm_host_alloc arrive at the link's rate): what a pipeline that
 * replaces the first half of CCloudOptimization::filter (CCloudOptimization.cpp:82-121) downloads instead of the raw cloud --
 * the surviving points and their oriented normals (the reference's cloud_normal, :110-121).  h_normals may be NULL. */
int rsm_filter_last_cloud_host(rsm_ctx *ctx, const rsm_filter_params *params, rsm_point16 *h_points, float *h_normals,
                               int64_t max_points, int64_t *n_kept, double *stats);

/* ---- kernel microbenchmark (MDE/s: pixel x candidate NCC evaluations) -------------------- */
/* Runs the NCC interval-argmax kernel `iters` times on a resident level-sized problem with
 * `cands` candidates per pixel and returns average milliseconds per launch. */
int rsm_bench_ncc(rsm_ctx *ctx, int W, int H, int r, int cands, int iters, double *ms_per_launch);

#ifdef __cplusplus
}
#endif
#endif /* RSM_H */
