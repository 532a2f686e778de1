// k_refine_skew.hip -- DisparityRefine's settled Jacobi sweeps (CStereoMatching.cpp:590-678), T per launch, time-skewed
// down the rows: the path's dominant kernel (round 6 restatement; the values are those of T single sweeps, bit for bit).
//
// A workgroup of T waves owns a strip of 64 lanes (lane i = column x0 + i) and a chunk of rows and streams down it.  In
// step s wave t (= sweep t of the launch, "level" t) advances row r = s - 2t + 1; the levels are two rows apart, so
// inside a step no wave reads what another one writes: ONE workgroup barrier per step.  Level t's rows live in a ring of
// four rows in LDS (level 0 = the staged input); the data-term cache rows (both ways' 16-byte entries + the key dword) of
// the 2T rows in flight sit beside them.  Level t can compute the lanes [t - 1, 64 - t] (level 1 reads the columns x0 - 1 and
// x0 + 64 from the staged input row, which is 68 columns wide), so a strip owns 66 - 2T columns; rows likewise shrink by
// one per level from the staged rows.  Everything else copies through, exactly as the reference leaves margin-border
// pixels and NOMATCH pixels alone (.cpp:592,608,613).
//
// What changed against rounds 3-5's kernel, and why (profiles/LAB_NOTES.md has the measurements):
//  * Staging is ONE 16-byte-per-lane load and ONE 16-byte LDS write per wave and step: the cache is laid out as 16-byte
//    (pwp, delta) records per way and one key dword per pixel (rsm_dev.h), so a row is four streams -- state (34 lanes),
//    keys (16 lanes), way 0, way 1 (64 lanes each) -- and wave w carries stream w.  Every wave does the same work per step
//    (no staging wave for the others to wait for at the barrier), 4 loads and 4 LDS writes per row instead of 7 and 5, and a
//    stream's row pointer advances by one scalar add.  Strips start on 8-column boundaries where the pitch allows it, so a
//    way's 1 KB row segment is whole 128-byte lines.
//  * The loop is unrolled by the rings' period: every ring position is an immediate offset from one per-wave LDS address
//    (ring j stores row y at slot (y + 2j - 1 - s0) & 3, which makes the slots functions of the step alone).
//  * The common row -- every live pixel has its four neighbours and hits the cache -- is straight-line code on all lanes
//    with nothing rare in it: both divisions run without the hardware sequence's operand scaling (div_unscaled's
//    operations, the two reciprocals started as soon as their denominators exist), the exp in its t < 512 form.  Whether a
//    lane's result may be kept (exp arguments <= 200, numerators not tiny, pwp != 0) is decided by lane masks BESIDE the
//    chain and looked at only after the row's result has been written: a rare row is then redone by skew_rare_row (not
//    inlined: the miss service's data-term routine and the general update never share registers or schedules with the
//    common path), which serves the cache misses, runs the reference's general update on exactly the lanes that need it
//    and overwrites their results before the step's barrier.
// Cache misses (rare once the iteration has settled, which is when this kernel is used) are served four lanes per entry;
// the new entry lives in the LDS copy for the rest of the launch and goes to the update list that k_refine_apply scatters
// afterwards (a neighbouring strip may be staging the same pixel's entry at that moment: an in-place write could be read
// torn).  At most one record per (pixel, way) and launch, so k_refine_apply never sees two writers of a slot.
#include "refine_common.h"

typedef unsigned long long u64;
typedef int skew_v2i __attribute__((ext_vector_type(2)));

template <int T>
struct SkewLds { // byte offsets into the workgroup's LDS block
    static constexpr int NE = 2 * T;           // rows of cache entries resident: row y is written at the end of step y and last read in step y + 2T - 1
    static constexpr int RPB = 68 * 8;         // a ring row: columns x0 - 2 .. x0 + 65
    static constexpr int RING = 4 * RPB;       // a level's ring
    static constexpr int OFF_D = 0;            // [T][4][68] doubles
    static constexpr int OFF_ENT = OFF_D + T * RING; // [NE][2][64] double2
    static constexpr int OFF_KEY = OFF_ENT + NE * 2048; // [NE][64] dwords
    static constexpr int OFF_EMIT = OFF_KEY + NE * 256; // [NE][2] lane masks: cache slots this launch already listed a new entry for
    static constexpr int OFF_EXP = OFF_EMIT + NE * 16;  // the specified exp's table (2 KB)
    static constexpr int OFF_ML = OFF_EXP + 2048;       // [T][64] bytes: a row's missing lanes, per wave
    static constexpr int SIZE = OFF_ML + T * 64;        // T = 4: 29 568 B -- five workgroups per CU (the limit is 32 000 B, not 32 768)
};

// what the rare path needs, wave-uniform (built only when it is taken)
struct SkewRare {
    const uint32_t *A, *B; // BGRX images, own / other view
    RfUpd *upd;            // this workgroup's shard of the update list
    int32_t *cnt;          // its append counter
    double *gout;          // level T: the output row's lane-0 address; nullptr below
    char *lds;             // the workgroup's LDS block (generic pointer)
    double ws;
    int W, H, upd_cap;
    int x0, r;             // lane 0's column, the row
    uint32_t rdC, rdN, rdS; // LDS offsets of lane 0's dC / dN / dS
    uint32_t wr;           // LDS offset of lane 0's result slot (levels below T)
    uint32_t ent, key, emit, ml, tab; // LDS offsets: the row's entries [2][64], keys [64], emit masks [2], this wave's list, the exp table
    uint32_t dir;          // direction << 31
    int owned_row;         // the row is one this workgroup owns (its new entries go to the update list)
};

// The rare row: cache misses, pixels with one valid neighbour pair (modes 1 / 2), results the common path's guards
// rejected.  `m_redo` = the lanes to recompute, `m_own` = the lanes this workgroup owns.  Reads its operands from LDS again,
// serves the misses four lanes per entry (16 entries per round), installs the new entries in the LDS copy of the cache row,
// lists those of owned pixels -- once per (pixel, way) and launch -- and writes the general update's result for the lanes.
__device__ __forceinline__ void skew_rare_row(SkewRare q, u64 m_redo, u64 m_own) {
    const int lane = (int)__lane_id();
    char *L = q.lds;
    const double NM = (double)NOMATCH;
    const double dC = *(const double *)(L + q.rdC + lane * 8), dW = *(const double *)(L + q.rdC + lane * 8 - 8), dE = *(const double *)(L + q.rdC + lane * 8 + 8);
    const double dN = *(const double *)(L + q.rdN + lane * 8), dS = *(const double *)(L + q.rdS + lane * 8);
    uint32_t kk = *(const uint32_t *)(L + q.key + lane * 4);
    const bool lv = (m_redo >> lane) & 1ull;
    const int mode = (int)(dE != NM && dW != NM) + (int)(dS != NM && dN != NM) * 2; // .cpp:620
    const int rel = (int)(dC - 1.5);                                                 // .cpp:625 (iMatch - x)
    const int way = rel & 1;
    const bool miss = lv && mode != 0 && (int)(int16_t)(kk >> (way << 4)) != rel;
    double2 *ent = (double2 *)(L + q.ent);
    const u64 mm = __ballot(miss);
    if (mm) { // wave-uniform
        uint8_t *mlist = (uint8_t *)(L + q.ml);
        const int n = __popcll(mm);
        const int rank = __popcll(mm & ((1ull << lane) - 1ull));
        if (miss) mlist[rank] = (uint8_t)lane;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        double2 pd = make_double2(0.0, 0.0);
#pragma unroll 1
        for (int done = 0; done < n; done += 16) { // 16 entries per round, four lanes each
            const int j = done + (lane >> 2);
            const int ml = mlist[j < n ? j : done]; // the lane this quad works for
            const int mrel = __shfl(rel, ml);
            const int mx = q.x0 + ml;
            double p, qq;
            refine_data_term_quad(q.A, q.B, q.W, q.H, mx, q.r, mrel + mx, lane & 3, p, qq);
            const double mp = __shfl(p, (rank & 15) << 2), mq = __shfl(qq, (rank & 15) << 2); // (every lane of a quad holds its result)
            if (miss && rank >= done && rank < done + 16) pd = make_double2(mp, mq);
        }
        if (miss) {
            ent[way * 64 + lane] = pd;
            kk = way ? ((kk & 0xffffu) | ((uint32_t)(rel & 0xffff) << 16)) : ((kk & 0xffff0000u) | (uint32_t)(rel & 0xffff));
            *(uint32_t *)(L + q.key + lane * 4) = kk;
        }
        u64 *emit_row = (u64 *)(L + q.emit);
        const u64 fl0 = emit_row[0], fl1 = emit_row[1]; // (one wave at a time works on a row slot)
        const bool emit = miss && q.owned_row && ((m_own >> lane) & 1ull) && !(((way ? fl1 : fl0) >> lane) & 1ull);
        const u64 em = __ballot(emit);
        if (em) {
            const int leader = __builtin_ctzll(em);
            int base = 0;
            if (lane == leader) base = atomicAdd(q.cnt, __popcll(em));
            base = __shfl(base, leader) + __popcll(em & ((1ull << lane) - 1ull));
            const bool listed = emit && base < q.upd_cap;
            if (listed) {
                RfUpd u;
                u.pix = (uint32_t)((size_t)q.r * q.W + q.x0 + lane) | q.dir;
                u.rel = rel;
                u.pwp = pd.x;
                u.delta = pd.y;
                q.upd[base] = u;
            }
            const u64 n0 = __ballot(listed && !way), n1 = __ballot(listed && way);
            if (lane == 0) {
                emit_row[0] = fl0 | n0;
                emit_row[1] = fl1 | n1;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    if (lv) {
        const double2 pd = ent[way * 64 + lane];
        const double val = mode == 0 ? dC /* .cpp:655 */ : refine_update(mode, dC, dE, dW, dN, dS, pd.x, pd.y, q.ws, (ExpTab)(L + q.tab));
        if (q.gout) {
            if ((m_own >> lane) & 1ull) q.gout[lane] = val;
        } else
            *(double *)(L + q.wr + lane * 8) = val;
    }
}

// .cpp:652-672 for a row whose live pixels all have four neighbours, straight-line on all lanes (the common row of the kernels
// in this file): the update `u`, the lanes whose `u` is the reference's as far as the masks known early in the chain say
// (`m_early`), and the lanes the rare path has to redo (`m_redo`).
__device__ __forceinline__ void skew_common_row(double dC, double dE, double dW, double dN, double dS, double2 pd, int crel, int rel, double ws, u64 m_try,
                                                ExpTab tab, double &u, u64 &m_early, u64 &m_redo) {
    const double ex = fabs(dE - dC) - fabs(dW - dC);
    const double ey = fabs(dS - dC) - fabs(dN - dC);
    const double tx = ex * ex, ty = ey * ey;
    double wx, wy;
    exp_neg2_small(tx, ty, wx, wy, tab); // .cpp:665-666 (the bits of exp_neg for t < 512)
    const double b2 = pd.x + ws;         // the second division's reciprocal needs pwp only: two Newton steps beside the exps
    double y2 = __builtin_amdgcn_rcp(b2);
    double e2 = __builtin_fma(-b2, y2, 1.0);
    y2 = __builtin_fma(y2, e2, y2);
    e2 = __builtin_fma(-b2, y2, 1.0);
    y2 = __builtin_fma(y2, e2, y2);
    const double b1 = 2 * (wx + wy);
    double y1r = __builtin_amdgcn_rcp(b1);
    double e1 = __builtin_fma(-b1, y1r, 1.0);
    y1r = __builtin_fma(y1r, e1, y1r);
    e1 = __builtin_fma(-b1, y1r, 1.0);
    y1r = __builtin_fma(y1r, e1, y1r);
    const double a1 = wx * (dE + dW) + wy * (dN + dS);
    const double q1 = a1 * y1r;
    const double ds = __builtin_fma(__builtin_fma(-b1, q1, a1), y1r, q1); // a1 / (2 (wx + wy)): div_unscaled's operations
    const double a2 = (dC + pd.y) * pd.x + ws * ds;
    const double q2 = a2 * y2;
    u = __builtin_fma(__builtin_fma(-b2, q2, a2), y2, q2); // .cpp:671
    m_early = m_try & RF_IEQ(crel, rel) & RF_FLE(fmax(tx, ty), 200.0) & RF_FGT(pd.x, 0.0);
    m_redo = m_try & ~(m_early & RF_FGT(fabs(a1), 0x1p-300) & RF_FGT(fabs(a2), 0x1p-300));
}

// which of a row's four streams (0 state, 1 keys, 2 / 3 the ways' entries; -1 none) wave w carries in slot j
template <int T>
__device__ __forceinline__ int skew_stream(int w, int j) {
    if (T == 4) return j == 0 ? w : -1;
    if (T == 3) return j == 0 ? (w == 0 ? 0 : w + 1) : (w == 0 ? 1 : -1);
    return w == 0 ? j : 2 + j; // T = 2
}

#ifndef SKEW_WPE
#define SKEW_WPE 5 // waves per SIMD the register budget is set for (5: 96 VGPRs, what the LDS per workgroup allows)
#endif
template <int T, int TOP>
__global__ __launch_bounds__(64 * T) __attribute__((amdgpu_waves_per_eu(SKEW_WPE, SKEW_WPE))) void k_refine_skew(StageArgs a) {
    typedef SkewLds<T> LY;
    constexpr int NE = LY::NE, RPB = LY::RPB;
    constexpr int NSTR = T == 4 ? 1 : 2; // streams a wave carries
#ifndef SKEW_PAD
#define SKEW_PAD 0 // (measurement builds: extra LDS per workgroup, i.e. fewer workgroups per CU)
#endif
    __shared__ __attribute__((aligned(16))) char s_mem[LY::SIZE + SKEW_PAD];
    const DirArgs &d = a.d[blockIdx.z];
    const int W = a.W;
    const int XL = d.own.XL, XR = d.own.XR, YL = d.own.YL, YR = d.own.YR;
    const int UW = a.skew_uw;
    const int x0 = ((XL + 1 - (T - 1)) & ~7) + (int)blockIdx.x * UW;                     // lane 0's column
    const int xa = max(x0 + T - 1, XL + 1), xb = min(x0 + T - 1 + UW, XR);               // owned columns [xa, xb)
    const int ya = YL + 1 + (int)blockIdx.y * a.skew_rows, yb = min(ya + a.skew_rows, YR); // owned rows [ya, yb)
    if (xa >= xb || ya >= YR) return; // workgroup-uniform
    const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), t = wid + 1; // this wave's sweep within the launch
    { // the specified exp's table and the emit masks (the loop's first barriers come before any use)
        unsigned long long *w = (unsigned long long *)(s_mem + LY::OFF_EXP);
        for (int i = threadIdx.x; i < 256; i += 64 * T) w[i] = RSM_EXP_TAB[i];
        if (threadIdx.x < 2 * NE) ((u64 *)(s_mem + LY::OFF_EMIT))[threadIdx.x] = 0ull;
    }
    const int x = x0 + lane;
    const int y0 = max(YL, ya - T), y1 = min(YR, yb - 1 + T); // staged rows [y0, y1]
    const int s0 = y0 - 1;                                    // the first step
    // what sweep t can compute here
    const int cy_lo = max(YL + 1, ya - (T - t)), cy_hi = min(YR - 1, yb - 1 + (T - t));
    const bool colok = lane >= t - 1 && lane <= 64 - t && x >= XL + 1 && x <= XR - 1;
    const u64 m_col = __builtin_amdgcn_ballot_w64(colok), m_own = __builtin_amdgcn_ballot_w64(x >= xa && x < xb);
    const unsigned shard = (blockIdx.x + blockIdx.y * gridDim.x + blockIdx.z * 7u) & (RF_UPD_SHARDS - 1);
    const double ws = a.ws;
    const bool wsok = ws >= 0x1p-200 && ws <= 0x1p200; // the unscaled divisions' guard on pwp + ws (pwp is in [0, 1])

    // ---- this wave's streams: a row's bytes come as 16-byte pieces, lane l of the stream's first nl lanes takes piece l.
    // A stream issues exactly ONE load per step, whatever the row (past the chunk's last row it re-reads that row, lanes beyond
    // nl re-read piece nl - 1): the number of loads in flight is then the same on every path, and the wait in front of the
    // LDS write is exact -- vmcnt(1): the row issued a step ago -- instead of the vmcnt(0) a conditional load forces.
    const char *gptr[NSTR]; // the next row to issue, lane 0's piece
    unsigned gpitch[NSTR];  // bytes between rows
    unsigned voff[NSTR];    // this lane's piece
    u64 st_mask[NSTR];      // the lanes that carry a piece
    unsigned st_base[NSTR], st_pitch[NSTR], st_wrap[NSTR], st_off[NSTR]; // the destination slots in LDS
    int iss_row[NSTR];      // the row the next issue is for
    bool st_key[NSTR];
#pragma unroll
    for (int j = 0; j < NSTR; j++) {
        const int sid = skew_stream<T>(wid, j);
        const size_t p0 = (size_t)y0 * W + x0;
        int nl;
        if (sid == 0) { // state: columns x0 - 2 .. x0 + 65
            gptr[j] = (const char *)(d.f64_a + p0 - 2);
            gpitch[j] = (unsigned)W * 8u;
            nl = 34;
            st_base[j] = LY::OFF_D, st_pitch[j] = RPB, st_wrap[j] = 4 * RPB;
            iss_row[j] = y0 + 1; // (row y0 is issued before the loop) -- a state row is issued two steps before its first use
        } else if (sid == 2 || sid == 3) {
            gptr[j] = (const char *)(d.rf_ent + p0 + (size_t)(sid - 2) * a.rf_stride);
            gpitch[j] = (unsigned)W * 16u;
            nl = 64;
            st_base[j] = LY::OFF_ENT + (sid - 2) * 1024, st_pitch[j] = 2048, st_wrap[j] = NE * 2048;
            iss_row[j] = y0;
        } else { // keys -- and, for a slot without a stream, the same loads with nothing written
            gptr[j] = (const char *)(d.rf_key + p0);
            gpitch[j] = (unsigned)W * 4u;
            nl = sid == 1 ? 16 : 0;
            st_base[j] = LY::OFF_KEY, st_pitch[j] = 256, st_wrap[j] = NE * 256;
            iss_row[j] = y0;
        }
        st_mask[j] = __builtin_amdgcn_ballot_w64(lane < nl);
        voff[j] = (unsigned)min(lane, max(nl - 1, 0)) * 16u;
        // the first write (end of the first step) is row y0 for the state stream -- slot 0 -- and the row BEFORE y0 for the others
        // (nothing loaded yet: garbage nobody reads), which therefore starts in the last slot so that row y0 lands in slot 0
        st_off[j] = sid == 0 ? 0u : st_wrap[j] - st_pitch[j];
        st_key[j] = sid == 1;
    }
    const unsigned lane16 = (unsigned)lane * 16u;
    const unsigned span_y = (unsigned)(y1 - y0);
    const bool last = t == T;
    // scalar row pointer + 32-bit lane byte offset = the load's own addressing mode: no vector address arithmetic
    // (the empty asm keeps the compiler from widening the offset into a 64-bit lane address that lives across the loop)
#define SKEW_LD16(dst, base, off)                                    \
    do {                                                             \
        asm volatile("" : "+v"(off));                                \
        dst = *(const double2 *)((base) + (off));                    \
    } while (0)
    double2 stg[NSTR][2]; // two rows in flight per stream: issued in step s, written to LDS at the end of step s + 1
#pragma unroll
    for (int j = 0; j < NSTR; j++) {
        stg[j][0] = make_double2(0.0, 0.0);
        SKEW_LD16(stg[j][1], gptr[j], voff[j]); // (the first state row; the other streams' first write is a step later)
        if (skew_stream<T>(wid, j) == 0) gptr[j] += gpitch[j];
    }
    // ---- this wave's rows
    const unsigned v_rd = LY::OFF_D + (unsigned)(t - 1) * LY::RING + (unsigned)lane * 8u; // ring t - 1, column x0 - 2 + lane
    unsigned ek = (unsigned)(((NE - 2 * t) % NE + NE) % NE) * 256u; // slot of this wave's row in the cache rows, x 256 B
    int r = s0 - 2 * t + 1;
    // the rows sweep t can compute, relative to y0, as ONE unsigned compare (none: a bound no row index reaches); likewise the
    // staged rows below level T, which copy through where they cannot be computed
    const unsigned c_lo = cy_hi >= cy_lo ? (unsigned)(cy_lo - y0) : 0x40000000u, c_span = cy_hi >= cy_lo ? (unsigned)(cy_hi - cy_lo) : 0u;
    const unsigned copy_rows = last ? 0u : span_y + 1u;
    const u64 m_store = last ? m_own : 0ull;    // level T: the lanes whose results go to memory
    const u64 m_wsbad = wsok ? 0ull : ~0ull;    // (a ws outside the unscaled division's guard: every row takes the general path)
    const char *optr = (const char *)(d.f64_b + (ptrdiff_t)r * W + x0); // level T's output row (lane 0)
    const unsigned opitch = (unsigned)W * 8u;
    const unsigned lane8 = (unsigned)lane * 8u;
    const unsigned lane4 = (unsigned)lane * 4u;
    const int nsteps = y1 - y0 + 2 * T + 1; // steps s0 .. y1 + 2T - 1
    { // (a store that stores nothing, so that the first step meets the same operations in flight as every other: load, store, load)
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(optr), 0, 512, 0x00020000);
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(skew_v2i, 0.0), rs, 0xffffffffu, 0, 0);
    }
#ifdef SKEW_TIMING // per-phase shader-clock split of a step (s_memtime), printed by a few waves: a measurement build, never shipped
    unsigned long long tm[4] = {0, 0, 0, 0}, tq0 = 0, tq1;
#define SKEW_TICK(i)                                   \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); \
    tq1 = __builtin_amdgcn_s_memtime();                \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); \
    tm[i] += tq1 - tq0;                                \
    tq0 = tq1;
#else
#define SKEW_TICK(i)
#endif
    __syncthreads();
    // Issue priority in rotation (a.skew_prio = log2 of the period in shader clocks; 0 = off).  The hardware issues from the
    // OLDEST ready wave of a SIMD first; the five workgroups of a CU differ in age, and the more efficiently a wave issues the
    // more completely the older ones starve the younger (measured per step: 1 500 ticks for the oldest workgroup of a CU, 4 000
    // for the youngest): the launch then ends in a long tail of young workgroups running one or two waves per SIMD, which cannot
    // fill the vector unit.  s_setprio outranks age (a static priority by reverse age inverts the order), so a priority that
    // rotates with the CLOCK -- every workgroup of the chip changes rank at the same moments, whatever its own progress -- from a
    // phase given by the workgroup's place in the dispatch order gives every age the same share of the time at every rank.
    const int prio_shift = a.skew_prio;
    const unsigned prio_phase = (blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)) >> 8; // ~ the workgroup's age rank on its CU
    unsigned prio_epoch = 0xffffffffu;
#pragma unroll 1
    for (int q4 = 0; q4 < nsteps; q4 += 4) {
        if (prio_shift > 0) { // wave-uniform, once per four steps
            const unsigned ep = (unsigned)(__builtin_amdgcn_s_memtime() >> prio_shift);
            if (ep != prio_epoch) {
                prio_epoch = ep;
                const unsigned p = (prio_phase + ep) % 5u;
                if (p == 0) __builtin_amdgcn_s_setprio(0);
                else if (p == 1) __builtin_amdgcn_s_setprio(1);
                else if (p == 2) __builtin_amdgcn_s_setprio(2);
                else __builtin_amdgcn_s_setprio(3);
            }
        }
#pragma unroll
        for (int K = 0; K < 4; K++) {
#ifdef SKEW_TIMING
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            tq0 = __builtin_amdgcn_s_memtime();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
            // (1) this step's loads
#pragma unroll
            for (int j = 0; j < NSTR; j++) {
                SKEW_LD16(stg[j][K & 1], gptr[j], voff[j]);
                gptr[j] += iss_row[j] < y1 ? gpitch[j] : 0u;
            }
            // (2) this wave's row
            const unsigned rs = (unsigned)(r - y0);
            double val = 0.0;
            u64 m_st = 0ull; // level T: the lanes whose result goes out from here
            if (rs - c_lo <= c_span) { // wave-uniform: a row sweep t can compute
                const unsigned rc = v_rd + ((K + 2) & 3) * RPB, rn = v_rd + ((K + 1) & 3) * RPB, rsl = v_rd + ((K + 3) & 3) * RPB;
                double dW = *(const double *)(s_mem + rc + 8), dC = *(const double *)(s_mem + rc + 16), dE = *(const double *)(s_mem + rc + 24);
                double dN = *(const double *)(s_mem + rn + 16), dS = *(const double *)(s_mem + rsl + 16);
                uint32_t kk = *(const uint32_t *)(s_mem + LY::OFF_KEY + ek + lane4);
                const double NM = (double)NOMATCH;
                // (one LDS round trip: none of the reads may sink below the live-row test, which needs dC only)
                asm volatile("" : "+v"(dN), "+v"(dS), "+v"(dE), "+v"(dW), "+v"(kk));
                val = dC;
                u64 m_redo = 0ull;
                const u64 m_lv = m_col & RF_FNE(dC, NM); // .cpp:613
                if (m_lv) { // wave-uniform: a row without a live pixel (outside an elliptic mask, a hole) copies through
                    const u64 m_ew = RF_FNE(dE, NM) & RF_FNE(dW, NM), m_ns = RF_FNE(dS, NM) & RF_FNE(dN, NM); // .cpp:620
                    if (!((m_lv & (m_ew ^ m_ns)) | m_wsbad)) { // no live pixel with exactly one valid neighbour pair: the common row
                        const int rel = (int)(dC - 1.5); // .cpp:625 (iMatch - x)
                        const int way = rel & 1;
                        // the way the state selects, as ONE 16-byte read (two 8-byte reads at a 16-byte lane stride conflict two ways)
                        const double2 pd = *(const double2 *)__builtin_assume_aligned(s_mem + LY::OFF_ENT + ek * 8u + (unsigned)way * 1024u + lane16, 16);
                        const int crel = (int)(int16_t)(kk >> (way << 4));
                        // the common row's update; the masks known EARLY in its chain select what is written, the two that need the
                        // numerators only decide whether the row is redone, so the row's result never waits for them
                        double u;
                        u64 m_early;
                        skew_common_row(dC, dE, dW, dN, dS, pd, crel, rel, ws, m_lv & m_ew & m_ns, (ExpTab)(s_mem + LY::OFF_EXP), u, m_early, m_redo);
                        val = rf_sel(m_early) ? u : dC; // (a live pixel without a valid pair keeps dC, .cpp:655)
                    } else
                        m_redo = m_lv & (m_ew | m_ns);
                }
                if (!last) *(double *)(s_mem + v_rd + LY::RING + K * RPB + 16) = val;
                m_st = m_lv & m_store & ~m_redo; // level T's computable rows are the owned rows
                if (__builtin_expect(m_redo != 0ull, 0)) { // wave-uniform, rare
                    SkewRare q;
                    q.A = d.img4_own, q.B = d.img4_oth;
                    q.upd = a.upd_list + (size_t)shard * a.upd_cap;
                    q.cnt = a.upd_cnt + (a.flag3 & 1) * RF_UPD_SHARDS + shard;
                    q.gout = last ? (double *)optr : nullptr;
                    q.lds = s_mem;
                    q.ws = ws;
                    q.W = W, q.H = a.H, q.upd_cap = a.upd_cap;
                    q.x0 = x0, q.r = r;
                    q.rdC = LY::OFF_D + (unsigned)(t - 1) * LY::RING + ((K + 2) & 3) * RPB + 16;
                    q.rdN = LY::OFF_D + (unsigned)(t - 1) * LY::RING + ((K + 1) & 3) * RPB + 16;
                    q.rdS = LY::OFF_D + (unsigned)(t - 1) * LY::RING + ((K + 3) & 3) * RPB + 16;
                    q.wr = LY::OFF_D + (unsigned)t * LY::RING + K * RPB + 16;
                    q.ent = LY::OFF_ENT + ek * 8u, q.key = LY::OFF_KEY + ek, q.emit = LY::OFF_EMIT + (ek >> 4), q.ml = LY::OFF_ML + wid * 64, q.tab = LY::OFF_EXP;
                    q.dir = (uint32_t)blockIdx.z << 31;
                    q.owned_row = r >= ya && r < yb;
                    skew_rare_row(q, m_redo, m_own);
                    // every memory operation of the rare path (its image loads, list appends, register spills) is complete before
                    // the common path resumes: otherwise the compiler's wait counts after the join must assume them in flight and
                    // the common row waits for its own staging loads in the middle of its arithmetic
                    __builtin_amdgcn_s_waitcnt(0x0F70); // vmcnt(0)
                }
            } else if (rs < copy_rows) { // a staged row beyond what this sweep can compute here: copies through
                *(double *)(s_mem + v_rd + LY::RING + K * RPB + 16) = *(const double *)(s_mem + v_rd + ((K + 2) & 3) * RPB + 16);
            }
            SKEW_TICK(0) // the row's update
#ifdef SKEW_TIMING
            asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
#endif
            SKEW_TICK(1) // waiting for the row issued a step ago
            // (3) the rows issued in the previous step go to LDS -- every step, whatever the row: a row before the chunk's first or
            // beyond its last lands in the slot its number gives it, which nothing valid reads before the slot's next row arrives
            // (the schedule that protects a slot's previous occupant holds for any row number).  So every path through a step
            // waits for exactly the load of the step before, and the compiler's wait counts stay exact.
#pragma unroll
            for (int j = 0; j < NSTR; j++) {
                const unsigned dst = st_base[j] + st_off[j];
                if (rf_sel(st_mask[j])) *(double2 *)(s_mem + dst + lane16) = stg[j][(K + 1) & 1];
                asm volatile("" ::"v"(stg[j][(K + 1) & 1].x), "v"(stg[j][(K + 1) & 1].y)); // (the wait also on the path of a wave without lanes in the mask)
                if (st_key[j] && lane < 2) *(u64 *)(s_mem + LY::OFF_EMIT + (st_off[j] >> 4) + lane8) = 0ull; // the slot's emit masks with its keys
                st_off[j] += st_pitch[j];
                if (st_off[j] == st_wrap[j]) st_off[j] = 0;
                iss_row[j]++;
            }
            // (4) level T's row goes out -- as a BUFFER store that every wave issues in every step: the descriptor covers the 64
            // columns of this wave's row, a lane that has nothing to store (every lane of the levels below T) gets an offset beyond it
            // and the hardware drops it.  A store inside a branch would leave the compiler two paths with different numbers of
            // memory operations in flight, and its wait for the staged row (above) would then also wait for level T's previous
            // STORE -- whose round trip has one step to complete, not two: exactly the wave every other wave's barrier waits for.
            {
                const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(optr), 0, 512, 0x00020000);
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(skew_v2i, val), rs, rf_sel(m_st) ? lane8 : 0xffffffffu, 0, 0);
            }
            r++;
            optr += opitch;
            ek += 256u;
            if (ek == NE * 256u) ek = 0;
            SKEW_TICK(2) // LDS write of the staged row, the store
            __syncthreads();
            SKEW_TICK(3) // barrier
        }
    }
#undef SKEW_LD16
#ifdef SKEW_TIMING
    if (TOP && lane == 0 && a.flag3 == 28 && blockIdx.x == 13 && (blockIdx.y % 4) == 1)
        printf("skewtime launch %d z %d wg %d %d wave %d steps %d: update %llu loadwait %llu write %llu barrier %llu\n", a.flag3, (int)blockIdx.z, (int)blockIdx.x, (int)blockIdx.y, wid, nsteps,
               tm[0] / nsteps, tm[1] / nsteps, tm[2] / nsteps, tm[3] / nsteps);
#endif
}

// strips of 64 lanes the interior of margin m takes (strip b owns the columns [xorg + T - 1 + b uw, + uw) of [XL + 1, XR - 1])
int refine_skew_strips(const Mg &m, int T, int uw) {
    const int xorg = (m.XL + 1 - (T - 1)) & ~7;
    return std::max(1, (m.XR - xorg - T + 1 + uw - 1) / uw);
}

// The kernel addresses rows of 64 + 4 columns on the flat row-major buffers without clamping: a strip may hang over a row's
// ends into the neighbouring rows, never out of the level (rows 1 .. H - 2 are the only ones a margin can hold, r >= 1).
bool refine_skew_fits(const StageArgs &a) {
    if (a.W < 80 || a.H < 3) return false;
    for (int v = 0; v < a.ndir; v++) {
        const Mg &m = a.d[v].own;
        if (m.XL < 1 || m.YL < 1 || m.XR > a.W - 2 || m.YR > a.H - 2) return false;
    }
    return true;
}

// T sweeps f64_a -> f64_b in one launch (a.flag3 = launch index, a.skew_rows = rows per chunk, a.skew_uw = columns per strip)
// + the launch that applies its cache updates.  T in {2, 3, 4}.
void launch_refine_skew(const StageArgs &a, int T, hipStream_t st, hipEvent_t ev0, hipEvent_t ev1) {
    int rows = 0, strips = 0;
    for (int v = 0; v < a.ndir; v++) {
        rows = std::max(rows, a.d[v].own.YR - a.d[v].own.YL - 1);
        strips = std::max(strips, refine_skew_strips(a.d[v].own, T, a.skew_uw));
    }
    if (rows <= 0 || a.skew_rows <= 0 || a.skew_uw <= 0 || a.skew_uw > 66 - 2 * T) return;
    const dim3 grid(strips, (rows + a.skew_rows - 1) / a.skew_rows, a.ndir);
    if (ev0) (void)hipEventRecord(ev0, st);
#define RF_LAUNCH(TT)                                                                               \
    do {                                                                                            \
        if (a.flag) hipLaunchKernelGGL((k_refine_skew<TT, 1>), grid, dim3(64 * TT), 0, st, a);    \
        else hipLaunchKernelGGL((k_refine_skew<TT, 0>), grid, dim3(64 * TT), 0, st, a);           \
    } while (0)
    if (T == 2) RF_LAUNCH(2);
    else if (T == 3) RF_LAUNCH(3);
    else RF_LAUNCH(4);
#undef RF_LAUNCH
    if (ev1) (void)hipEventRecord(ev1, st);
    launch_refine_apply(a, st);
}
