// k_constraints.hip -- the integer constraint / filter stages of MatchOneLayer.
//   smooth        CStereoMatching::SmoothConstraint      .cpp:370-448  (gather form of the scatter)
//   order         CStereoMatching::OrderConstraint       .cpp:310-368  (one workgroup per row)
//   uniq<T>       CStereoMatching::UniquenessContraint_  .cpp:463-497  (one wave per row, carry chain)
//   set_boundary  CStereoMatching::SetBoundary_smooth    .cpp:817-942  (column sweeps, then row sweeps)
//   median        CStereoMatching::MedianFilter          .cpp:763-815
#include "rsm_dev.h"

__device__ __forceinline__ bool in_margin(const Mg &m, int x, int y) {
    return x >= m.XL && x <= m.XR && y >= m.YL && y <= m.YR;
}

// ---------------------------------------------------------------- SmoothConstraint
// The reference scatters neighbour counts into a CV_8UC2 map serially (.cpp:380-432) and then kills
// pixels with total==0 || 2*differ>total (.cpp:443).  Gather form per target pixel (X,Y); "src in M"
// means the scattering pixel lies inside the own margin.  The SE total counts land at BYTE index x
// of row y and x+2 of row y+1 (.cpp:423-424), i.e. at pixel x/2 (total if x even, differ if odd):
// the four "slip" terms below.
__global__ void k_smooth(StageArgs a) {
    const DirArgs &d = a.d[blockIdx.z];
    const int W = a.W, H = a.H;
    const Mg M = d.own;
    const int X = M.XL + blockIdx.x * blockDim.x + threadIdx.x; // the margin only: outside it the map is copied as it is
    const int Y = M.YL + blockIdx.y;
    if (X > M.XR || Y > M.YR) return;
    const int16_t *D = d.d16_in;
    const size_t pix = (size_t)Y * W + X;
    const int16_t dc = D[pix];
    auto val = [&](int x, int y) -> int {
        return (x >= 0 && x < W && y >= 0 && y < H) ? (int)D[(size_t)y * W + x] : NOMATCH;
    };
    auto differ = [](int p, int q) -> int { return abs(p - q) > 1; };
    int T = 0, F = 0;
    const bool vc = dc != NOMATCH;
    // the neighbourhood is loaded whether or not (X,Y) is valid: together with the centre, not after it
    const int e = val(X + 1, Y), sw = val(X - 1, Y + 1), s = val(X, Y + 1), se = val(X + 1, Y + 1);
    const int w = val(X - 1, Y), ne = val(X + 1, Y - 1), nn = val(X, Y - 1), nw = val(X - 1, Y - 1);
    if (vc) {
        // (X,Y) as source (it is inside M): E, SW, S totals; E, SW, S, SE differs
        if (e != NOMATCH) { T++; F += differ(dc, e); }
        if (sw != NOMATCH) { T++; F += differ(dc, sw); }
        if (s != NOMATCH) { T++; F += differ(dc, s); }
        if (se != NOMATCH) { F += differ(dc, se); }
        // (X,Y) as target of W (its E), NE (its SW), N (its S), NW (its SE: differ only)
        if (w != NOMATCH && in_margin(M, X - 1, Y)) { T++; F += differ(w, dc); }
        if (ne != NOMATCH && in_margin(M, X + 1, Y - 1)) { T++; F += differ(ne, dc); }
        if (nn != NOMATCH && in_margin(M, X, Y - 1)) { T++; F += differ(nn, dc); }
        if (nw != NOMATCH && in_margin(M, X - 1, Y - 1)) { F += differ(nw, dc); }
    }
    // slip terms (independent of the validity of (X,Y)); the eight loads are issued together -- behind short-circuit
    // conditions they would be dependent memory round trips
    const int s0 = val(2 * X, Y), s1 = val(2 * X + 1, Y + 1), s2 = val(2 * X - 2, Y - 1), s3 = val(2 * X - 1, Y);
    const int s4 = val(2 * X + 1, Y), s5 = val(2 * X + 2, Y + 1), s6 = val(2 * X - 1, Y - 1);
    T += (int)(in_margin(M, 2 * X, Y) && s0 != NOMATCH && s1 != NOMATCH);
    T += (int)(in_margin(M, 2 * X - 2, Y - 1) && s2 != NOMATCH && s3 != NOMATCH);
    F += (int)(in_margin(M, 2 * X + 1, Y) && s4 != NOMATCH && s5 != NOMATCH);
    F += (int)(in_margin(M, 2 * X - 1, Y - 1) && s6 != NOMATCH && s0 != NOMATCH);
    T &= 255; // the reference counters are uchar
    F &= 255;
    d.d16_out[pix] = (T == 0 || (F << 1) > T) ? (int16_t)NOMATCH : dc;
}

void launch_smooth(const StageArgs &a, hipStream_t st, bool copy_outside) {
    int rows = 0, cols = 0;
    for (int v = 0; v < a.ndir; v++) {
        if (copy_outside) // the caller may know that both maps already agree outside the margin (NOMATCH there)
            (void)hipMemcpyAsync(a.d[v].d16_out, a.d[v].d16_in, (size_t)a.W * a.H * sizeof(int16_t), hipMemcpyDeviceToDevice, st);
        rows = max(rows, a.d[v].own.YR - a.d[v].own.YL + 1);
        cols = max(cols, a.d[v].own.XR - a.d[v].own.XL + 1);
    }
    if (rows <= 0 || cols <= 0) return;
    hipLaunchKernelGGL(k_smooth, dim3((cols + 255) / 256, rows, a.ndir), dim3(256), 0, st, a);
}

// ---------------------------------------------------------------- OrderConstraint
// One wave per row.  Valid pixels are compacted in ascending x into LDS (m = p[x] + x).  Crossing count of i =
// #{j<i : m_j > m_i} + #{j>i : m_j < m_i} -- the row sums of the reference's symmetric matrix A (.cpp:337-353)
// without ever materialising it -- then the greedy loop of .cpp:354-364: remove the first pixel with the largest
// count until no crossings remain.
// Crossings are local: position i is a cut (no pair j <= i < k crosses) iff max(m_0..m_i) <= min(m_i+1..), and
// removals on one side of a cut never change a count on the other side, so the greedy order restricted to a
// cut-free run ("component") is the order the reference's global arg-max visits that run in.  The row therefore
// splits into independent components (typically 2-5 pixels around a disparity step): counts only look inside
// the component, and every lane runs the greedy loop of its own components; only components longer than
// ORD_BIG are worked on by the whole wave.
#define ORD_BIG 48
#define ORD_U 8
#define ORD_NW 4 // waves per row: wave 0 does the serial passes, all of them the counting and the greedy loops
__global__ __launch_bounds__(64 * ORD_NW) void k_order(StageArgs a, int maxL) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    __shared__ int s_n, s_win, s_go, s_ncomp, s_nbig, s_removed, s_bv[ORD_NW], s_bi[ORD_NW];
    const DirArgs &d = a.d[blockIdx.z];
    const int y = d.own.YL + blockIdx.x;
    if (y > d.own.YR) return;
    const int W = a.W, XL = d.own.XL, XR = d.own.XR;
    int16_t *line = (int16_t *)smem;                    // [maxL] m of the valid pixels, ascending x
    int16_t *cnt = line + maxL;                         // [maxL] prefix max, later crossing counts (-1 = removed)
    uint32_t *comps = (uint32_t *)(cnt + maxL);         // [maxL / 2] components with crossings: first | last << 16
    unsigned long long *ends = (unsigned long long *)(comps + maxL / 2); // [maxL / 64 + 1] bit i: a component ends at i
    int16_t *bmx = (int16_t *)(ends + maxL / 64 + 1), *bmn = bmx + maxL / 64 + 1; // max / min of every 64-element block
    uint32_t *bigq = (uint32_t *)(bmn + maxL / 64 + 1); // [maxL / ORD_BIG + 1] the long components
    int16_t *p = d.d16_in + (size_t)y * W;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const unsigned long long lt = (1ull << lane) - 1ull; // lanes below this one
    if (tid == 0) {
        s_go = 0;
        s_ncomp = 0;
        s_nbig = 0;
        s_removed = 0;
    }
    __syncthreads();
    if (wv == 0) {
        int n = 0, pmin = 0x7fffffff, pmax = -0x7fffffff;
        for (int x0 = XL; x0 <= XR; x0 += 64 * ORD_U) { // ORD_U row segments in flight: the pass is load latency
            int v[ORD_U];
#pragma unroll
            for (int u = 0; u < ORD_U; u++) {
                const int x = x0 + u * 64 + lane;
                v[u] = (x <= XR) ? (int)p[x] : NOMATCH;
            }
#pragma unroll
            for (int u = 0; u < ORD_U; u++) {
                const bool valid = v[u] != NOMATCH;
                const unsigned long long m = __ballot(valid);
                if (valid) {
                    line[n + __popcll(m & lt)] = (int16_t)(v[u] + x0 + u * 64 + lane);
                    pmin = min(pmin, v[u]);
                    pmax = max(pmax, v[u]);
                }
                n += __popcll(m);
            }
        }
        // prefix max (kept in cnt for now); a row that is already non-decreasing has no crossings at all
        bool inv = false;
        int carry = -32768;
        for (int c0 = 0; c0 < n; c0 += 64) {
            const int i = c0 + lane;
            const int own = (i < n) ? (int)line[i] : -32768;
            int v = own;
#pragma unroll
            for (int s = 1; s < 64; s <<= 1) {
                const int o = __shfl_up(v, s);
                if (lane >= s) v = max(v, o);
            }
            if (lane == 63) bmx[c0 >> 6] = (int16_t)v; // the block's own max (before the carry) ...
            v = max(v, carry);
            inv = inv || (i < n && own < v);
            if (i < n) cnt[i] = (int16_t)v;
            carry = __shfl(v, 63);
            int mn = (i < n) ? own : 32767; // ... and min
            for (int o = 32; o > 0; o >>= 1) mn = min(mn, __shfl_xor(mn, o));
            if (lane == 0) bmn[c0 >> 6] = (int16_t)mn;
        }
        if (n >= 2 && __any(inv)) {
            for (int o = 32; o > 0; o >>= 1) {
                pmin = min(pmin, __shfl_xor(pmin, o));
                pmax = max(pmax, __shfl_xor(pmax, o));
            }
            // suffix min -> where components end
            carry = 32767;
            for (int c = ((n + 63) >> 6) - 1; c >= 0; c--) {
                const int i = c * 64 + lane;
                int v = (i < n) ? (int)line[i] : 32767;
#pragma unroll
                for (int s = 1; s < 64; s <<= 1) {
                    const int o = __shfl_down(v, s);
                    if (lane + s < 64) v = min(v, o);
                }
                v = min(v, carry);           // min(m_i ..)
                int nxt = __shfl_down(v, 1); // min(m_i+1 ..)
                if (lane == 63) nxt = carry;
                const bool end = i < n && (i == n - 1 || (int)cnt[i] <= nxt);
                const unsigned long long e = __ballot(end);
                if (lane == 0) ends[c] = e;
                carry = __shfl(v, 0);
            }
            if (lane == 0) {
                s_n = n;
                // j < i crosses i only if x_i - x_j < p_j - p_i <= pmax - pmin, and x_i - x_j >= i - j: besides the
                // component, a window of `win` list positions on each side bounds every search
                s_win = pmax - pmin;
                s_go = 1;
            }
        }
    }
    __syncthreads();
    if (!s_go) return;
    const int n = s_n, win = s_win, nchunk = (n + 63) >> 6;
    // counts inside the components; list of the components that have any crossing
    for (int c = wv; c < nchunk; c += ORD_NW) {
        const int c0 = c << 6, i = c0 + lane;
        int cs = 0, ce = 0;
        bool head = false;
        const unsigned long long E = ends[c];
        int prev_end = -1, next_end = n - 1; // last end before this chunk / first end after it (uniform look-ups)
        for (int q = c - 1; q >= 0; q--) {
            const unsigned long long m = ends[q];
            if (m) {
                prev_end = q * 64 + 63 - __builtin_clzll(m);
                break;
            }
        }
        for (int q = c + 1; q < nchunk; q++) {
            const unsigned long long m = ends[q];
            if (m) {
                next_end = q * 64 + __builtin_ctzll(m);
                break;
            }
        }
        if (i < n) {
            unsigned long long m = E & ~lt; // ends at or after i
            ce = m ? c0 + __builtin_ctzll(m) : next_end;
            m = E & lt;                     // ends before i
            cs = m ? c0 + 64 - __builtin_clzll(m) : prev_end + 1;
            head = (i == cs) && (ce > cs);
        }
        // crossing counts of the chunk's 64 pixels, the wave walking the 64-element blocks their components (and
        // the +-win window) reach: block loads are wave-uniform (LDS broadcasts in a pipelined loop), and a block
        // is skipped when its min / max say nothing in it can cross, counted whole when everything does.
        const int mi = (i < n) ? (int)line[i] : 0;
        int j0 = 0x7fffffff, j1 = -1, k = 0; // this lane's search range [j0, j1] (empty for a lone pixel)
        if (i < n && ce > cs) {
            j0 = max(cs, i - win);
            j1 = min(ce, i + win);
        }
        int lo = j0, hi = j1;
        for (int o = 32; o > 0; o >>= 1) {
            lo = min(lo, __shfl_xor(lo, o));
            hi = max(hi, __shfl_xor(hi, o));
        }
        for (int b = lo >> 6; hi >= 0 && b <= (hi >> 6); b++) { // uniform
            const int t0 = b << 6, t1 = min(t0 + 63, n - 1);
            const int mx = bmx[b], mn = bmn[b];
            bool scan = false;
            if (t1 < j0 || t0 > j1) {
                // out of this lane's range
            } else if (t0 >= j0 && t1 < i) { // wholly to the left: crosses where m_j > m_i
                if (mn > mi) k += t1 - t0 + 1;
                else scan = mx > mi;
            } else if (t0 > i && t1 <= j1) { // wholly to the right: crosses where m_j < m_i
                if (mx < mi) k += t1 - t0 + 1;
                else scan = mn < mi;
            } else {
                scan = true;
            }
            if (__any(scan)) {
                for (int t = t0; t <= t1; t++) {
                    const int v = line[t];
                    if (scan) k += (t >= j0 && t < i && v > mi) || (t > i && t <= j1 && v < mi);
                }
            }
        }
        const unsigned long long hm = __ballot(head);
        int base = 0;
        if (hm && lane == 0) base = atomicAdd(&s_ncomp, __popcll(hm));
        base = __shfl(base, 0);
        if (head) comps[base + __popcll(hm & lt)] = (uint32_t)cs | ((uint32_t)ce << 16);
        // (the prefix max in cnt is no longer needed: the ends are final)
        if (i < n) cnt[i] = (int16_t)k;
    }
    __syncthreads();
    // greedy removal: a lane per small component, the long ones are queued
    const int ncomp = s_ncomp;
    for (int t = tid; t < ncomp; t += 64 * ORD_NW) {
        const uint32_t cc = comps[t];
        const int cs = cc & 0xffff, ce = cc >> 16;
        if (ce - cs + 1 > ORD_BIG) {
            bigq[atomicAdd(&s_nbig, 1)] = cc;
            continue;
        }
        for (;;) {
            int bv = -1, bi = cs;
            for (int j = cs; j <= ce; j++) { // first index of the maximum (Armadillo's max(idx))
                const int c = cnt[j];
                if (c > bv) {
                    bv = c;
                    bi = j;
                }
            }
            if (bv <= 0) break; // no crossing left here (.cpp:354)
            const int mb = line[bi];
            for (int j = max(cs, bi - win); j <= min(ce, bi + win); j++) {
                const int c = cnt[j];
                if (c < 0 || j == bi) continue; // removed pixels have no edges left
                const int mj = line[j];
                if ((j < bi && mj > mb) || (j > bi && mj < mb)) cnt[j] = (int16_t)(c - 1);
            }
            cnt[bi] = -1; // row/col of A zeroed (.cpp:359-361)
            s_removed = 1;
        }
    }
    __syncthreads();
    const int nbig = s_nbig;
    for (int t = 0; t < nbig; t++) { // the whole workgroup on one long component at a time
        const uint32_t cc = bigq[t];
        const int cs = cc & 0xffff, ce = cc >> 16;
        for (;;) {
            int bv = -1, bi = 0x7fffffff;
            for (int i = cs + tid; i <= ce; i += 64 * ORD_NW) {
                const int c = cnt[i];
                if (c > bv) {
                    bv = c;
                    bi = i;
                }
            }
            for (int o = 32; o > 0; o >>= 1) {
                const int ov = __shfl_xor(bv, o), oi = __shfl_xor(bi, o);
                if (ov > bv || (ov == bv && oi < bi)) {
                    bv = ov;
                    bi = oi;
                }
            }
            if (lane == 0) {
                s_bv[wv] = bv;
                s_bi[wv] = bi;
            }
            __syncthreads();
            bv = s_bv[0];
            bi = s_bi[0];
#pragma unroll
            for (int w = 1; w < ORD_NW; w++)
                if (s_bv[w] > bv || (s_bv[w] == bv && s_bi[w] < bi)) {
                    bv = s_bv[w];
                    bi = s_bi[w];
                }
            if (bv <= 0) break; // uniform
            const int mb = line[bi];
            const int j0 = max(cs, bi - win), j1 = min(ce, bi + win);
            for (int j = j0 + tid; j <= j1; j += 64 * ORD_NW) {
                const int c = cnt[j];
                if (c < 0 || j == bi) continue;
                const int mj = line[j];
                if ((j < bi && mj > mb) || (j > bi && mj < mb)) cnt[j] = (int16_t)(c - 1);
            }
            if (tid == 0) {
                cnt[bi] = -1;
                s_removed = 1;
            }
            __syncthreads();
        }
        __syncthreads();
    }
    __syncthreads();
    if (!s_removed || wv != 0) return;
    // the removed pixels become NOMATCH (.cpp:363): same compaction order as above
    int k = 0;
    for (int x0 = XL; x0 <= XR; x0 += 64 * ORD_U) {
        int v[ORD_U];
#pragma unroll
        for (int u = 0; u < ORD_U; u++) {
            const int x = x0 + u * 64 + lane;
            v[u] = (x <= XR) ? (int)p[x] : NOMATCH;
        }
#pragma unroll
        for (int u = 0; u < ORD_U; u++) {
            const bool valid = v[u] != NOMATCH;
            const unsigned long long m = __ballot(valid);
            if (valid && cnt[k + __popcll(m & lt)] < 0) p[x0 + u * 64 + lane] = (int16_t)NOMATCH;
            k += __popcll(m);
        }
    }
}

void launch_order(const StageArgs &a, hipStream_t st) {
    int rows = 0, maxL = 0;
    for (int v = 0; v < a.ndir; v++) {
        rows = max(rows, a.d[v].own.YR - a.d[v].own.YL + 1);
        maxL = max(maxL, a.d[v].own.XR - a.d[v].own.XL + 1);
    }
    if (rows <= 0 || maxL <= 0) return;
    maxL = (maxL + 63) & ~63;
    const size_t lds = (size_t)maxL * 6 + ((size_t)maxL / 64 + 1) * 12 + ((size_t)maxL / ORD_BIG + 1) * 4;
    if (lds > 65536) // margins wider than ~10 k columns: beyond the default 64 KB of dynamic LDS per workgroup
        (void)hipFuncSetAttribute((const void *)k_order, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(k_order, dim3(rows, 1, a.ndir), dim3(64 * ORD_NW), lds, st, a, maxL);
}

// ---------------------------------------------------------------- UniquenessContraint_<T>
// Sequential dependency of the reference: the rescue test at .cpp:492 reads p[x-1] AFTER it may have
// been killed.  With  g(x) = valid && (direct || rescue by the untouched p[x+1])  and
// h(x) = valid && !direct && |q[bL+1] + p_orig[x-1]| < 2,  alive(x) = g(x) || (h(x) && alive(x-1)):
// a carry chain, resolved per 64-pixel chunk with ballots and a Kogge-Stone prefix on the masks.
template <typename T>
__device__ __forceinline__ bool close2(T q, T p);
template <>
__device__ __forceinline__ bool close2<int>(int q, int p) { return abs(q + p) < 2; }
template <>
__device__ __forceinline__ bool close2<double>(double q, double p) { return fabs(q + p) < 2; }

template <typename T, typename CT> // T storage (int16_t/double), CT compute (int/double)
__global__ void k_uniq(T *__restrict__ P, const T *__restrict__ Q, int W, int H, Mg own, Mg oth) {
    const int lane = threadIdx.x & 63;
    const int y = own.YL + blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (y > own.YR) return;
    const int XL = own.XL, XR = own.XR, XL1 = oth.XL, XR1 = oth.XR;
    const long long total = (long long)W * H;
    T *p = P + (size_t)y * W;
    unsigned long long cin = 1ull; // p[XL-1] is never modified: "alive" as far as the chain goes
    CT prev_last = (XL - 1 >= 0) ? (CT)p[XL - 1] : (CT)NOMATCH;
    for (int x0 = XL; x0 <= XR; x0 += 64) {
        const int x = x0 + lane;
        const CT raw = (x <= XR + 1 && x < W) ? (CT)p[x] : (CT)NOMATCH;
        const bool valid = (x <= XR) && (raw != (CT)NOMATCH);
        CT pm1 = (CT)__shfl_up(raw, 1);
        if (lane == 0) pm1 = prev_last;
        CT pp1 = (CT)__shfl_down(raw, 1);
        if (lane == 63) pp1 = (x + 1 <= XR + 1 && x + 1 < W) ? (CT)p[x + 1] : (CT)NOMATCH;
        bool g = false, h = false;
        if (valid) {
            const int bL = max((int)(raw + 0.5) + x - 1, XL1); // .cpp:482 (C truncation)
            const int bR = min(bL + 2, XR1);
            // all four loads at once (a short-circuit loop would make them dependent round trips)
            const long long rowo = (long long)y * W;
            CT qq[3];
#pragma unroll
            for (int u = 0; u < 3; u++) {
                const long long fi = rowo + bL + u; // bL + u <= bR is inside the row; beyond it the value is unused
                qq[u] = (bL + u <= bR && fi >= 0 && fi < total) ? (CT)Q[fi] : (CT)NOMATCH;
            }
            const long long fi = rowo + bL + 1; // flat-buffer emulation of q[bL+1]
            const CT qv = (fi >= 0 && fi < total) ? (CT)Q[fi] : (CT)NOMATCH;
            bool direct = false;
#pragma unroll
            for (int u = 0; u < 3; u++) direct = direct || (bL + u <= bR && close2<CT>(qq[u], raw));
            if (direct) g = true;
            else {
                g = close2<CT>(qv, pp1);
                h = close2<CT>(qv, pm1);
            }
        }
        unsigned long long G = __ballot(g), Pm = __ballot(h);
        G |= Pm & cin; // carry into lane 0
        for (int s = 1; s < 64; s <<= 1) {
            G |= Pm & (G << s);
            Pm &= (Pm << s);
        }
        const bool alive = (G >> lane) & 1ull;
        if (valid && !alive) p[x] = (T)NOMATCH;
        cin = (G >> 63) & 1ull;
        prev_last = (CT)__shfl(raw, 63);
    }
}

void launch_uniq_s16(int16_t *p, const int16_t *q, int W, int H, Mg own, Mg oth, hipStream_t st) {
    const int rows = own.YR - own.YL + 1;
    if (rows <= 0 || own.XR < own.XL) return;
    hipLaunchKernelGGL((k_uniq<int16_t, int>), dim3((rows + 3) / 4), dim3(256), 0, st, p, q, W, H, own, oth);
}
void launch_uniq_f64(double *p, const double *q, int W, int H, Mg own, Mg oth, hipStream_t st) {
    const int rows = own.YR - own.YL + 1;
    if (rows <= 0 || own.XR < own.XL) return;
    hipLaunchKernelGGL((k_uniq<double, double>), dim3((rows + 3) / 4), dim3(256), 0, st, p, q, W, H, own, oth);
}

// ---------------------------------------------------------------- SetBoundary_smooth<short>
#define MAX_DISPARITY 2 // .cpp:4
// vertical sweeps (.cpp:842-901): columns are independent, rows are a serial recurrence on the (bl, br) pair --
// but a recurrence of max-plus (bl) / min-plus (br) maps  x -> max(x + a, c)  only (assign a constant, step by
// MAX_DISPARITY against a floor, or for the upward sweep against the downward result of the row above).  So a
// column is cut into SBV_S row segments:
//   k_setb_vert<UP, 0>  runs the reference's own row step over a segment on two probe pairs and stores the
//                       segment's map (a, c) per bound (4 ints);
//   k_setb_vert<UP, 1>  starts each segment from the true incoming pair (the maps of the segments before it
//                       applied in order) and runs the same row step again, now storing the rows.
// Threads = 64 columns x SBV_S segments per direction instead of one thread per column.
// The upward sweep reads the downward result of the row above the one it is at; at a segment's last row that row
// belongs to the neighbouring segment, whose thread overwrites it first thing, so the summary pass saves it.
#define SBV_U 16              // rows loaded per batch
#define SBV_LO (-(1 << 24))   // probes: far outside any value the recurrence can produce or clamp to
#define SBV_HI (1 << 24)
#define SBV_INF (1 << 26)
// scratch (int32, in rf_list): per direction [SBV_S][W][6] = map (a_bl, c_bl, a_br, c_br) + saved pair, then [W][2]
__device__ __forceinline__ int *sbv_scratch(const StageArgs &a, int dir) {
    return (int *)a.rf_list + (size_t)dir * SETB_SCRATCH(a.W);
}

struct SbvPair {
    int bl, br;
};
// one row of .cpp:844-868 (downward; UP = 0) or .cpp:876-899 (upward; UP = 1, (ul, ur) = the pair stored at the
// row above by the downward sweep): `out` is what the row stores, `s` becomes the pair handed to the next row
template <int UP>
__device__ __forceinline__ void sbv_step(int m, int r, int ul, int ur, SbvPair &s, SbvPair &out) {
    int nbl = UP ? ul : -10000, nbr = UP ? ur : 10000; // .cpp:832-833 / :880-881
    if (m == 255) {
        if (r != NOMATCH) {
            s.bl = r;
            s.br = r;
        }
        nbl = max(s.bl - MAX_DISPARITY, nbl);
        nbr = min(s.br + MAX_DISPARITY, nbr);
    }
    out = s;
    s.bl = nbl;
    s.br = nbr;
}

template <int UP, int APPLY>
__global__ __launch_bounds__(64) void k_setb_vert(StageArgs a) {
    const DirArgs &d = a.d[blockIdx.z];
    const int x = d.own.XL + blockIdx.x * 64 + (int)threadIdx.x;
    if (x > d.own.XR) return;
    const int W = a.W, YL = d.own.YL, YR = d.own.YR, seg = blockIdx.y;
    const int R = YR - YL, L = (R + SBV_S - 1) / SBV_S, base = YL + UP; // rows base .. base + R - 1 are swept
    const int y0 = base + seg * L, y1 = min(y0 + L, base + R);          // this segment: rows [y0, y1), maybe empty
    const uint8_t *__restrict__ mk = d.mask_own;
    const int16_t *__restrict__ src = d.d16_in;
    int16_t *BL = d.BL, *BR = d.BR;
    int *sc = sbv_scratch(a, blockIdx.z);
    int *initp = sc + (size_t)SBV_S * W * 6 + (size_t)x * 2;
    auto rec = [&](int k) { return sc + ((size_t)k * W + x) * 6; };

    SbvPair s0, s1; // APPLY: s0 = the true pair.  Summary: s0 = (LO, HI) gives the maps' constants, s1 = (HI, LO) their slopes
    int sav_l = 0, sav_r = 0;
    if (APPLY) {
        if (UP) { // from the pair the downward sweep ended with (:870-874), through the segments below this one
            s0.bl = initp[0];
            s0.br = initp[1];
            for (int k = SBV_S - 1; k > seg; k--) {
                const int *q = rec(k);
                s0.bl = max(s0.bl + q[0], q[1]);
                s0.br = min(s0.br + q[2], q[3]);
            }
            sav_l = rec(seg)[4];
            sav_r = rec(seg)[5];
        } else {
            s0.bl = -10000;
            s0.br = 10000;
            for (int k = 0; k < seg; k++) {
                const int *q = rec(k);
                s0.bl = max(s0.bl + q[0], q[1]);
                s0.br = min(s0.br + q[2], q[3]);
            }
        }
        s1 = s0;
    } else {
        s0.bl = SBV_LO;
        s0.br = SBV_HI;
        s1.bl = SBV_HI;
        s1.br = SBV_LO;
        if (UP && seg == 0) { // rows are intact during the summary pass
            initp[0] = BL[(size_t)YR * W + x];
            initp[1] = BR[(size_t)YR * W + x];
        }
    }
    // the segment's rows, SBV_U at a time (all loads of a batch before its stores), in sweep order
    for (int t0 = 0; t0 < y1 - y0; t0 += SBV_U) {
        int m[SBV_U], r[SBV_U], ul[SBV_U], ur[SBV_U];
#pragma unroll
        for (int i = 0; i < SBV_U; i++) {
            const int t = min(t0 + i, y1 - y0 - 1);
            const int y = UP ? y1 - 1 - t : y0 + t;
            const size_t o = (size_t)y * W + x;
            m[i] = mk[o];
            r[i] = src[o];
            if (UP) {
                ul[i] = BL[o - W];
                ur[i] = BR[o - W];
                if (APPLY && y == y0) { // the neighbouring segment's thread may already have overwritten that row
                    ul[i] = sav_l;
                    ur[i] = sav_r;
                }
            } else {
                ul[i] = ur[i] = 0;
            }
        }
#pragma unroll
        for (int i = 0; i < SBV_U; i++) {
            if (t0 + i < y1 - y0) {
                const int y = UP ? y1 - 1 - (t0 + i) : y0 + t0 + i;
                SbvPair out;
                sbv_step<UP>(m[i], r[i], ul[i], ur[i], s0, out);
                if (APPLY) {
                    BL[(size_t)y * W + x] = (int16_t)out.bl;
                    BR[(size_t)y * W + x] = (int16_t)out.br;
                } else {
                    sbv_step<UP>(m[i], r[i], ul[i], ur[i], s1, out);
                    if (UP && y == y0) {
                        sav_l = ul[i];
                        sav_r = ur[i];
                    }
                }
            }
        }
    }
    if (APPLY) {
        // the sweep's last store: row YR after the downward sweep (:870-874), row YL after the upward one (:900-901)
        if (y0 < y1 && (UP ? seg == 0 : y1 == base + R)) {
            const size_t o = (size_t)(UP ? YL : YR) * W + x;
            BL[o] = (int16_t)s0.bl;
            BR[o] = (int16_t)s0.br;
        }
    } else {
        int *q = rec(seg);
        q[0] = s1.bl > (1 << 23) ? s1.bl - SBV_HI : -SBV_INF; // slope part survived: x + a, else a constant map
        q[1] = s0.bl;
        q[2] = s1.br < -(1 << 23) ? s1.br - SBV_LO : SBV_INF;
        q[3] = s0.br;
        q[4] = sav_l;
        q[5] = sav_r;
    }
}

// horizontal sweeps (.cpp:903-941): sequential along x in the reference, but both are max-plus / min-plus
// recurrences, i.e. segmented prefix scans:
//   left -> right (:909-916)   bl[x] = max(bl[x], bl[x-1] - 1), br[x] = min(br[x], br[x-1] + 2) where mask[x-1] == 255
//     <=> bl[x] + x = running max of (bl + x), br[x] - 2x = running min of (br - 2x) over the run of linked pixels;
//   right -> left (:917-931)   with c = value arriving at x: where mask[x] == 255 the pixel stores the absolute
//     columns max(c + x, XL1) / min(c' + x, XR1) and hands max(c, XL1 - x) - 2 / min(c', XR1 - x) + 1 on to x - 1
//     <=> c[x] - 2x = running max (from the right) of u - 2x, c'[x] + x = running min of u' + x, where
//     u[x] = max(bl[x], XL1 - x - 3), u'[x] = min(br[x], XR1 - x) at pixels that have a link from x + 1.
// One wave per row walks it in 64-column chunks with a segmented wave scan (6 shuffle steps) and a carry.
// All values stay within +-(10000 + W), so the reference's int16 stores never wrap and the scan is exact.
__global__ __launch_bounds__(256) void k_setb_horiz(StageArgs a, int emit_list) {
    const DirArgs &d = a.d[blockIdx.z];
    const int lane = threadIdx.x & 63;
    const int y = d.own.YL + blockIdx.x * 4 + ((int)threadIdx.x >> 6);
    if (y > d.own.YR) return; // wave-uniform
    const int W = a.W, XL = d.own.XL, XR = d.own.XR, XL1 = d.oth.XL, XR1 = d.oth.XR;
    int16_t *__restrict__ bl = d.BL + (size_t)y * W;
    int16_t *__restrict__ br = d.BR + (size_t)y * W;
    const uint8_t *__restrict__ mk = d.mask_own + (size_t)y * W;
    const int nchunk = (XR - XL) / 64 + 1;
    // ---- left -> right
    int carryL = 0, carryR = 0;
    for (int k = 0; k < nchunk; k++) {
        const int x = XL + k * 64 + lane;
        const bool in = x <= XR;
        const bool link = in && x > XL && mk[x - 1] == 255;
        int VL = in ? (int)bl[x] + x : 0, VR = in ? (int)br[x] - 2 * x : 0;
        const unsigned long long heads = __ballot(!link);
        const unsigned long long below = heads & ((2ull << lane) - 1ull); // lanes <= this one
        const int head = below ? 63 - __builtin_clzll(below) : -1;       // -1: the run started in an earlier chunk
        const int lo = head < 0 ? 0 : head;
#pragma unroll
        for (int s = 1; s < 64; s <<= 1) {
            const int oL = __shfl_up(VL, s), oR = __shfl_up(VR, s);
            if (lane - s >= lo) {
                VL = max(VL, oL);
                VR = min(VR, oR);
            }
        }
        if (head < 0) {
            VL = max(VL, carryL);
            VR = min(VR, carryR);
        }
        if (link) {
            bl[x] = (int16_t)(VL - x);
            br[x] = (int16_t)(VR + 2 * x);
        }
        carryL = __shfl(VL, 63);
        carryR = __shfl(VR, 63);
    }
    // ---- right -> left (each lane re-reads the columns it wrote itself above)
    uint32_t *rowlist = a.rf_list + ((size_t)blockIdx.z * a.H + y) * W; // emit_list: this row's Rematch pixels
    int nemit = 0;
    for (int k = nchunk - 1; k >= 0; k--) {
        const int x = XL + k * 64 + lane;
        const bool in = x <= XR;
        bool emit = false;
        const int m = in ? (int)mk[x] : 0;
        const bool link = in && x < XR && mk[x + 1] == 255; // the pixel receives from x + 1
        int tl = in ? (int)bl[x] : 0, tr = in ? (int)br[x] : 0;
        if (link) {
            tl = max(tl, XL1 - x - 3);
            tr = min(tr, XR1 - x);
        }
        int VL = tl - 2 * x, VR = tr + x;
        const unsigned long long heads = __ballot(!link);
        const unsigned long long above = heads & ~((1ull << lane) - 1ull); // lanes >= this one
        const int head = above ? __builtin_ctzll(above) : 64;             // 64: the run continues in the chunk to the right
        const int hi = head > 63 ? 63 : head;
#pragma unroll
        for (int s = 1; s < 64; s <<= 1) {
            const int oL = __shfl_down(VL, s), oR = __shfl_down(VR, s);
            if (lane + s <= hi) {
                VL = max(VL, oL);
                VR = min(VR, oR);
            }
        }
        if (head == 64) {
            VL = max(VL, carryL);
            VR = min(VR, carryR);
        }
        carryL = __shfl(VL, 0);
        carryR = __shfl(VR, 0);
        if (in) {
            const int cl = VL + 2 * x, cr = VR - x; // the values arriving at x
            int A = cl, B = cr;
            if (m == 255) {
                A = (int16_t)(cl + x);
                B = (int16_t)(cr + x);
                if (x > XL) {
                    if (A < XL1) A = XL1;
                    if (B > XR1) B = XR1;
                } else { // .cpp:932-940 with the bl/br typo of :938-939
                    if (A < XL1) A = XL1;
                    if (B > XR1) A = XR1;
                }
            }
            bl[x] = (int16_t)A;
            br[x] = (int16_t)B;
            // Rematch evaluates NCC only where the pixel is masked, still unmatched (.cpp:538) and its interval is
            // not empty after the window clamp: about 1 % of a level.  Those go to per-row lists for k_ncc_sparse.
            emit = emit_list && m == 255 && (int)d.d16_in[(size_t)y * W + x] == NOMATCH &&
                   max(A, a.r) <= min(B, W - 1 - a.r);
        }
        if (emit_list) { // uniform: the row's own list, no atomics (a shared counter would see ~30 appends per row)
            const unsigned long long mm = __ballot(emit);
            if (emit) rowlist[nemit + __popcll(mm & ((1ull << lane) - 1ull))] = (uint32_t)((size_t)y * W + x);
            nemit += __popcll(mm);
        }
    }
    if (emit_list && lane == 0) a.ncc_cnt[16 + blockIdx.z * a.H + y] = nemit;
}

void launch_set_boundary(const StageArgs &a, hipStream_t st, bool emit_list) {
    int rows = 0, cols = 0;
    for (int v = 0; v < a.ndir; v++) {
        rows = max(rows, a.d[v].own.YR - a.d[v].own.YL + 1);
        cols = max(cols, a.d[v].own.XR - a.d[v].own.XL + 1);
    }
    if (rows <= 0 || cols <= 0) return;
    const dim3 vgrid((cols + 63) / 64, SBV_S, a.ndir);
    hipLaunchKernelGGL((k_setb_vert<0, 0>), vgrid, dim3(64), 0, st, a);
    hipLaunchKernelGGL((k_setb_vert<0, 1>), vgrid, dim3(64), 0, st, a);
    hipLaunchKernelGGL((k_setb_vert<1, 0>), vgrid, dim3(64), 0, st, a);
    hipLaunchKernelGGL((k_setb_vert<1, 1>), vgrid, dim3(64), 0, st, a);
    // (the vertical sweeps are done with their scratch in rf_list: the horizontal kernel may refill it)
    hipLaunchKernelGGL(k_setb_horiz, dim3((rows + 3) / 4, 1, a.ndir), dim3(256), 0, st, a, emit_list ? 1 : 0);
}

// ---------------------------------------------------------------- MedianFilter (1 iteration)
__global__ void k_median(StageArgs a) {
    const DirArgs &d = a.d[blockIdx.z];
    const int x = d.own.XL + blockIdx.x * blockDim.x + threadIdx.x;
    const int y = d.own.YL + blockIdx.y;
    if (x > d.own.XR || y > d.own.YR) return;
    const int W = a.W;
    const int16_t *D = d.d16_in;
    // mask and the six samples (.cpp:792: columns x-1 and x only) in one round trip
    const int m = d.mask_own[(size_t)y * W + x];
    int u[6];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) u[i * 3 + j] = D[(size_t)(y + j - 1) * W + x - 1 + i];
    if (m != 255) return; // output stays NOMATCH (.cpp:772,788)
    const int c = u[4];
    int k = 0;
#pragma unroll
    for (int i = 0; i < 6; i++) {
        k += u[i] != NOMATCH;
        if (u[i] == NOMATCH) u[i] = 0x7fffffff; // sorts behind the k valid samples
    }
    int out = NOMATCH;
    const bool take = (c == NOMATCH) ? (k >= 4) : (k > 2);
    if (take) {
        // 12-exchange sorting network for 6 keys (registers only; the reference sorts the k valid ones)
#define MED_CE(A, B)                 \
    {                                \
        const int lo_ = min(u[A], u[B]); \
        u[B] = max(u[A], u[B]);      \
        u[A] = lo_;                  \
    }
        MED_CE(0, 5) MED_CE(1, 3) MED_CE(2, 4) MED_CE(1, 2) MED_CE(3, 4) MED_CE(0, 3)
        MED_CE(2, 5) MED_CE(0, 1) MED_CE(2, 3) MED_CE(4, 5) MED_CE(1, 2) MED_CE(3, 4)
#undef MED_CE
        // arma::median: odd -> middle; even -> lo + (hi - lo)/2 (op_mean::robust_mean, integer division); k = 3..6
        const int lo = (k <= 4) ? u[1] : u[2]; // k = 3: the middle, 4: the lower middle, 5: the middle, 6: the lower middle
        const int hi = (k == 6) ? u[3] : u[2]; // the upper middle of an even k
        out = (k & 1) ? lo : lo + (hi - lo) / 2;
    }
    d.d16_out[(size_t)y * W + x] = (int16_t)out;
}

void launch_median(const StageArgs &a, hipStream_t st) {
    int rows = 0, cols = 0;
    for (int v = 0; v < a.ndir; v++) {
        rows = max(rows, a.d[v].own.YR - a.d[v].own.YL + 1);
        cols = max(cols, a.d[v].own.XR - a.d[v].own.XL + 1);
    }
    if (rows <= 0 || cols <= 0) return;
    hipLaunchKernelGGL(k_median, dim3((cols + 255) / 256, rows, a.ndir), dim3(256), 0, st, a);
}
