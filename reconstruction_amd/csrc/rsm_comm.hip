// rsm_comm.hip -- multi-GPU exchange of the C ABI: one process per GPU, the per-pair clouds gathered to one rank with
// RCCL over xGMI.  Replaces the process-global accumulation of the reference (CloudOptimization/
// CCloudOptimization.cpp:59-62 InsertPoint, :123 `cloud_in += cloud`) fed by the sequential pair loop
// (CStereoMatching.cpp:17-33): pairs are matched where they live, only 16-byte point records travel.
//
// The exchange is a FAN-IN, not a ring: xGMI is point to point, every peer owns a link to the root, so all peers
// send at once and nothing is forwarded (a ring all-gather would be per-link bound and move world-1 times the data).
//   1. one all-reduce of 3 * n_pairs + 2 int64 (per pair: point count, number of claims, owner; an error count; the
//      root's capacity) -- sizes the receives, and lets every rank reach the SAME verdict (bad argument anywhere, a pair
//      claimed twice, root too small) before any payload is posted, so nobody waits for a peer that has already left;
//   2. one group: the root posts a receive per remote pair straight into its slot of the output (pair order), every
//      other rank a send per local pair; the root's own pairs are local copies.  Point-to-point operations between
//      two ranks match in posting order, so BOTH sides walk the pairs in ascending pair id (rsm_gather_plan).
// The protocol (rsm_gather_meta_fill / rsm_gather_plan) is transport-free; rsm_gather_clouds runs it over an
// rsm_transport table: RCCL (rsm_comm_create) or whatever the caller supplies (rsm_comm_create_transport; the CPU tests
// run world 2/3/8 over an in-process mock).
// librccl is opened at run time (dlopen), so the library loads and every other entry point works without RCCL, and a
// process that already carries an RCCL (PyTorch's) shares that one.
#include "../../include/rsm.h"
#include "rsm_dev.h"

#include <dlfcn.h>
#include <rccl/rccl.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <vector>

namespace {
struct Rccl {
    void *h = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
};
Rccl load_rccl() {
    Rccl r;
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char *n : names) {
        r.h = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        if (r.h) break;
    }
    if (!r.h) return r;
#define SYM(field, name) *(void **)(&r.field) = dlsym(r.h, name)
    SYM(GetUniqueId, "ncclGetUniqueId");
    SYM(CommInitRank, "ncclCommInitRank");
    SYM(CommDestroy, "ncclCommDestroy");
    SYM(AllReduce, "ncclAllReduce");
    SYM(Send, "ncclSend");
    SYM(Recv, "ncclRecv");
    SYM(GroupStart, "ncclGroupStart");
    SYM(GroupEnd, "ncclGroupEnd");
    SYM(GetErrorString, "ncclGetErrorString");
#undef SYM
    r.ok = r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.AllReduce && r.Send && r.Recv && r.GroupStart && r.GroupEnd;
    return r;
}
Rccl &rccl() {
    static Rccl r = load_rccl(); // initialised once, thread-safe (C++11)
    return r;
}
} // namespace

struct rsm_comm {
    rsm_transport tp{};
    bool own_rccl = false;
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1, device = 0;
    hipStream_t stream = nullptr;
    int64_t *d_meta = nullptr; // RSM_GATHER_META_WORDS(RSM_COMM_MAX_PAIRS) int64
    std::string err;
    std::string tp_err;
};

static int comm_err(rsm_comm *c, int code, const char *what, const char *detail) {
    if (c) c->err = std::string(what) + ": " + (detail ? detail : "");
    return code;
}

// ---- the RCCL transport ---------------------------------------------------------------------------------------------
namespace {
int rccl_fail(rsm_comm *c, const char *what, const char *detail) {
    c->tp_err = std::string(what) + ": " + (detail ? detail : "");
    return 1;
}
#define TN(c, call)                                                                                                   \
    do {                                                                                                              \
        ncclResult_t r__ = (call);                                                                                    \
        if (r__ != ncclSuccess) return rccl_fail((c), #call, rccl().GetErrorString ? rccl().GetErrorString(r__) : "rccl error"); \
    } while (0)
#define TH(c, call)                                                              \
    do {                                                                         \
        hipError_t e__ = (call);                                                 \
        if (e__ != hipSuccess) return rccl_fail((c), #call, hipGetErrorString(e__)); \
    } while (0)

int rccl_allreduce(void *self, int64_t *host, int n) {
    rsm_comm *c = (rsm_comm *)self;
    if (n > RSM_GATHER_META_WORDS(RSM_COMM_MAX_PAIRS)) return rccl_fail(c, "allreduce", "too many words");
    TH(c, hipSetDevice(c->device));
    TH(c, hipMemcpyAsync(c->d_meta, host, sizeof(int64_t) * n, hipMemcpyHostToDevice, c->stream));
    TN(c, rccl().AllReduce(c->d_meta, c->d_meta, (size_t)n, ncclInt64, ncclSum, c->comm, c->stream));
    TH(c, hipMemcpyAsync(host, c->d_meta, sizeof(int64_t) * n, hipMemcpyDeviceToHost, c->stream));
    TH(c, hipStreamSynchronize(c->stream));
    return 0;
}
int rccl_group_begin(void *self) {
    rsm_comm *c = (rsm_comm *)self;
    TH(c, hipSetDevice(c->device));
    TN(c, rccl().GroupStart());
    return 0;
}
int rccl_send(void *self, const void *buf, uint64_t bytes, int peer) {
    rsm_comm *c = (rsm_comm *)self;
    TN(c, rccl().Send(buf, (size_t)bytes, ncclUint8, peer, c->comm, c->stream));
    return 0;
}
int rccl_recv(void *self, void *buf, uint64_t bytes, int peer) {
    rsm_comm *c = (rsm_comm *)self;
    TN(c, rccl().Recv(buf, (size_t)bytes, ncclUint8, peer, c->comm, c->stream));
    return 0;
}
int rccl_copy(void *self, void *dst, const void *src, uint64_t bytes) {
    rsm_comm *c = (rsm_comm *)self;
    TH(c, hipMemcpyAsync(dst, src, (size_t)bytes, hipMemcpyDeviceToDevice, c->stream));
    return 0;
}
int rccl_group_end(void *self) {
    rsm_comm *c = (rsm_comm *)self;
    TN(c, rccl().GroupEnd());
    TH(c, hipStreamSynchronize(c->stream));
    return 0;
}
} // namespace

extern "C" int rsm_comm_unique_id(char id[RSM_COMM_ID_BYTES]) {
    if (!id) return RSM_E_INVALID;
    Rccl &R = rccl();
    if (!R.ok) return RSM_E_COMM;
    static_assert(RSM_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "id size");
    ncclUniqueId u;
    if (R.GetUniqueId(&u) != ncclSuccess) return RSM_E_COMM;
    memcpy(id, u.internal, RSM_COMM_ID_BYTES);
    return RSM_OK;
}

extern "C" int rsm_comm_create(rsm_comm **out, const char id[RSM_COMM_ID_BYTES], int rank, int world, int hip_device) {
    if (!out || !id || world < 1 || rank < 0 || rank >= world) return RSM_E_INVALID;
    *out = nullptr;
    Rccl &R = rccl();
    if (!R.ok) return RSM_E_COMM;
    if (hipSetDevice(hip_device) != hipSuccess) return RSM_E_HIP;
    rsm_comm *c = new rsm_comm();
    c->rank = rank;
    c->world = world;
    c->device = hip_device;
    c->own_rccl = true;
    ncclUniqueId u;
    memcpy(u.internal, id, RSM_COMM_ID_BYTES);
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess ||
        hipMalloc((void **)&c->d_meta, sizeof(int64_t) * RSM_GATHER_META_WORDS(RSM_COMM_MAX_PAIRS)) != hipSuccess ||
        R.CommInitRank(&c->comm, world, u, rank) != ncclSuccess) {
        if (c->d_meta) (void)hipFree(c->d_meta);
        if (c->stream) (void)hipStreamDestroy(c->stream);
        delete c;
        return RSM_E_COMM;
    }
    c->tp = rsm_transport{c, rccl_allreduce, rccl_group_begin, rccl_send, rccl_recv, rccl_copy, rccl_group_end};
    *out = c;
    return RSM_OK;
}

extern "C" int rsm_comm_create_transport(rsm_comm **out, const rsm_transport *tp, int rank, int world) {
    if (!out || !tp || world < 1 || rank < 0 || rank >= world || !tp->allreduce_sum_i64 || !tp->group_begin || !tp->send ||
        !tp->recv || !tp->copy || !tp->group_end)
        return RSM_E_INVALID;
    rsm_comm *c = new rsm_comm();
    c->rank = rank;
    c->world = world;
    c->tp = *tp;
    *out = c;
    return RSM_OK;
}

extern "C" void rsm_comm_destroy(rsm_comm *c) {
    if (!c) return;
    if (c->own_rccl) {
        (void)hipSetDevice(c->device);
        (void)hipStreamSynchronize(c->stream);
        if (c->comm) (void)rccl().CommDestroy(c->comm);
        (void)hipFree(c->d_meta);
        (void)hipStreamDestroy(c->stream);
    }
    delete c;
}

extern "C" const char *rsm_comm_last_error(const rsm_comm *c) { return c ? c->err.c_str() : "null comm"; }

// ---- the protocol (host arithmetic only) ----------------------------------------------------------------------------
// meta[p] = points of pair p, meta[P + p] = ranks claiming it, meta[2P + p] = the claimant's rank (meaningful when the
// claim count is 1), meta[3P] = ranks with a bad argument, meta[3P + 1] = the root's capacity in records.
extern "C" int rsm_gather_meta_fill(int rank, int world, int root, int n_local, const int *pair_ids, const int64_t *n_points,
                                    int n_pairs_total, int64_t max_out, int64_t *meta) {
    const int P = n_pairs_total;
    if (!meta || P < 0 || P > RSM_COMM_MAX_PAIRS) return RSM_E_INVALID;
    memset(meta, 0, sizeof(int64_t) * RSM_GATHER_META_WORDS(P));
    int st = RSM_OK;
    if (world < 1 || rank < 0 || rank >= world || root < 0 || root >= world || n_local < 0 || (n_local > 0 && (!pair_ids || !n_points)))
        st = RSM_E_INVALID;
    for (int i = 0; st == RSM_OK && i < n_local; i++) {
        const int p = pair_ids[i];
        if (p < 0 || p >= P || n_points[i] < 0) {
            st = RSM_E_INVALID;
            break;
        }
        meta[p] += n_points[i];
        meta[(size_t)P + p] += 1; // a pair listed twice by this rank shows up as two claims
        meta[(size_t)2 * P + p] += rank;
    }
    if (st != RSM_OK) { // this rank's contribution is the error alone: the others must not act on half a list
        memset(meta, 0, sizeof(int64_t) * RSM_GATHER_META_WORDS(P));
        meta[(size_t)3 * P] = 1;
    }
    if (rank == root) meta[(size_t)3 * P + 1] = max_out > 0 ? max_out : 0;
    return st;
}

// 0 = go; the verdict is a function of the SUMMED vector only, so every rank reaches the same one
static const char *const VERDICT_TEXT[] = {"ok", "bad argument (pair id, count or buffer) on some rank", "a pair is held by two ranks",
                                           "the root's output capacity is too small for the gathered clouds"};
static int meta_verdict(const int64_t *meta, int P, int world) {
    if (meta[(size_t)3 * P] != 0) return 1;
    int64_t total = 0;
    for (int p = 0; p < P; p++) {
        const int64_t claims = meta[(size_t)P + p];
        if (claims > 1) return 2; // held by two ranks, or listed twice by one
        if (claims < 0 || (claims == 1 && (meta[(size_t)2 * P + p] < 0 || meta[(size_t)2 * P + p] >= world)) || meta[p] < 0 ||
            (claims == 0 && meta[p] != 0))
            return 1;
        total += meta[p];
    }
    return total > meta[(size_t)3 * P + 1] ? 3 : 0;
}

extern "C" int rsm_gather_plan(int rank, int world, int root, int n_local, const int *pair_ids, const int64_t *n_points,
                               int n_pairs_total, const int64_t *meta, int64_t *offsets, rsm_gather_op *ops, int max_ops,
                               int *n_ops) {
    const int P = n_pairs_total;
    if (!meta || !offsets || !n_ops || P < 0 || P > RSM_COMM_MAX_PAIRS || world < 1 || rank < 0 || rank >= world || root < 0 ||
        root >= world)
        return RSM_E_INVALID;
    *n_ops = 0;
    offsets[0] = 0;
    for (int p = 0; p < P; p++) offsets[p + 1] = offsets[p] + (meta[p] > 0 ? meta[p] : 0);
    if (meta_verdict(meta, P, world) != 0) return RSM_E_INVALID;
    int n = 0;
    if (rank == root) {
        // receives: ascending pair id (hence ascending within every peer) -- the order the peers send in
        for (int p = 0; p < P; p++) {
            if (meta[(size_t)P + p] != 1 || meta[p] == 0) continue;
            const int owner = (int)meta[(size_t)2 * P + p];
            if (owner == root) continue;
            if (n >= max_ops || !ops) return RSM_E_INVALID;
            ops[n++] = rsm_gather_op{1, owner, p, -1, offsets[p], meta[p]};
        }
    }
    // this rank's own pairs in ascending pair id, whatever order the caller listed them in
    std::vector<int> idx;
    for (int i = 0; i < n_local; i++)
        if (n_points[i] > 0) idx.push_back(i);
    std::sort(idx.begin(), idx.end(), [&](int a, int b) { return pair_ids[a] < pair_ids[b]; });
    for (int i : idx) {
        if (n >= max_ops || !ops) return RSM_E_INVALID;
        ops[n++] = rsm_gather_op{rank == root ? 2 : 0, root, pair_ids[i], i, offsets[pair_ids[i]], n_points[i]};
    }
    *n_ops = n;
    return RSM_OK;
}

extern "C" int rsm_gather_counts(rsm_comm *c, int n_local, const int *pair_ids, const int64_t *n_points, int n_pairs_total,
                                 int64_t *counts) {
    if (!c || !counts || n_pairs_total < 0 || n_pairs_total > RSM_COMM_MAX_PAIRS) return RSM_E_INVALID;
    const int P = n_pairs_total;
    std::vector<int64_t> meta((size_t)RSM_GATHER_META_WORDS(P), 0);
    (void)rsm_gather_meta_fill(c->rank, c->world, 0, n_local, pair_ids, n_points, P, 0, meta.data());
    if (c->tp.allreduce_sum_i64(c->tp.self, meta.data(), (int)meta.size()) != 0)
        return comm_err(c, RSM_E_COMM, "all-reduce of the gather metadata", c->tp_err.c_str());
    meta[(size_t)3 * P + 1] = INT64_MAX; // no capacity question here
    const int v = meta_verdict(meta.data(), P, c->world);
    if (v) return comm_err(c, RSM_E_INVALID, "rsm_gather_counts", VERDICT_TEXT[v]);
    for (int p = 0; p < P; p++) counts[p] = meta[p];
    return RSM_OK;
}

extern "C" int rsm_gather_clouds(rsm_comm *c, int root, int n_local, const int *pair_ids, const rsm_point16 *const *d_clouds,
                                 const int64_t *n_points, int n_pairs_total, rsm_point16 *d_out, int64_t max_out,
                                 int64_t *out_offsets) {
    // only what EVERY rank sees alike may end the call before the collective
    if (!c || root < 0 || root >= c->world || n_pairs_total < 0 || n_pairs_total > RSM_COMM_MAX_PAIRS) return RSM_E_INVALID;
    const int P = n_pairs_total;
    const rsm_transport &T = c->tp;
    std::vector<int64_t> meta((size_t)RSM_GATHER_META_WORDS(P), 0);
    const bool args_ok = n_local >= 0 && (n_local == 0 || (pair_ids && d_clouds && n_points));
    int st = rsm_gather_meta_fill(c->rank, c->world, root, args_ok ? n_local : -1, pair_ids, n_points, P,
                                  (c->rank == root && d_out) ? max_out : 0, meta.data());
    for (int i = 0; st == RSM_OK && i < n_local; i++)
        if (n_points[i] > 0 && !d_clouds[i]) { // a count without a buffer: report it through the collective too
            std::fill(meta.begin(), meta.begin() + 3 * (size_t)P, 0);
            meta[(size_t)3 * P] = 1;
            st = RSM_E_INVALID;
        }
    if (T.allreduce_sum_i64(T.self, meta.data(), (int)meta.size()) != 0) return comm_err(c, RSM_E_COMM, "all-reduce of the gather metadata", c->tp_err.c_str());
    std::vector<int64_t> off((size_t)P + 1, 0);
    std::vector<rsm_gather_op> ops((size_t)P + (size_t)(n_local > 0 ? n_local : 0) + 1);
    int n_ops = 0;
    const int verdict = rsm_gather_plan(c->rank, c->world, root, st == RSM_OK ? n_local : 0, pair_ids, n_points, P, meta.data(),
                                        off.data(), ops.data(), (int)ops.size(), &n_ops);
    if (verdict != RSM_OK) {
        const char *why = VERDICT_TEXT[meta_verdict(meta.data(), P, c->world)];
        return comm_err(c, RSM_E_INVALID, "rsm_gather_clouds", why);
    }
    if (c->rank == root && out_offsets) memcpy(out_offsets, off.data(), sizeof(int64_t) * ((size_t)P + 1));
    // payload fan-in: one group, every peer's sends and the root's receives in flight together.  group_end is always
    // reached, also after a failed post, so that whatever was posted completes and the transport is left usable.
    if (T.group_begin(T.self) != 0) return comm_err(c, RSM_E_COMM, "group begin", c->tp_err.c_str());
    int fail = 0;
    for (int k = 0; k < n_ops && !fail; k++) {
        const rsm_gather_op &o = ops[(size_t)k];
        const uint64_t bytes = (uint64_t)o.count * sizeof(rsm_point16);
        if (o.kind == 0) fail = T.send(T.self, d_clouds[o.local_index], bytes, o.peer);
        else if (o.kind == 1) fail = T.recv(T.self, d_out + o.offset, bytes, o.peer);
        else fail = T.copy(T.self, d_out + o.offset, d_clouds[o.local_index], bytes);
    }
    std::string first_err = c->tp_err;
    const int end_fail = T.group_end(T.self);
    if (fail) return comm_err(c, RSM_E_COMM, "posting the cloud transfers", first_err.c_str());
    if (end_fail) return comm_err(c, RSM_E_COMM, "group end", c->tp_err.c_str());
    return RSM_OK;
}
