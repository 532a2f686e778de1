// rsm_comm.hip -- multi-GPU exchange of the C ABI: one process per GPU, the per-pair clouds gathered to one rank with
// RCCL over xGMI.  Replaces the process-global accumulation of the reference (CloudOptimization/
// CCloudOptimization.cpp:59-62 InsertPoint, :123 `cloud_in += cloud`) fed by the sequential pair loop
// (CStereoMatching.cpp:17-33): pairs are matched where they live, only 16-byte point records travel.
//
// The exchange is a FAN-IN, not a ring: xGMI is point to point, every peer owns a link to the root, so all peers
// send at once and nothing is forwarded (a ring all-gather would be per-link bound and move world-1 times the data).
//   1. one ncclAllReduce of 2 * n_pairs int64 (point count and owner of every pair) -- sizes the receives -- and one of
//      a status word (can the root hold them?), so that all ranks fail together instead of half of them waiting;
//   2. one ncclGroup: the root posts a ncclRecv per remote pair straight into its slot of the output (pair order),
//      every other rank a ncclSend per local pair; the root's own pairs are device-to-device copies.
// librccl is opened at run time (dlopen), so the library loads and every other entry point works without RCCL, and a
// process that already carries an RCCL (PyTorch's) shares that one.
#include "../../include/rsm.h"
#include "rsm_dev.h"

#include <dlfcn.h>
#include <rccl/rccl.h>
#include <stdio.h>
#include <string.h>

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
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1, device = 0;
    hipStream_t stream = nullptr;
    int64_t *d_meta = nullptr; // 2 * RSM_COMM_MAX_PAIRS int64
    std::string err;
};

static int comm_err(rsm_comm *c, int code, const char *what, const char *detail) {
    if (c) c->err = std::string(what) + ": " + (detail ? detail : "");
    return code;
}
#define NCHK(c, call)                                                                                             \
    do {                                                                                                          \
        ncclResult_t r__ = (call);                                                                                \
        if (r__ != ncclSuccess) return comm_err((c), RSM_E_COMM, #call, R.GetErrorString ? R.GetErrorString(r__) : "rccl error"); \
    } while (0)
#define HCHK(c, call)                                                                      \
    do {                                                                                   \
        hipError_t e__ = (call);                                                           \
        if (e__ != hipSuccess) return comm_err((c), RSM_E_HIP, #call, hipGetErrorString(e__)); \
    } while (0)

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
    ncclUniqueId u;
    memcpy(u.internal, id, RSM_COMM_ID_BYTES);
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess ||
        hipMalloc((void **)&c->d_meta, sizeof(int64_t) * 2 * RSM_COMM_MAX_PAIRS) != hipSuccess ||
        R.CommInitRank(&c->comm, world, u, rank) != ncclSuccess) {
        if (c->d_meta) (void)hipFree(c->d_meta);
        if (c->stream) (void)hipStreamDestroy(c->stream);
        delete c;
        return RSM_E_COMM;
    }
    *out = c;
    return RSM_OK;
}

extern "C" void rsm_comm_destroy(rsm_comm *c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    if (c->comm) (void)rccl().CommDestroy(c->comm);
    (void)hipFree(c->d_meta);
    (void)hipStreamDestroy(c->stream);
    delete c;
}

extern "C" const char *rsm_comm_last_error(const rsm_comm *c) { return c ? c->err.c_str() : "null comm"; }

extern "C" int rsm_gather_clouds(rsm_comm *c, int root, int n_local, const int *pair_ids, const rsm_point16 *const *d_clouds,
                                 const int64_t *n_points, int n_pairs_total, rsm_point16 *d_out, int64_t max_out,
                                 int64_t *out_offsets) {
    if (!c || root < 0 || root >= c->world || n_local < 0 || n_pairs_total < 0 || n_pairs_total > RSM_COMM_MAX_PAIRS ||
        (n_local > 0 && (!pair_ids || !d_clouds || !n_points)))
        return RSM_E_INVALID;
    Rccl &R = rccl();
    HCHK(c, hipSetDevice(c->device));
    const int P = n_pairs_total;
    // 1. who holds which pair, and how many points: meta[p] = count, meta[P + p] = owner + 1; summed over the ranks
    std::vector<int64_t> meta((size_t)2 * P, 0);
    for (int i = 0; i < n_local; i++) {
        const int p = pair_ids[i];
        if (p < 0 || p >= P || n_points[i] < 0 || meta[(size_t)P + p] != 0) return comm_err(c, RSM_E_INVALID, "rsm_gather_clouds", "pair id");
        meta[(size_t)p] = n_points[i];
        meta[(size_t)P + p] = c->rank + 1;
    }
    if (P > 0) {
        HCHK(c, hipMemcpyAsync(c->d_meta, meta.data(), sizeof(int64_t) * 2 * P, hipMemcpyHostToDevice, c->stream));
        NCHK(c, R.AllReduce(c->d_meta, c->d_meta, (size_t)2 * P, ncclInt64, ncclSum, c->comm, c->stream));
        HCHK(c, hipMemcpyAsync(meta.data(), c->d_meta, sizeof(int64_t) * 2 * P, hipMemcpyDeviceToHost, c->stream));
        HCHK(c, hipStreamSynchronize(c->stream));
    }
    std::vector<int64_t> off((size_t)P + 1, 0);
    for (int p = 0; p < P; p++) {
        if (meta[(size_t)P + p] < 0 || meta[(size_t)P + p] > c->world) return comm_err(c, RSM_E_INVALID, "rsm_gather_clouds", "a pair is held by two ranks");
        off[(size_t)p + 1] = off[(size_t)p] + meta[(size_t)p];
    }
    // the root may not be able to take the clouds (capacity): every rank has to learn that BEFORE the payload group, or
    // the peers' sends would wait for receives that are never posted
    int64_t bad = (c->rank == root && (off[(size_t)P] > max_out || (off[(size_t)P] > 0 && !d_out))) ? 1 : 0;
    HCHK(c, hipMemcpyAsync(c->d_meta, &bad, sizeof bad, hipMemcpyHostToDevice, c->stream));
    NCHK(c, R.AllReduce(c->d_meta, c->d_meta, 1, ncclInt64, ncclMax, c->comm, c->stream));
    HCHK(c, hipMemcpyAsync(&bad, c->d_meta, sizeof bad, hipMemcpyDeviceToHost, c->stream));
    HCHK(c, hipStreamSynchronize(c->stream));
    if (bad) return comm_err(c, RSM_E_INVALID, "rsm_gather_clouds", "the root's output capacity is too small for the gathered clouds");
    if (c->rank == root && out_offsets) memcpy(out_offsets, off.data(), sizeof(int64_t) * ((size_t)P + 1));
    // 2. payload fan-in: one group, every peer's sends and the root's receives in flight together
    NCHK(c, R.GroupStart());
    if (c->rank == root) {
        for (int p = 0; p < P; p++) {
            const int owner = (int)meta[(size_t)P + p] - 1;
            if (owner < 0 || owner == root || meta[(size_t)p] == 0) continue;
            NCHK(c, R.Recv(d_out + off[(size_t)p], (size_t)meta[(size_t)p] * sizeof(rsm_point16), ncclUint8, owner, c->comm, c->stream));
        }
    } else {
        for (int i = 0; i < n_local; i++)
            if (n_points[i] > 0)
                NCHK(c, R.Send(d_clouds[i], (size_t)n_points[i] * sizeof(rsm_point16), ncclUint8, root, c->comm, c->stream));
    }
    NCHK(c, R.GroupEnd());
    if (c->rank == root)
        for (int i = 0; i < n_local; i++)
            if (n_points[i] > 0)
                HCHK(c, hipMemcpyAsync(d_out + off[(size_t)pair_ids[i]], d_clouds[i], (size_t)n_points[i] * sizeof(rsm_point16),
                                       hipMemcpyDeviceToDevice, c->stream));
    HCHK(c, hipStreamSynchronize(c->stream));
    return RSM_OK;
}
