"""Multi-GPU layer: independent stereo pairs shard across ranks (one process per GPU), the only
exchange is the final gather of per-pair clouds to rank 0 -- the data-parallel replacement of the
reference's process-global cloud accumulation (CloudOptimization/CCloudOptimization.cpp:61,123) fed
by the sequential pair loop of CStereoMatching::MatchAllLayer (.cpp:17).

What travels is the 16-byte point record of include/rsm.h (`rsm_point16`: float x, y, z -- InsertPoint
casts the fp64 point to float, CCloudOptimization.cpp:61 -- + b, g, r + pad), packed on the GPU by
`rsm_pack_cloud16`.  Two transports with the same protocol (counts first, then a fan-in of the payload):
  * `rsm_gather_clouds` of the C ABI, straight on RCCL (`Comm` below) -- what a C++ pipeline links;
  * torch.distributed (this module): backend "nccl" (= RCCL over xGMI) on GPUs, "gloo" in CPU tests.
The gather is a fan-in (every peer sends straight to rank 0, rank 0 posts all receives in one batch = one
ncclGroup), not a ring: xGMI is point-to-point, so rank 0's seven inbound links carry the seven peers
concurrently and nothing is forwarded twice.  It can be started asynchronously (`gather_clouds_async`) so that
a pair's cloud travels while the next pair is being matched.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch
import torch.distributed as dist

META_PAIRS = 64  # pairs per rank and call that fit the fixed-size metadata message
REC = 16         # bytes per point record (rsm_point16)


def shard_pairs(n_pairs: int, world: int, rank: int) -> list:
    """Static block-cyclic assignment: pair p runs on rank p % world."""
    return list(range(rank, n_pairs, world))


def pack_records(xyz, bgr) -> torch.Tensor:
    """Host-side twin of rsm_pack_cloud16 (tests, CPU transports): (n,3) fp64/fp32 xyz + (n,3) uint8 bgr ->
    uint8 [n, 16] records."""
    xyz = np.asarray(xyz)
    n = len(xyz)
    rec = np.zeros((n, REC), np.uint8)
    rec[:, :12] = np.ascontiguousarray(xyz.astype(np.float32)).view(np.uint8).reshape(n, 12)
    rec[:, 12:15] = np.asarray(bgr, np.uint8).reshape(n, 3)
    return torch.from_numpy(rec)


def unpack_records(rec):
    """uint8 [n, 16] records -> (xyz float32 [n,3], bgr uint8 [n,3]) numpy arrays."""
    a = rec.detach().cpu().numpy() if isinstance(rec, torch.Tensor) else np.asarray(rec)
    a = np.ascontiguousarray(a).reshape(-1, REC)
    return a[:, :12].copy().view(np.float32).reshape(-1, 3), a[:, 12:15].copy()


class CloudGather:
    """Handle of a gather in flight: `wait()` returns what `gather_clouds` returns."""

    def __init__(self, works, out, keep, is_dst):
        self._works, self._out, self._keep, self._is_dst = works, out, keep, is_dst

    def wait(self):
        for w in self._works:
            w.wait()
        self._works, self._keep = [], []
        return sorted(self._out, key=lambda t: t[0]) if self._is_dst else None


def _batch(ops):
    return dist.batch_isend_irecv(ops) if ops else []


def gather_clouds_async(local: list, dst: int = 0, group=None) -> CloudGather:
    """Starts the fan-in of `local` = [(pair_id, records uint8 [n,16]), ...] (tensors on the rank's device for nccl,
    CPU for gloo) to rank `dst` and returns a handle.  Only the fixed-size count message (8 B per pair: it sizes
    the receive buffers, so the host has to see it) is waited for here; the payload is still in flight."""
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return CloudGather([], list(local), [], True)
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    if len(local) > META_PAIRS:
        raise ValueError("at most %d pairs per rank and gather call" % META_PAIRS)
    dev = local[0][1].device if local else (torch.device("cuda", torch.cuda.current_device())
                                            if dist.get_backend(group) == "nccl" else torch.device("cpu"))
    # 1) metadata: [n_pairs, pid0, n0, pid1, n1, ...] from every peer to dst (fixed size, so it can be posted blind)
    mlen = 1 + 2 * META_PAIRS
    if rank == dst:
        metas = torch.zeros((world, mlen), dtype=torch.int64, device=dev)
        for w in _batch([dist.P2POp(dist.irecv, metas[src], src, group) for src in range(world) if src != dst]):
            w.wait()
        m_all = metas.tolist()  # one device -> host copy for all peers
        ops, out = [], []
        for src in range(world):
            if src == dst:
                out += list(local)
                continue
            m = m_all[src]
            for k in range(m[0]):
                pid, n = m[1 + 2 * k], m[2 + 2 * k]
                rec = torch.empty((n, REC), dtype=torch.uint8, device=dev)
                out.append((pid, rec))
                if n > 0:
                    ops.append(dist.P2POp(dist.irecv, rec, src, group))
        # 2) payload fan-in, left in flight
        return CloudGather(_batch(ops), out, [], True)
    flat = [len(local)]
    for pid, rec in local:
        flat += [int(pid), int(rec.shape[0])]
    meta = torch.tensor(flat + [0] * (mlen - len(flat)), dtype=torch.int64, device=dev)
    works = _batch([dist.P2POp(dist.isend, meta, dst, group)])
    ops, keep = [], [meta]
    for pid, rec in local:
        if rec.shape[0] > 0:
            rec = rec.contiguous()
            keep.append(rec)
            ops.append(dist.P2POp(dist.isend, rec, dst, group))
    return CloudGather(works + _batch(ops), [], keep, False)


def gather_clouds(local: list, dst: int = 0, group=None):
    """local: list of (pair_id, records uint8 [n,16]) held by this rank.
    Returns on `dst` a list of (pair_id, records) for ALL pairs ordered by pair_id, else None."""
    return gather_clouds_async(local, dst, group).wait()


class Comm:
    """rsm_comm of the C ABI: RCCL communicator + rsm_gather_clouds, without torch.distributed in the data path.
    The 128-byte id made by rank 0 (`Comm.unique_id()`) reaches the other ranks by any side channel."""

    def __init__(self, unique_id: bytes, rank: int, world: int, device: int):
        from . import _lib
        self._lib = _lib.load()
        h = C.c_void_p()
        st = self._lib.rsm_comm_create(C.byref(h), C.c_char_p(bytes(unique_id)), rank, world, device)
        if st != 0:
            raise _lib.RsmError(st, "rsm_comm_create (is librccl loadable?)")
        self._h, self.rank, self.world = h, rank, world

    @staticmethod
    def unique_id() -> bytes:
        from . import _lib
        buf = C.create_string_buffer(128)
        st = _lib.load().rsm_comm_unique_id(buf)
        if st != 0:
            raise _lib.RsmError(st, "rsm_comm_unique_id (is librccl loadable?)")
        return buf.raw

    def counts(self, local, n_pairs_total):
        """Point count of every pair, on every rank (rsm_gather_counts: one all-reduce)."""
        from . import _lib
        n = len(local)
        ids = (C.c_int * max(n, 1))(*[int(p) for p, _ in local])
        cnts = (C.c_int64 * max(n, 1))(*[int(r.shape[0]) for _, r in local])
        out = (C.c_int64 * max(n_pairs_total, 1))()
        st = self._lib.rsm_gather_counts(self._h, n, ids, cnts, n_pairs_total, out)
        if st != 0:
            raise _lib.RsmError(st, (self._lib.rsm_comm_last_error(self._h) or b"").decode())
        return list(out[:n_pairs_total])

    def gather(self, local, n_pairs_total, root=0, capacity=None):
        """local: [(pair_id, records uint8 [n,16] CUDA tensor)] in any order.  Root returns [(pair_id, records view)]
        for all pairs.  capacity = records the root's buffer holds (only the root's value matters: it travels to the
        peers inside rsm_gather_clouds' own all-reduce); None sizes it exactly from the counts.  The counts-only
        exchange is posted by EVERY rank on EVERY call, whatever its `capacity` argument: ranks that disagreed about
        it would otherwise post different collective sequences and hang."""
        from . import _lib
        n = len(local)
        total = sum(self.counts(local, n_pairs_total))
        if capacity is None:
            capacity = total
        ids = (C.c_int * max(n, 1))(*[int(p) for p, _ in local])
        ptrs = (C.c_void_p * max(n, 1))(*[int(r.data_ptr()) if r.shape[0] else None for _, r in local])
        cnts = (C.c_int64 * max(n, 1))(*[int(r.shape[0]) for _, r in local])
        offs = (C.c_int64 * (n_pairs_total + 1))()
        out = None
        cap = 0
        if self.rank == root:
            cap = int(capacity)
            out = torch.empty((max(cap, 1), REC), dtype=torch.uint8, device=local[0][1].device if local else "cuda")
        st = self._lib.rsm_gather_clouds(self._h, root, n, ids, ptrs, cnts, n_pairs_total,
                                         C.c_void_p(out.data_ptr()) if out is not None else None, C.c_int64(cap), offs)
        if st != 0:
            raise _lib.RsmError(st, (self._lib.rsm_comm_last_error(self._h) or b"").decode())
        if self.rank != root:
            return None
        return [(p, out[offs[p]:offs[p + 1]]) for p in range(n_pairs_total)]

    def close(self):
        if getattr(self, "_h", None):
            self._lib.rsm_comm_destroy(self._h)
            self._h = None
