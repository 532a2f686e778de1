"""Multi-GPU layer: independent stereo pairs shard across ranks (one process per GPU), the only
exchange is the final gather of per-pair clouds to rank 0 -- the data-parallel replacement of the
reference's process-global cloud accumulation (CloudOptimization/CCloudOptimization.cpp:61,123) fed
by the sequential pair loop of CStereoMatching::MatchAllLayer (.cpp:17).

torch.distributed is plumbing only: backend "nccl" (= RCCL over xGMI) on GPUs, "gloo" in CPU tests.
The gather is a fan-in (every peer sends straight to rank 0, rank 0 posts all receives in one
batch = one ncclGroup), not a ring: xGMI is point-to-point, so rank 0's seven inbound links carry
the seven peers concurrently and nothing is forwarded twice.  It can be started asynchronously
(`gather_clouds_async`) so that a pair's cloud travels while the next pair is being matched.
"""
from __future__ import annotations

import torch
import torch.distributed as dist

META_PAIRS = 64  # pairs per rank and call that fit the fixed-size metadata message


def shard_pairs(n_pairs: int, world: int, rank: int) -> list:
    """Static block-cyclic assignment: pair p runs on rank p % world."""
    return list(range(rank, n_pairs, world))


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
    """Starts the fan-in of `local` = [(pair_id, xyz[n,3] float64, bgr[n,3] uint8), ...] (tensors on the rank's
    device for nccl, CPU for gloo) to rank `dst` and returns a handle.  Only the tiny metadata message (pair ids
    and point counts, which size the receive buffers) is waited for here; the payload is still in flight."""
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
        metas = [torch.zeros(mlen, dtype=torch.int64, device=dev) for _ in range(world)]
        for w in _batch([dist.P2POp(dist.irecv, metas[src], src, group) for src in range(world) if src != dst]):
            w.wait()
        ops, out = [], []
        for src in range(world):
            if src == dst:
                out += list(local)
                continue
            m = metas[src].cpu().tolist()
            for k in range(m[0]):
                pid, n = m[1 + 2 * k], m[2 + 2 * k]
                xyz = torch.empty((n, 3), dtype=torch.float64, device=dev)
                bgr = torch.empty((n, 3), dtype=torch.uint8, device=dev)
                out.append((pid, xyz, bgr))
                if n > 0:
                    ops.append(dist.P2POp(dist.irecv, xyz, src, group))
                    ops.append(dist.P2POp(dist.irecv, bgr, src, group))
        # 2) payload fan-in, left in flight
        return CloudGather(_batch(ops), out, [], True)
    flat = [len(local)]
    for pid, xyz, _ in local:
        flat += [int(pid), int(xyz.shape[0])]
    meta = torch.tensor(flat + [0] * (mlen - len(flat)), dtype=torch.int64, device=dev)
    works = _batch([dist.P2POp(dist.isend, meta, dst, group)])
    ops, keep = [], [meta]
    for pid, xyz, bgr in local:
        if xyz.shape[0] > 0:
            xyz = xyz.contiguous()
            bgr = bgr.contiguous()
            keep += [xyz, bgr]
            ops.append(dist.P2POp(dist.isend, xyz, dst, group))
            ops.append(dist.P2POp(dist.isend, bgr, dst, group))
    return CloudGather(works + _batch(ops), [], keep, False)


def gather_clouds(local: list, dst: int = 0, group=None):
    """local: list of (pair_id, xyz[n,3] float64 tensor, bgr[n,3] uint8 tensor) held by this rank.
    Returns on `dst` a list of (pair_id, xyz, bgr) for ALL pairs ordered by pair_id, else None."""
    return gather_clouds_async(local, dst, group).wait()
