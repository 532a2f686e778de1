"""Multi-GPU layer: independent stereo pairs shard across ranks (one process per GPU), the only
exchange is the final gather of per-pair clouds to rank 0 -- the data-parallel replacement of the
reference's process-global cloud accumulation (CloudOptimization/CCloudOptimization.cpp:61,123) fed
by the sequential pair loop of CStereoMatching::MatchAllLayer (.cpp:17).

torch.distributed is plumbing only: backend "nccl" (= RCCL over xGMI) on GPUs, "gloo" in CPU tests.
The gather is a fan-in (every peer sends straight to rank 0, rank 0 posts all receives in one
batch = one ncclGroup), not a ring: xGMI is point-to-point, so rank 0's seven inbound links carry
the seven peers concurrently and nothing is forwarded twice.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def shard_pairs(n_pairs: int, world: int, rank: int) -> list:
    """Static block-cyclic assignment: pair p runs on rank p % world."""
    return list(range(rank, n_pairs, world))


def gather_clouds(local: list, dst: int = 0, group=None):
    """local: list of (pair_id, xyz[n,3] float64 tensor, bgr[n,3] uint8 tensor) held by this rank
    (tensors on the rank's device for nccl, CPU for gloo).
    Returns on `dst` a list of (pair_id, xyz, bgr) for ALL pairs ordered by pair_id, else None."""
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return sorted(local, key=lambda t: t[0])
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    dev = local[0][1].device if local else (torch.device("cuda", torch.cuda.current_device())
                                            if dist.get_backend(group) == "nccl" else torch.device("cpu"))
    # 1) tiny metadata exchange: (pair_id, n_points) of every pair on every rank
    meta = [(int(pid), int(xyz.shape[0])) for pid, xyz, _ in local]
    all_meta = [None] * world
    dist.all_gather_object(all_meta, meta, group=group)
    # 2) payload fan-in to dst
    ops, keep = [], []
    out = []
    if rank == dst:
        for src in range(world):
            for k, (pid, n) in enumerate(all_meta[src]):
                if src == dst:
                    out.append(local[k])
                    continue
                xyz = torch.empty((n, 3), dtype=torch.float64, device=dev)
                bgr = torch.empty((n, 3), dtype=torch.uint8, device=dev)
                out.append((pid, xyz, bgr))
                if n > 0:
                    ops.append(dist.P2POp(dist.irecv, xyz, src, group))
                    ops.append(dist.P2POp(dist.irecv, bgr, src, group))
    else:
        for pid, xyz, bgr in local:
            if xyz.shape[0] > 0:
                xyz = xyz.contiguous()
                bgr = bgr.contiguous()
                keep += [xyz, bgr]
                ops.append(dist.P2POp(dist.isend, xyz, dst, group))
                ops.append(dist.P2POp(dist.isend, bgr, dst, group))
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()
    if rank != dst:
        return None
    return sorted(out, key=lambda t: t[0])
