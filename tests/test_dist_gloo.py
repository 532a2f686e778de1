"""Multi-rank path on CPU: pairs shard statically across ranks, clouds are gathered to rank 0 in pair order
(world_size 2 and 3, gloo).  Same code path bench.py / a multi-GPU driver uses with backend nccl (= RCCL)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _cloud(pair):
    rng = np.random.default_rng(pair)
    n = 5 + 3 * pair if pair != 2 else 0          # pair 2 has an empty cloud
    return rng.normal(size=(n, 3)), rng.integers(0, 256, (n, 3)).astype(np.uint8)


def _worker(rank, world, port, n_pairs, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from reconstruction_amd.dist import gather_clouds, pack_records, shard_pairs, unpack_records
    mine = shard_pairs(n_pairs, world, rank)
    local = []
    for p in mine:
        xyz, bgr = _cloud(p)
        local.append((p, pack_records(xyz, bgr)))
    res = gather_clouds(local, dst=0)
    if rank == 0:
        q.put([(pid,) + unpack_records(r) for pid, r in res])
    else:
        assert res is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n_pairs", [(2, 5), (3, 4), (2, 1)])
def test_shard_and_gather(world, n_pairs):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_pairs, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [pid for pid, _, _ in res] == list(range(n_pairs))   # pair order, independent of the sharding
    for pid, xyz, bgr in res:
        ex, eb = _cloud(pid)
        assert xyz.dtype == np.float32 and np.array_equal(xyz, ex.astype(np.float32)) and np.array_equal(bgr, eb)


def test_shard_pairs_is_a_partition():
    from reconstruction_amd.dist import shard_pairs
    for world in (1, 2, 3, 8):
        for n in (0, 1, 10, 16):
            got = sorted(p for r in range(world) for p in shard_pairs(n, world, r))
            assert got == list(range(n))
    # the 10-camera rig on 8 GPUs: 2,2,1,1,1,1,1,1 pairs
    assert [len(shard_pairs(10, 8, r)) for r in range(8)] == [2, 2, 1, 1, 1, 1, 1, 1]


def test_single_process_gather_is_identity():
    from reconstruction_amd.dist import gather_clouds, pack_records, unpack_records
    loc = [(3, pack_records(np.zeros((2, 3)), np.zeros((2, 3), np.uint8))),
           (1, pack_records(np.ones((1, 3)), np.ones((1, 3), np.uint8)))]
    out = gather_clouds(loc)
    assert [p for p, _ in out] == [1, 3]
    xyz, bgr = unpack_records(out[0][1])
    assert xyz.tolist() == [[1.0, 1.0, 1.0]] and bgr.tolist() == [[1, 1, 1]] and out[0][1].shape == (1, 16)


def _worker_async(rank, world, port, rounds, q):
    """bench.py's pattern: the gather of round i is left in flight while round i + 1 is prepared."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from reconstruction_amd.dist import gather_clouds_async, pack_records, unpack_records
    got, pending = [], None
    for i in range(rounds):
        pair = i * world + rank
        xyz, bgr = _cloud(pair)
        h = gather_clouds_async([(pair, pack_records(xyz, bgr))], dst=0)
        if pending is not None:
            got.append(pending.wait())
        pending = h
    got.append(pending.wait())
    if rank == 0:
        q.put([[(pid,) + unpack_records(r) for pid, r in res] for res in got])
    else:
        assert all(r is None for r in got)
    dist.barrier()
    dist.destroy_process_group()


def test_pipelined_gathers_stay_in_order():
    world, rounds = 2, 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_async, args=(r, world, port, rounds, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert len(res) == rounds
    for i, rnd in enumerate(res):
        assert [pid for pid, _, _ in rnd] == [i * world + r for r in range(world)]
        for pid, xyz, bgr in rnd:
            ex, eb = _cloud(pid)
            assert np.array_equal(xyz, ex.astype(np.float32)) and np.array_equal(bgr, eb)
