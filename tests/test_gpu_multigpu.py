"""The multi-GPU path with what one GPU box offers:
  * two gloo ranks sharing device 0 shard three pairs through Context, pack 16-byte point records on the GPU and
    fan them in to rank 0 (reconstruction_amd.dist) -- equal to the single-process clouds, in pair order;
  * the C ABI's RCCL transport (rsm_comm_*, rsm_gather_clouds) on a one-rank communicator (RCCL refuses two ranks on
    one device): id, init, the count all-reduce, the grouped exchange, pair-ordered output;
  * bench.py --gpus 2 through torch.distributed.run on the gloo stand-in backend."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from reconstruction_amd import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

PAIRS = [dict(width=160, height=96, levels=3, radius=2, pair=1, holes=True),
         dict(width=256, height=128, levels=3, radius=2, pair=5, occlude=True, mask_l0_width=48, border_l0=5),
         dict(width=160, height=96, levels=2, radius=5, offset=4, pair=2, border_l0=7)]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from reconstruction_amd import Context
    from reconstruction_amd.dist import gather_clouds, shard_pairs, unpack_records
    local = []
    with Context(0) as ctx:
        for p in shard_pairs(len(PAIRS), world, rank):
            ctx.upload_pair(synth.config_small(**PAIRS[p]))
            ctx.run_pair()
            n = ctx.n_points
            rec = torch.empty((n, 16), dtype=torch.uint8, device="cuda:0")
            assert ctx.pack_cloud16(rec.data_ptr(), n) == n
            local.append((p, rec.cpu()))        # gloo: through host memory (nccl would send the device tensor)
    res = gather_clouds(local, dst=0)
    if rank == 0:
        q.put([(pid,) + unpack_records(r) for pid, r in res])
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_shard_pairs_and_gather_like_one_process(ctx):
    want = [ctx.match_pair(synth.config_small(**kw)) for kw in PAIRS]
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    port = _free_port()
    procs = [mpc.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = q.get(timeout=300)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert [pid for pid, _, _ in res] == [0, 1, 2]
    for (pid, xyz, bgr), w in zip(res, want):
        assert len(xyz) == w.n_points > 0
        assert np.array_equal(xyz, w.xyz.astype(np.float32), equal_nan=True)   # InsertPoint's cast (CCloudOptimization.cpp:61)
        assert np.array_equal(bgr, w.bgr)


def test_rccl_gather_of_the_c_abi_on_one_rank(ctx):
    from reconstruction_amd.dist import Comm, pack_records, unpack_records
    comm = Comm(Comm.unique_id(), 0, 1, 0)
    try:
        local, want = [], {}
        for pid, kw in ((2, PAIRS[0]), (0, PAIRS[2])):      # pair 1 belongs to nobody: an empty slot in the output
            res = ctx.match_pair(synth.config_small(**kw))
            rec = torch.empty((res.n_points, 16), dtype=torch.uint8, device="cuda:0")
            assert ctx.pack_cloud16(rec.data_ptr(), res.n_points) == res.n_points
            assert torch.equal(rec.cpu(), pack_records(res.xyz, res.bgr))   # device packing = the host twin
            local.append((pid, rec))
            want[pid] = res
        out = comm.gather(local, n_pairs_total=3, root=0)
        assert [p for p, _ in out] == [0, 1, 2] and out[1][1].shape[0] == 0
        for pid in (0, 2):
            xyz, bgr = unpack_records(out[pid][1])
            assert np.array_equal(xyz, want[pid].xyz.astype(np.float32), equal_nan=True) and np.array_equal(bgr, want[pid].bgr)
    finally:
        comm.close()


@pytest.mark.parametrize("launcher", ["self", "torchrun"])
def test_bench_runs_with_two_ranks_on_the_gloo_stand_in(launcher):
    """`python bench.py --gpus 2` started plainly launches its two ranks itself (the driver's command); under
    torch.distributed.run it takes the ranks it is given.  Either way the line says n_gpus = 2."""
    env = dict(os.environ, RSM_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    tail = [os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--config", "c2s"]
    if launcher == "self":
        cmd = [sys.executable] + tail
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
               "--master-port", str(_free_port())] + tail
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, cwd=ROOT, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                      # ONE json line, from rank 0
    d = json.loads(lines[-1])
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["scaling"] == "weak" and "roofline" in d
    assert d["config"]["pairs_per_gpu"] == 3  # bench.py's default --inflight


def test_bench_refuses_to_measure_fewer_gpus_than_asked():
    """One visible GPU, --gpus 2 on the real (nccl) backend: exit non-zero, never a silent 1-GPU number."""
    if torch.cuda.device_count() >= 2:
        pytest.skip("this box has two GPUs")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "RSM_BENCH_BACKEND")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, env=env, cwd=ROOT, timeout=300)
    assert r.returncode != 0 and "needs 2 visible GPUs" in r.stderr and not [l for l in r.stdout.splitlines() if l.startswith("{")]
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, env=dict(env, WORLD_SIZE="1", RANK="0"), cwd=ROOT, timeout=300)
    assert r.returncode != 0 and "refusing" in r.stderr
