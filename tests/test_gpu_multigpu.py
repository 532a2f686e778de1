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


def test_bench_gathers_through_the_librarys_own_rccl_path():
    """The N > 1 control flow of bench.py with its DEFAULT transport -- rsm_comm_create + rsm_gather_clouds, the product's own
    RCCL path (csrc/rsm_comm.hip), posted by a gather thread while the next step is matched -- on a one-GPU box: at N = 1
    RSM_BENCH_SELF_GATHER=1 keeps the packing, the hand-over and the gather (a one-rank communicator: RCCL refuses two ranks on
    one device).  The line names the transport; the driver's scaling run times exactly this path."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "RSM_BENCH_BACKEND")}
    env["RSM_BENCH_SELF_GATHER"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--config", "c2s", "--no-cpu-baseline",
                        "--measure-traffic", "0", "--adapter-pairs", "0"], capture_output=True, text=True, env=env, cwd=ROOT, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 1 and d["value"] > 0 and d["config"]["transport"] == "rccl"


def test_bench_one_process_mode_drives_the_gpus_through_the_c_abi():
    """`bench.py --gpus N --one-process` = rsm_match_pairs_multi_gpu (N = 1 here): an auxiliary, PCIe-inclusive figure."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "RSM_BENCH_BACKEND")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--one-process", "--steps", "2", "--inflight", "2", "--config", "c2s"],
                       capture_output=True, text=True, env=env, cwd=ROOT, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 1 and d["value"] > 0 and d["config"]["pairs"] == 4 and "rsm_match_pairs_multi_gpu" in d["config"]["parallelism"]


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


def test_match_pairs_multi_gpu_from_one_process(ctx):
    """rsm_match_pairs_multi_gpu: the pair loop sharded over the node's GPUs from ONE process, the library
    owning its contexts.  n_gpus = 0 (every visible GPU) and 1, two pairs in flight per GPU, a degenerate pair in the
    middle (the reference's exit(0), CStereoMatching.cpp:827-830): every other pair equals rsm_match_pair bit for bit."""
    from reconstruction_amd import match_pairs_multi_gpu
    cfgs = [synth.config_small(**kw) for kw in PAIRS] + [synth.config_small(**PAIRS[0])]
    bad = synth.config_small(**PAIRS[1])
    bad.mask = [np.zeros_like(bad.mask[0]), np.zeros_like(bad.mask[1])]
    cfgs.insert(2, bad)
    want = [None if i == 2 else ctx.match_pair(c) for i, c in enumerate(cfgs)]
    for n_gpus in (0, 1):
        res, status, rc = match_pairs_multi_gpu(cfgs, n_gpus=n_gpus, pairs_in_flight=2)
        assert status == [0, 0, -2, 0, 0] and rc == -2      # RSM_E_DEGENERATE_MARGIN reported, the others still ran
        assert res[2] is None
        for r, w in zip(res, want):
            if w is None:
                continue
            assert r.margin == w.margin and r.n_points == w.n_points > 0 and r.v_top == w.v_top
            assert all(np.array_equal(r.disparity[v], w.disparity[v]) for v in range(2))
            assert np.array_equal(r.xyz, w.xyz, equal_nan=True) and np.array_equal(r.bgr, w.bgr)
    # no pairs, and bad arguments
    assert match_pairs_multi_gpu([], n_gpus=1)[2] == 0
    from reconstruction_amd import _lib
    assert _lib.load().rsm_match_pairs_multi_gpu(None, 1, 1, 1, None, None) != 0
    assert _lib.load().rsm_match_pairs_multi_gpu(None, 0, -1, 1, None, None) != 0


def _rccl_worker(rank, world, uid, owned, q):
    """one process per GPU: match the owned pairs on device `rank`, pack them, fan them in to rank 0 over RCCL"""
    sys.path.insert(0, ROOT)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(rank)
    from reconstruction_amd import Context
    from reconstruction_amd.dist import Comm, unpack_records
    comm = Comm(uid, rank, world, rank)
    try:
        local = []
        with Context(rank) as ctx:
            for pid, kw in owned:
                if kw is None:                 # an owned pair whose cloud is empty
                    local.append((pid, torch.empty((0, 16), dtype=torch.uint8, device="cuda:%d" % rank)))
                    continue
                res = ctx.match_pair(synth.config_small(**kw))
                rec = torch.empty((res.n_points, 16), dtype=torch.uint8, device="cuda:%d" % rank)
                assert ctx.pack_cloud16(rec.data_ptr(), res.n_points) == res.n_points
                local.append((pid, rec))
            out = comm.gather(local, n_pairs_total=5, root=0)
            if rank == 0:
                q.put([(pid,) + unpack_records(r.cpu()) for pid, r in out])
            else:
                assert out is None
    finally:
        comm.close()


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs: RCCL refuses two ranks on one device")
def test_real_rccl_fan_in_between_two_gpus(ctx):
    """The C ABI's own transport for real: two processes, one GPU each, rsm_comm_create + rsm_gather_clouds (ncclSend /
    ncclRecv group over xGMI).  Pair ids listed out of order, a pair nobody owns, an owned pair with an empty cloud:
    rank 0 receives every cloud in pair order, equal to the single-process clouds."""
    from reconstruction_amd.dist import Comm
    owned = {0: [(3, PAIRS[0]), (0, PAIRS[2])], 1: [(4, None), (2, PAIRS[1])]}   # pair 1 belongs to nobody
    want = {pid: ctx.match_pair(synth.config_small(**kw)) for r in owned for pid, kw in owned[r] if kw is not None}
    uid = Comm.unique_id()
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    procs = [mpc.Process(target=_rccl_worker, args=(r, 2, uid, owned[r], q)) for r in range(2)]
    for p in procs:
        p.start()
    res = q.get(timeout=300)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert [pid for pid, _, _ in res] == [0, 1, 2, 3, 4]
    for pid, xyz, bgr in res:
        if pid in want:
            assert len(xyz) == want[pid].n_points > 0
            assert np.array_equal(xyz, want[pid].xyz.astype(np.float32), equal_nan=True) and np.array_equal(bgr, want[pid].bgr)
        else:
            assert len(xyz) == 0


def test_bench_rig_config_shards_ten_pairs_strong_scaling():
    """bench.py --config c4s (BASELINE configs[3] at the shipped scale): the rig's ten pairs, pair % N -> rank, every step all
    ten pairs + the gather of the ten clouds; strong scaling, so N = 1 and N = 2 (gloo stand-in) report the same V_top total."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    out = {}
    for n, extra in ((1, {}), (2, {"RSM_BENCH_BACKEND": "gloo", "MASTER_ADDR": "127.0.0.1"})):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "1", "--warmup", "1", "--config", "c4s",
                            "--no-cpu-baseline", "--measure-traffic", "0"], capture_output=True, text=True, env=dict(env, **extra), cwd=ROOT, timeout=900)
        assert r.returncode == 0, r.stderr[-2000:]
        lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
        assert len(lines) == 1
        out[n] = json.loads(lines[0])
    d1, d2 = out[1], out[2]
    assert d1["scaling"] == d2["scaling"] == "strong" and d1["config"]["pairs"] == 10
    assert d1["config"]["pairs_per_rank"] == [10] and d2["config"]["pairs_per_rank"] == [5, 5] and d2["n_gpus"] == 2
    # value = sum of V_top over the ten pairs / step time: the same numerator at both N
    v1 = d1["value"] * d1["ms_per_step"]
    v2 = d2["value"] * d2["ms_per_step"]
    assert abs(v1 - v2) <= 1e-3 * v1 and d1["value"] > 0
