"""The C++ glue of include/rsm_stereo_adapter.hpp, executed: tests/cpp/mock_adapter.cpp (mock CStereoMatching /
CManageData behind the accessor traits, no OpenCV) is built with g++ here on the GPU box, linked against
librsm_mi355.so, fed three pairs -- the middle one degenerate -- and its InsertPoint stream, bounds, disparity and
cloud%d.ply (isoutput) are compared with the ctypes path: once as a MatchPair call per pair, once as ONE MatchAll call (the
pair loop of CStereoMatching.cpp:17-33 with two / three pairs in flight, results replayed in pair order)."""
import os
import struct
import subprocess

import numpy as np
import pytest

from reconstruction_amd import _lib, synth, write_ply

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def build(tmp_path):
    exe = str(tmp_path / "mock_adapter")
    cmd = ["g++", "-std=c++11", "-O1", "-Wall", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "mock_adapter.cpp"),
           "-o", exe, "-L" + os.path.dirname(_lib.LIB_PATH), "-lrsm_mi355", "-Wl,-rpath," + os.path.dirname(_lib.LIB_PATH),
           "-Wl,-rpath-link,/opt/rocm/lib", "-Wl,--allow-shlib-undefined", "-pthread"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    return exe


def _write_input(tmp_path, cfgs, kw, bad):
    W, H = kw["width"], kw["height"]
    with open(tmp_path / "in.bin", "wb") as f:
        f.write(struct.pack("<9i", len(cfgs), W, H, kw["levels"], kw["radius"], kw["offset"], W, 1, bad))
        f.write(struct.pack("<d", 0.03))
        for c in cfgs:
            f.write(np.asarray(c.Q, np.float64).tobytes() + np.asarray(c.R_final, np.float64).tobytes() + np.asarray(c.T_final, np.float64).tobytes())
            for a in (c.image[0], c.image[1], c.mask[0], c.mask[1]):
                f.write(np.ascontiguousarray(a, np.uint8).tobytes())


# flags: 1 fp64 points over PCIe (round 4's form), 4 the device list {0, 0}, 8 a callback throws during a first MatchAll
@pytest.mark.parametrize("mode,flags", [(0, 0), (0, 1), (2, 0), (3, 1), (2, 4), (3, 8), (1, 12)])
def test_mock_pipeline_through_the_cpp_adapter(ctx, tmp_path, mode, flags):
    exe = build(tmp_path)
    kw = dict(width=192, height=128, levels=3, radius=2, offset=2, mask_l0_width=30, border_l0=4)
    cfgs = [synth.config_small(pair=p, **kw) for p in ((1, 2, 4) if mode == 0 else (1, 2, 4, 5, 3))]
    bad = 1
    W, H = kw["width"], kw["height"]
    _write_input(tmp_path, cfgs, kw, bad)
    env = dict(os.environ, LD_LIBRARY_PATH="/opt/rocm/lib:" + os.environ.get("LD_LIBRARY_PATH", ""))
    r = subprocess.run([exe, str(tmp_path / "in.bin"), str(tmp_path / "out.bin"), str(mode), str(flags)], capture_output=True, text=True,
                       cwd=tmp_path, env=env, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    buf = open(tmp_path / "out.bin", "rb").read()
    off = 0
    for p, c in enumerate(cfgs):
        ok, status = struct.unpack_from("<2i", buf, off); off += 8
        if p == bad:
            assert (ok, status) == (0, -2)      # RSM_E_DEGENERATE_MARGIN: reported, and the NEXT pair still runs
            assert ("level" in r.stderr) if mode == 0 else ("pair 1: status -2" in r.stderr)
            continue
        assert (ok, status) == (1, 0)
        mg = struct.unpack_from("<12i", buf, off); off += 48
        n, = struct.unpack_from("<q", buf, off); off += 8
        farg, = struct.unpack_from("<i", buf, off); off += 4
        xyz = np.frombuffer(buf, np.float64, 3 * n, off).reshape(n, 3); off += 24 * n
        d0 = np.frombuffer(buf, np.float64, W * H, off).reshape(H, W); off += 8 * W * H
        ref = ctx.match_pair(c)
        assert [tuple(mg[:6]), tuple(mg[6:])] == ref.margin
        assert n == ref.n_points > 1000 and farg == p          # one InsertPoint per point, then filter(CamPair)
        assert np.array_equal(d0, ref.disparity[0])
        # what a PCL-side InsertPoint stores -- pcl::PointXYZ(p[0], p[1], p[2]): the float cast, CCloudOptimization.cpp:61 --
        # is the same float stream whichever form crossed PCIe
        assert np.array_equal(xyz.astype(np.float32), ref.xyz.astype(np.float32), equal_nan=True)
        if flags & 1:   # fp64 points: InsertPoint receives the kernel's doubles
            assert np.array_equal(xyz, ref.xyz, equal_nan=True)
        else:           # 16-byte records: InsertPoint receives (double)float
            assert np.array_equal(xyz, ref.xyz.astype(np.float32).astype(np.float64), equal_nan=True)
        # isoutput: the in-call cloud%d.ply of DisparityToCloud (.cpp:707-757)
        write_ply(tmp_path / "want.ply", ref.xyz, ref.bgr)
        assert open(tmp_path / ("cloud%d.ply" % p), "rb").read() == open(tmp_path / "want.ply", "rb").read()
    assert off == len(buf)


@pytest.mark.parametrize("mode,flags", [(2, 2), (1, 6)])
def test_mock_pipeline_with_the_filter_on_the_gpu(ctx, tmp_path, mode, flags):
    """MatchAllFiltered: the per-pair cloud filter (CCloudOptimization.cpp:82-121) runs on the pair's GPU inside the loop; the
    pipeline receives the surviving points (16-byte records) and their oriented normals instead of InsertPoint x n +
    filter(pair).  Equal to Context.filter_last_cloud_host on the same pair, in pair order, a degenerate pair skipped."""
    exe = build(tmp_path)
    kw = dict(width=192, height=128, levels=3, radius=2, offset=2, mask_l0_width=30, border_l0=4)
    cfgs = [synth.config_small(pair=p, **kw) for p in (1, 2, 4, 5)]
    bad = 1
    W, H = kw["width"], kw["height"]
    _write_input(tmp_path, cfgs, kw, bad)
    env = dict(os.environ, LD_LIBRARY_PATH="/opt/rocm/lib:" + os.environ.get("LD_LIBRARY_PATH", ""))
    r = subprocess.run([exe, str(tmp_path / "in.bin"), str(tmp_path / "out.bin"), str(mode), str(flags)], capture_output=True, text=True,
                       cwd=tmp_path, env=env, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    buf = open(tmp_path / "out.bin", "rb").read()
    off = 0
    for p, c in enumerate(cfgs):
        ok, status = struct.unpack_from("<2i", buf, off); off += 8
        if p == bad:
            assert (ok, status) == (0, -2)
            continue
        assert (ok, status) == (1, 0)
        mg = struct.unpack_from("<12i", buf, off); off += 48
        n, = struct.unpack_from("<q", buf, off); off += 8
        farg, = struct.unpack_from("<i", buf, off); off += 4
        assert n == 0 and farg == p                           # no InsertPoint calls; filtered_cloud(pair) once
        n_kept, n_raw = struct.unpack_from("<2q", buf, off); off += 16
        kept = np.frombuffer(buf, ctx.POINT16, n_kept, off); off += 16 * n_kept
        nrm = np.frombuffer(buf, np.float32, 4 * n_kept, off).reshape(n_kept, 4); off += 16 * n_kept
        d0 = np.frombuffer(buf, np.float64, W * H, off).reshape(H, W); off += 8 * W * H
        ref = ctx.match_pair(c)
        assert [tuple(mg[:6]), tuple(mg[6:])] == ref.margin and n_raw == ref.n_points and np.array_equal(d0, ref.disparity[0])
        rec, rn, st = ctx.filter_last_cloud_host(100, 1.0, 2.5, (10.0 * p, -5.0, 3.0))
        assert 0 < n_kept == len(rec) <= n_raw
        assert kept.tobytes() == rec.tobytes() and np.array_equal(nrm, rn, equal_nan=True)
    assert off == len(buf)


def test_points16_download_is_the_float_cast_of_the_cloud(ctx):
    """rsm_pair_out.points16: the cloud packed on the GPU as InsertPoint keeps it (float xyz, CCloudOptimization.cpp:61) + BGR."""
    cfg = synth.config_small(width=192, height=128, levels=3, radius=2, offset=2, pair=3, mask_l0_width=30, border_l0=4)
    ref = ctx.match_pair(cfg)
    rec = ctx.download_points16()
    assert len(rec) == ref.n_points > 1000
    got = np.stack([rec["x"], rec["y"], rec["z"]], 1)
    assert np.array_equal(got, ref.xyz.astype(np.float32), equal_nan=True)
    assert np.array_equal(np.stack([rec["b"], rec["g"], rec["r"]], 1), ref.bgr) and not rec["pad"].any()
