"""Rectify on the GPU against the CPU oracle: fixed-point maps, bilinear remap and grey erosion are
integer arithmetic -> bit-exact; the fp64 plan (stereoRectify etc.) is host code on both sides -> identical."""
import numpy as np
import pytest

from oracle import oracle as orc
from reconstruction_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def raw():
    return synth.make_raw_pair()


def test_rect_maps_bit_exact(ctx, raw):
    K, E = raw["K"], raw["E"]
    R = E[1][:, :3] @ E[0][:, :3].T
    T = -R @ E[0][:, 3] + E[1][:, 3]
    R1, R2, P1, P2, Q = orc.stereo_rectify(K[0], K[1], raw["origin"], R, T)
    for Kv, Rv, Pv in ((K[0], R1, P1), (K[1], R2, P2)):
        newA = Pv[:, :3].copy()
        newA[:2] *= 0.5
        for (W, H) in ((320, 240), (333, 201)):
            g1, g2 = ctx.rect_map(Kv, Rv, newA, W, H)
            o1, o2 = orc.init_rectify_map(Kv, Rv, newA, W, H)
            assert np.array_equal(g1, o1) and np.array_equal(g2, o2)


def test_remap_bit_exact(ctx, raw):
    rng = np.random.default_rng(3)
    H, W = 150, 210
    m1 = np.zeros((H, W, 2), np.int16)
    m1[..., 0] = rng.integers(-5, raw["origin"][0] + 5, (H, W))
    m1[..., 1] = rng.integers(-5, raw["origin"][1] + 5, (H, W))
    m2 = rng.integers(0, 1024, (H, W)).astype(np.uint16)
    for src in (raw["image"][0], raw["mask"][1], rng.integers(0, 256, raw["mask"][0].shape).astype(np.uint8)):
        assert np.array_equal(ctx.remap_linear(src, m1, m2), orc.remap_linear(src, m1, m2))


@pytest.mark.parametrize("ksize", [1, 2, 3, 6, 12, 24, 48, 61])
def test_grey_erosion_bit_exact(ctx, ksize):
    rng = np.random.default_rng(ksize)
    m = rng.integers(0, 256, (97, 131)).astype(np.uint8)
    m[20:70, 30:110] = 255
    m[40:44, 60:66] = rng.integers(0, 255, (4, 6))
    assert np.array_equal(ctx.erode_ellipse_gray(m, ksize), orc.erode_ellipse(m, ksize))


def test_rectify_pair_plan_satisfies_the_geometry(ctx, raw):
    """What rsm_rectify_pair leaves behind (Q, R_final, T_final, cam[j].P: CStereoMatching.cpp:132-145) held to the geometry
    itself, not to the oracle's copy of the same transcription (tests/test_rectify_properties.py does it for random poses on
    the host routine): world points projected with cam[0].P / cam[1].P land on one row, and DisparityToCloud's formula with
    the reference's disparity convention followed by R_final X + T_final returns them to 1e-9."""
    g = ctx.rectify_pair(raw["K"], raw["E"], raw["origin"], raw["lowest"], raw["pyr_levels"], raw["image"], raw["mask"], want_images=False)
    scale = float(raw["lowest"][0]) / raw["origin"][0] * (1 << (raw["pyr_levels"] - 1))
    q = np.array(g["Q"], np.float64)
    q[:, 3] *= scale                                        # :697-699
    assert np.abs(g["R_final"] @ g["R_final"].T - np.eye(3)).max() < 1e-13
    rng = np.random.default_rng(5)
    R0, t0 = raw["E"][0][:, :3], raw["E"][0][:, 3]
    for _ in range(50):
        Xw = R0.T @ (np.array([rng.uniform(-0.25, 0.25), rng.uniform(-0.2, 0.2), 1.0]) * rng.uniform(400, 4000) - t0)
        p0, p1 = g["P"][0] @ np.append(Xw, 1.0), g["P"][1] @ np.append(Xw, 1.0)
        x0, y0, x1, y1 = p0[0] / p0[2], p0[1] / p0[2], p1[0] / p1[2], p1[1] / p1[2]
        assert abs(y0 - y1) < 1e-9
        iW = 1.0 / (q[3, 3] + q[3, 2] * (x1 - x0))          # :744, disparity = x_other - x
        F = np.array([(q[0, 3] + x0) * iW, (y0 + q[1, 3]) * iW, q[2, 3] * iW])
        assert np.abs(g["R_final"] @ F + g["T_final"] - Xw).max() < 1e-9 * np.abs(Xw).max()


def test_rectify_pair_matches_oracle(ctx, raw):
    """(A regression guard: the oracle's Rectify is the same transcription of OpenCV's routines as the product's -- equality
    says the two dialects have not drifted apart, the property tests say what they compute is a rectification.)"""
    g = ctx.rectify_pair(raw["K"], raw["E"], raw["origin"], raw["lowest"], raw["pyr_levels"], raw["image"], raw["mask"])
    o = orc.rectify_pair(raw["K"], raw["E"], raw["origin"], raw["lowest"], raw["pyr_levels"], raw["image"], raw["mask"])
    for v in range(2):
        assert np.array_equal(g["image"][v], o["image"][v])
        assert np.array_equal(g["mask"][v], o["mask"][v])
        assert np.array_equal(g["P"][v], o["P"][v])
    for k in ("Q", "R_final", "T_final"):
        assert np.array_equal(g[k], o[k]), k


def test_rectify_then_match_end_to_end(ctx):
    """raw images + calibration -> Rectify -> MatchAllLayer body, all on the GPU, against the oracle chain.
    The second camera sits at -X so that view-0 disparities are positive: DisparityRefine's int(d-1.5)
    truncates toward zero (CStereoMatching.cpp:625), which biases NEGATIVE disparities by about -1 px in the
    reference itself (measured with the oracle: Z = 1412 instead of 1500 for baseline +60)."""
    raw = synth.make_raw_pair(baseline=-150.0)
    g = ctx.rectify_pair(raw["K"], raw["E"], raw["origin"], raw["lowest"], raw["pyr_levels"], raw["image"], raw["mask"],
                         radius=2, ws=0.03, offset=2)
    ctx.run_pair()
    res = ctx.download_pair()
    o = orc.rectify_pair(raw["K"], raw["E"], raw["origin"], raw["lowest"], raw["pyr_levels"], raw["image"], raw["mask"])
    W, H = g["size"]
    cfg = synth.PairConfig(width=W, height=H, pyr_levels=raw["pyr_levels"], radius=2, ws=0.03, offset=2,
                           origin_width=raw["origin"][0], image=o["image"], mask=o["mask"], Q=o["Q"],
                           R_final=o["R_final"], T_final=o["T_final"])
    ref = orc.match_pair(cfg)
    assert res.margin == ref["margin"] and res.n_points == ref["n_points"] and res.n_points > 1000
    for v in range(2):
        a, b = res.disparity[v], ref["disparity"][v]
        assert np.array_equal(a == -10000, b == -10000)
        ok = b != -10000
        assert np.array_equal(a[ok], b[ok])
    fin = np.isfinite(ref["xyz"])
    assert np.array_equal(res.xyz[fin], ref["xyz"][fin])
    # the scene is a plane at Z = depth in the camera-0 frame up to its small tilt: the cloud must be flat
    z = res.xyz[:, 2]
    assert abs(np.median(z) - raw["depth"]) < 0.02 * raw["depth"]


def test_mirror_rectifies_from_raw_inputs(ctx, raw):
    from reconstruction_amd import Camera, ManageData, StereoMatching
    cams = [Camera(camID=v, MatIntrinsics=raw["K"][v], MatExtrinsics=raw["E"][v]) for v in range(2)]
    for v in range(2):
        cams[v].raw_image, cams[v].raw_mask = raw["image"][v], raw["mask"][v]
    data = ManageData(cam=[cams], m_PyrmNum=raw["pyr_levels"], m_LowestLevelSize=raw["lowest"], m_OriginSize=raw["origin"])
    sm = StereoMatching(0)
    sm.Init(data, None, 2, 0.03)
    sm.Verbose = 0
    sm.MatchAllLayer()
    o = orc.rectify_pair(raw["K"], raw["E"], raw["origin"], raw["lowest"], raw["pyr_levels"], raw["image"], raw["mask"])
    assert np.array_equal(sm.Q, o["Q"]) and np.array_equal(data.cam[0][1].P, o["P"][1])
    assert np.array_equal(data.cam[0][0].mask, o["mask"][0]) and data.cam[0][0].bound is not None
    assert sm.last_result.n_points > 1000


def test_cli_config_to_ply(ctx, tmp_path):
    """python -m reconstruction_amd config.yml: configuration + calibration + image files -> Rectify -> MatchAllLayer ->
    PLY of the cloud, the reference's main() (main.cpp:5-23) without CloudOptimization::run."""
    from PIL import Image
    from reconstruction_amd import config as cfgmod
    from reconstruction_amd.__main__ import main
    raw = synth.make_raw_pair(baseline=-150.0)
    root = str(tmp_path) + "/"
    (tmp_path / "mask").mkdir()
    for j in range(2):
        Image.fromarray(raw["image"][j][:, :, ::-1]).save(root + "0001_Cam%d.png" % j)   # files are RGB, arrays BGR
        Image.fromarray(raw["mask"][j]).save(root + "mask/0001_Cam%d.png" % j)
    cfgmod.dump_opencv_yaml(root + "calib_camera.yml", {"intrinsic-0": raw["K"][0], "extrinsic-0": raw["E"][0],
                                                         "intrinsic-1": raw["K"][1], "extrinsic-1": raw["E"][1]})
    cfgmod.dump_opencv_yaml(root + "config.yml", {
        "filepath": root, "outfilename": root + "out", "isoutput": 0, "camera_calib_name": "calib_camera.yml",
        "PyrmNum": raw["pyr_levels"], "LowestLevelWidth": raw["lowest"][0], "LowestLevelHeight": raw["lowest"][1],
        "imagelist": ["0001_Cam%d.png" % j for j in range(2)], "masklist": ["mask\\0001_Cam%d.png" % j for j in range(2)],
        "camID": np.array([[0, 1]], np.uint8)})
    assert main([root + "config.yml"]) == 0
    head = open(root + "out.ply", "rb").read(400).decode("latin1")
    assert head.startswith("ply") and "element vertex" in head
    n = int(head.split("element vertex")[1].split()[0])
    o = orc.rectify_pair(raw["K"], raw["E"], raw["origin"], raw["lowest"], raw["pyr_levels"], raw["image"], raw["mask"])
    W, H = o["mask"][0].shape[1], o["mask"][0].shape[0]
    cfg = synth.PairConfig(width=W, height=H, pyr_levels=raw["pyr_levels"], radius=2, ws=0.03, offset=2,
                           origin_width=raw["origin"][0], image=o["image"], mask=o["mask"], Q=o["Q"],
                           R_final=o["R_final"], T_final=o["T_final"])
    assert n == orc.match_pair(cfg)["n_points"] and n > 1000


def test_cli_isoutput_and_filter(ctx, tmp_path, monkeypatch):
    """isoutput = 1: the rectified "%d_%d.jpg" images of Rectify (CStereoMatching.cpp:159-166) and the in-call cloud%d.ply of
    DisparityToCloud (.cpp:707-757) appear in the working directory; outfilename already ending in .ply is used as is
    (BatchProcess/main.cpp:56); --filter runs CCloudOptimization::filter's outlier removal per pair."""
    from PIL import Image
    from reconstruction_amd import config as cfgmod
    from reconstruction_amd.__main__ import main
    raw = synth.make_raw_pair(baseline=-150.0)
    root = str(tmp_path) + "/"
    (tmp_path / "mask").mkdir()
    for j in range(2):
        Image.fromarray(raw["image"][j][:, :, ::-1]).save(root + "0001_Cam%d.png" % j)
        Image.fromarray(raw["mask"][j]).save(root + "mask/0001_Cam%d.png" % j)
    cfgmod.dump_opencv_yaml(root + "calib_camera.yml", {"intrinsic-0": raw["K"][0], "extrinsic-0": raw["E"][0],
                                                         "intrinsic-1": raw["K"][1], "extrinsic-1": raw["E"][1]})
    cfgmod.dump_opencv_yaml(root + "config.yml", {
        "filepath": root, "outfilename": root + "scan1.ply", "isoutput": 1, "camera_calib_name": "calib_camera.yml",
        "PyrmNum": raw["pyr_levels"], "LowestLevelWidth": raw["lowest"][0], "LowestLevelHeight": raw["lowest"][1],
        "imagelist": ["0001_Cam%d.png" % j for j in range(2)], "masklist": ["mask\\0001_Cam%d.png" % j for j in range(2)],
        "camID": np.array([[0, 1]], np.uint8)})
    monkeypatch.chdir(tmp_path)
    assert main([root + "config.yml"]) == 0
    import os
    assert os.path.exists(root + "scan1.ply") and not os.path.exists(root + "scan1.ply.ply")
    n_all = int(open(root + "scan1.ply", "rb").read(400).decode("latin1").split("element vertex")[1].split()[0])
    assert os.path.exists("0_0.jpg") and os.path.exists("0_1.jpg") and os.path.exists("cloud0.ply")
    assert Image.open("0_0.jpg").size == (raw["lowest"][0] << (raw["pyr_levels"] - 1), raw["lowest"][1] << (raw["pyr_levels"] - 1))
    # (this scene's points are 4.3 units apart: with the reference's radius of 2.5 every normal is NaN, as in PCL)
    assert main([root + "config.yml", "--filter", "--mls-radius", "10", "--out", root + "filtered.ply"]) == 0
    n_f = int(open(root + "filtered.ply", "rb").read(400).decode("latin1").split("element vertex")[1].split()[0])
    assert 0.5 * n_all < n_f < n_all
    # the filtered cloud keeps its colours and carries the normals (pcl::PointNormal's nx, ny, nz, curvature)
    raw_f = open(root + "filtered.ply", "rb").read()
    hdr, body = raw_f.split(b"end_header\n", 1)
    assert b"property uchar red" in hdr and b"property float nx" in hdr and b"property float curvature" in hdr
    rec = np.frombuffer(body, dtype=[("xyz", "<f4", 3), ("bgr", "u1", 3), ("n", "<f4", 4)])
    assert len(rec) == n_f and rec["bgr"].any()
    nn = np.linalg.norm(rec["n"][:, :3], axis=1)        # NaN where fewer than 3 points lie within the search radius
    fin = np.isfinite(nn)
    assert fin.mean() > 0.9 and np.abs(nn[fin] - 1).max() < 1e-3   # unit normals wherever one exists: nearly everywhere on a plane
    # an unreadable first mask is reported like the reference's "read image ... error", not an assertion
    os.remove(root + "mask/0001_Cam0.png")
    assert main([root + "config.yml"]) == 1
