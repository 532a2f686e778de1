"""Host-side pieces that need no GPU: the synthetic workload generator and the reference-shaped mirror."""
import numpy as np

from reconstruction_amd import synth


def test_synthetic_pairs_are_seeded_and_shaped():
    a = synth.config_small(96, 64, 2, pair=3)
    b = synth.config_small(96, 64, 2, pair=3)
    for v in range(2):
        assert np.array_equal(a.image[v], b.image[v]) and np.array_equal(a.mask[v], b.mask[v])
        assert a.image[v].shape == (64, 96, 3) and a.image[v].dtype == np.uint8
    c = synth.config_small(96, 64, 2, pair=4)
    assert not np.array_equal(a.image[0], c.image[0])
    # odd pairs have negative disparities (exercises the int() truncation asymmetry)
    assert a.true_disparity.mean() < 0 < c.true_disparity.mean()


def test_baseline_configs_have_the_documented_shapes():
    c1 = synth.config_c1()
    assert (c1.width, c1.height, c1.pyr_levels, c1.radius) == (640, 480, 3, 2)
    top = 1 << (c1.pyr_levels - 1)
    assert (c1.mask[0][c1.height // 2] == 255).sum() == 64 * top   # 64 disparities at the lowest level
    cs = synth.config_c2_sample()
    assert (cs.pyr_levels, cs.radius, cs.offset) == (5, 5, 2)


def test_calibration_q_matrix_form():
    Q, R, T = synth.pinhole_calibration(640, 480, 0)
    assert Q[3, 2] < 0 and Q[2, 3] == 1.2 * 640 and np.array_equal(R, np.eye(3)) and not T.any()
    Q, R, T = synth.pinhole_calibration(640, 480, 2)
    assert np.allclose(R @ R.T, np.eye(3))


def test_mirror_classes_keep_the_reference_surface():
    from reconstruction_amd.api import ManageData, StereoMatching, Camera
    import inspect
    sig = inspect.signature(StereoMatching.Init)
    assert list(sig.parameters)[1:] == ["data", "CloudOptimization", "radii", "ws", "disparity_offset"]
    assert sig.parameters["radii"].default == 2 and sig.parameters["ws"].default == 0.5
    assert sig.parameters["disparity_offset"].default == 2          # CStereoMatching.h:47
    assert hasattr(StereoMatching, "MatchAllLayer")
    d = ManageData(cam=[[Camera(camID=0), Camera(camID=1)]], m_PyrmNum=4)
    assert d.m_CampairNum == 1


def test_ply_writer_matches_the_reference_format(tmp_path):
    """cloud%d.ply as CStereoMatching::DisparityToCloud writes it (.cpp:723-729 header, :754-756 records)."""
    import struct
    from reconstruction_amd import write_ply
    rng = np.random.default_rng(3)
    xyz = rng.normal(0, 500, (37, 3))
    bgr = rng.integers(0, 256, (37, 3)).astype(np.uint8)
    p = tmp_path / "cloud0.ply"
    write_ply(p, xyz, bgr)
    raw = p.read_bytes()
    header = (b"ply\nformat binary_little_endian 1.0\nelement vertex 37\nproperty float x\nproperty float y\n"
              b"property float z\nproperty uchar blue\nproperty uchar green\nproperty uchar red\nend_header\n")
    assert raw.startswith(header)
    body = raw[len(header):]
    assert len(body) == 37 * 15
    for i in (0, 17, 36):
        x, y, z, b, g, r = struct.unpack_from("<fffBBB", body, 15 * i)
        assert (x, y, z) == tuple(np.float32(xyz[i])) and (b, g, r) == tuple(bgr[i])
    write_ply(tmp_path / "empty.ply", np.zeros((0, 3)), np.zeros((0, 3), np.uint8))
    assert b"element vertex 0\n" in (tmp_path / "empty.ply").read_bytes()


def test_config_and_calibration_loader(tmp_path):
    """CReconstrction::Init / CManageData::Init on an OpenCV-YAML config + calibration + images (schema:
    CManageData.cpp:26-66, BatchProcess/main.cpp:53-72)."""
    from PIL import Image
    from reconstruction_amd import config as cfgmod
    root = str(tmp_path) + "/"
    (tmp_path / "mask").mkdir()
    rng = np.random.default_rng(0)
    for j in range(3):
        Image.fromarray(rng.integers(0, 255, (48, 64, 3)).astype(np.uint8)).save(root + "0001_Cam%d.png" % j)
        Image.fromarray(np.full((48, 64), 255, np.uint8)).save(root + "mask/0001_Cam%d.png" % j)
    K = np.array([[800.0, 0, 32], [0, 800.0, 24], [0, 0, 1]])
    calib = {}
    for j in range(3):
        E = np.hstack([np.eye(3), np.array([[-100.0 * j], [0], [0]])])
        calib["intrinsic-%d" % j] = K
        calib["extrinsic-%d" % j] = E
    cfgmod.dump_opencv_yaml(root + "calib_camera.yml", calib)
    cfgmod.dump_opencv_yaml(root + "config.yml", {
        "filepath": root, "outfilename": root + "1.ply", "isoutput": 0, "camera_calib_name": "calib_camera.yml",
        "PyrmNum": 3, "LowestLevelWidth": 16, "LowestLevelHeight": 12,
        "imagelist": ["0001_Cam%d.png" % j for j in range(3)],
        "masklist": ["mask\\0001_Cam%d.png" % j for j in range(3)],   # Windows separators, as BatchProcess writes
        "camID": np.array([[0, 1], [2, 1]], np.uint8)})
    assert open(root + "config.yml").read().startswith("%YAML:1.0")
    data, info = cfgmod.load_config(root + "config.yml")
    assert data.m_CampairNum == 2 and data.m_PyrmNum == 3 and data.m_LowestLevelSize == (16, 12)
    assert data.m_OriginSize == (64, 48) and data.m_CameraNum == 3
    assert [c.camID for c in data.cam[1]] == [2, 1]
    assert np.array_equal(data.cam[1][0].MatIntrinsics, K)
    assert np.allclose(data.cam[1][0].CamCenter, [200.0, 0, 0])          # -R^T t
    img = cfgmod.imread_bgr(data.cam[0][0].image_name)
    assert img.shape == (48, 64, 3)
    assert cfgmod.imread_gray(data.cam[0][1].mask_name).max() == 255
    assert cfgmod.imread_bgr(root + "missing.jpg") is None              # reference: silent return (.cpp:147-151)
    import pytest
    with pytest.raises(FileNotFoundError):
        cfgmod.load_config(root + "nope.yml")


def test_cli_reports_a_missing_configuration_like_the_reference(tmp_path, capsys):
    """python -m reconstruction_amd: 'cannot open file ...' and a non-zero exit (CReconstruction.cpp:9-13), before
    anything touches the GPU."""
    from reconstruction_amd.__main__ import main
    rc = main([str(tmp_path / "nope.yml")])
    assert rc != 0
    assert "cannot open file" in capsys.readouterr().out
