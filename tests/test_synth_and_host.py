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
