"""Known-answer tests of the CPU oracle that follow from the reference source -- they pin
the stages the reference probe cannot run (everything that allocates a cv::Mat)."""
import numpy as np
import pytest

from oracle import oracle as orc
from reconstruction_amd import synth

from helpers import NOMATCH, host_libm_is_glibc_with_fma, oracle_stages


def test_ncc_of_identical_windows_is_one_and_flat_is_zero():
    rng = np.random.default_rng(1)
    img = rng.integers(0, 256, (12, 12, 3)).astype(np.uint8)
    n, u = orc.window_to_vec(img, 2, 3, 5)
    assert abs(orc.arma_dot(u / n, u) / n - 1.0) < 1e-14
    flat = np.full((12, 12, 3), 77, np.uint8)
    n, u = orc.window_to_vec(flat, 2, 3, 5)
    assert n == 1.0 and not u.any()  # normu == 0 ? 1 (CManageData.cpp:89) -> every score against it is 0


def test_integer_shift_scene_lowest_level():
    """view1(x) = view0(x - 7): every masked interior pixel must match at disparity exactly 7 (and -7 back)."""
    W, H, r, sh = 96, 40, 2, 7
    img0 = synth.make_texture(W, H, 5)
    img1 = np.roll(img0, sh, axis=1)
    m0 = np.zeros((H, W), np.uint8); m0[6:H - 6, 12:60] = 255
    m1 = np.roll(m0, sh, axis=1)
    mg0, mg1 = orc.find_margin(m0, r).astuple(), orc.find_margin(m1, r).astuple()
    d0 = orc.lowest_level_initial_match(img0, img1, m0, m1, r, mg0, mg1)
    d1 = orc.lowest_level_initial_match(img1, img0, m1, m0, r, mg1, mg0)
    assert (d0[m0 == 255] == sh).all() and (d0[m0 != 255] == NOMATCH).all()
    assert (d1[m1 == 255] == -sh).all()


def test_high_level_follows_parent_and_carries_stale_bounds():
    W, H, r, off = 64, 24, 2, 2
    img0 = synth.make_texture(W, H, 9)
    img1 = np.roll(img0, 6, axis=1)
    m0 = np.zeros((H, W), np.uint8); m0[4:H - 4, 8:40] = 255
    m1 = np.roll(m0, 6, axis=1)
    mg0, mg1 = orc.find_margin(m0, r).astuple(), orc.find_margin(m1, r).astuple()
    parent = np.full((H // 2, W // 2), 3.0)       # 2 * 3 = 6 at this level
    d = orc.high_level_initial_match(img0, img1, m0, m1, r, off, mg0, mg1, parent)
    assert (d[m0 == 255] == 6).all()
    parent[:, :] = NOMATCH                          # no parent anywhere: L = XL1, R = XR1 for the whole row
    d = orc.high_level_initial_match(img0, img1, m0, m1, r, off, mg0, mg1, parent)
    assert (d[m0 == 255] == 6).all()                # full-range search still finds the true shift


def test_median_filter_rules():
    H, W = 9, 12
    d = np.full((H, W), NOMATCH, np.int16)
    mask = np.full((H, W), 255, np.uint8)
    own = (1, H - 2, 1, W - 2, W - 2, H - 2)
    d[3:6, 4:6] = [[2, 9], [4, 5], [9, 2]]        # centre (4,5)=5 sees columns 4,5 of rows 3..5
    out = orc.median_filter(d, mask, own)
    # samples for (y=4, x=5): cols {4,5} x rows {3,4,5} = {2,9,4,5,9,2} -> sorted 2,2,4,5,9,9 -> 4 + (5-4)/2 = 4
    assert out[4, 5] == 4
    # a NOMATCH centre with >= 4 valid samples gets the median, with fewer stays NOMATCH
    assert out[4, 6] == NOMATCH                     # only column 5 (3 samples) + column 6 (0) = 3 samples
    d2 = d.copy(); d2[4, 4] = NOMATCH
    assert orc.median_filter(d2, mask, own)[4, 4] == NOMATCH  # col 3 empty, col 4 has 2 valid -> k = 2 < 4


def test_pyr_down_constant_and_size():
    a = np.full((20, 36, 3), 131, np.uint8)
    b = orc.pyr_down(a)
    assert b.shape == (10, 18, 3) and (b == 131).all()
    m = np.zeros((16, 16), np.uint8); m[:, 8:] = 255
    e = orc.pyr_down(m)
    # [1 4 6 4 1]/16 across the step: columns straddling the edge are strictly between 0 and 255
    assert e[0, 3] < 255 and e[0, 4] > 0 and e[0, 0] == 0 and e[0, 7] == 255


def test_erode_ellipse_small_kernels():
    m = np.zeros((15, 15), np.uint8); m[3:12, 3:12] = 255
    e3 = orc.erode_ellipse(m, 3)                    # 3x3 ellipse = plus-shaped cross
    assert e3[3, 3] == 0 and e3[4, 4] == 255 and e3[3, 7] == 0 and e3[7, 7] == 255
    e1 = orc.erode_ellipse(m, 1)
    assert np.array_equal(e1, m)


def test_smooth_constraint_kills_isolated_and_disagreeing_pixels():
    H, W = 10, 14
    own = (2, H - 3, 2, W - 3, W - 4, H - 4)
    d = np.full((H, W), NOMATCH, np.int16)
    d[5, 9] = 3                                     # isolated -> total == 0 -> killed
    d[3:5, 3:6] = 1                                 # consistent block survives
    d[3, 4] = 9                                     # disagrees with all its neighbours -> killed
    out = orc.smooth_constraint(d, own)
    assert out[5, 9] == NOMATCH and out[3, 4] == NOMATCH
    assert out[4, 4] == 1 and out[4, 3] == 1


def test_set_boundary_degenerate_margin_is_reported():
    d = np.zeros((8, 8), np.int16)
    st, _, _ = orc.set_boundary_smooth(d, np.zeros((8, 8), np.uint8), (5, 2, 5, 2, -2, -2), (5, 2, 5, 2, -2, -2))
    assert st == -2                                 # the reference calls exit(0) here (.cpp:827-830)


def test_refine_keeps_nomatch_and_border_ring():
    cfg = synth.config_small(96, 64, 2)
    rec, fin = oracle_stages(cfg, max_levels=1)
    q = [r for r in rec if r["stage"] == "refine" and r["v"] == 0][0]
    out, inp = q["out"], q["inp"].astype(np.float64)
    YL, YR, XL, XR = q["mg"][0][:4]
    assert np.array_equal(out == NOMATCH, inp == NOMATCH)
    ring = np.ones(out.shape, bool); ring[YL + 1:YR, XL + 1:XR] = False
    assert np.array_equal(out[ring], inp[ring])     # only the interior is ever updated (.cpp:595,611)


def test_cloud_of_a_fronto_parallel_plane_has_constant_depth():
    W, H = 64, 48
    d = np.full((H, W), 5.0)
    mask = np.full((H, W), 255, np.uint8)
    img = np.zeros((H, W, 3), np.uint8); img[..., 1] = 200
    Q, R, T = synth.pinhole_calibration(W, H, 0)
    own = orc.find_margin(mask, 2).astuple()
    xyz, bgr = orc.disparity_to_cloud(d, mask, img, Q, 1.0, R, T, own)
    assert len(xyz) > 0 and np.ptp(xyz[:, 2]) < 1e-9
    assert (bgr[:, 1] == 200).all()
    # Z = q23 / (q32 * d) with Q as Rectify leaves it: f / (-1/B * d)
    assert abs(xyz[0, 2] - (1.2 * W) / (-0.01 * 5.0)) < 1e-9


@pytest.mark.parametrize("case", [dict(width=96, height=64, levels=2), dict(width=128, height=96, levels=3, pair=1, holes=True,
                                  mask_l0_width=20, border_l0=4)])
def test_staged_sequence_equals_whole_pair(case):
    """The 17-call MatchOneLayer sequence driven from Python reproduces orc_match_pair exactly."""
    cfg = synth.config_small(**case)
    rec, fin = oracle_stages(cfg)
    ref = orc.match_pair(cfg)
    assert ref["status"] == 0
    for v in range(2):
        assert np.array_equal(ref["disparity"][v], fin["disparity"][v])
    assert ref["margin"] == [tuple(m) for m in fin["margin"]]
    assert 0 < ref["n_points"] <= ref["v_top"]


def test_truncation_asymmetry_for_negative_disparities():
    """int(d - 1.5) truncates toward zero: the reference's refinement is biased for negative disparities
    (kept on purpose)."""
    pos = orc.match_pair(synth.config_small(160, 96, 2, pair=0))
    neg = orc.match_pair(synth.config_small(160, 96, 2, pair=1))
    def bias(res, cfg_pair):
        cfg = synth.config_small(160, 96, 2, pair=cfg_pair)
        d = res["disparity"][0]; ok = d != NOMATCH
        return float(np.mean(d[ok] - cfg.true_disparity[ok]))
    assert abs(bias(pos, 0)) < 0.3
    assert bias(neg, 1) < -0.5


def test_specified_exp_is_a_faithful_exp():
    """orc_exp_neg (the fully specified exp(-t) the oracle and the GPU kernels share): within 1 ulp of the host libm on
    a dense sample incl. the subnormal range, exact at 0, 0 beyond the underflow threshold."""
    import math
    import struct
    rng = np.random.default_rng(0)
    ts = np.concatenate([rng.random(60000) * 2, rng.random(30000) * 40, rng.random(20000) * 800,
                         10.0 ** rng.uniform(-12, 0, 10000), np.arange(0, 130, 1.0),
                         [0.0, 1e-9, 0.34657359027997264, 0.3465735902799727, 708.3, 709.0, 710.0, 745.0, 745.13, 745.2]])
    worst = 0
    for t in ts:
        a, b = orc.exp_neg(float(t)), math.exp(-float(t))
        ia, ib = struct.unpack("<q", struct.pack("<d", a))[0], struct.unpack("<q", struct.pack("<d", b))[0]
        worst = max(worst, abs(ia - ib))
    assert worst <= 1
    assert orc.exp_neg(0.0) == 1.0 and orc.exp_neg(746.0) == 0.0 and orc.exp_neg(1e300) == 0.0


def exp_test_arguments():
    """Arguments of the specified exp: dense over [0, 2], the whole range up to and beyond the underflow threshold, the
    k switch points n * ln2 / 256 (table index changes) and n * ln2 / 2 with their neighbours, the 512 / 1024 limits of the
    special-case path, the subnormal results (t in [708.4, 745.14]), tiny and huge t."""
    rng = np.random.default_rng(7)
    ln2 = float(np.log(2.0))
    sw = np.arange(1, 2200) * (ln2 / 2)
    sw2 = (np.arange(1, 140000) + 0.5) * (ln2 / 128)     # where round-to-nearest switches the table index, up to t = 758
    edge = np.array([512.0, 1024.0, 708.3964185322641, 745.1332191019411])
    return np.concatenate([rng.random(200000) * 2, rng.random(100000) * 40, rng.random(100000) * 800, 708.0 + rng.random(100000) * 38,
                           505.0 + rng.random(50000) * 14, 1000.0 + rng.random(20000) * 50,
                           10.0 ** rng.uniform(-300, 0, 20000), 10.0 ** rng.uniform(0, 300, 2000), np.arange(0, 800, 0.25),
                           sw, np.nextafter(sw, 0), np.nextafter(sw, 1e9), sw2, np.nextafter(sw2, 0), np.nextafter(sw2, 1e9),
                           edge, np.nextafter(edge, 0), np.nextafter(edge, 1e9),
                           [0.0, 5e-324, 1e-9, 2.0 ** -54, 2.0 ** -53, 0.34657359027997264, 0.3465735902799727, 708.3, 709.0, 710.0, 745.0, 745.13,
                            745.13321910194110842, 745.1332191019412, 745.2, 1e19, 1e300, np.inf]])


def test_specified_exp_does_not_depend_on_the_fma_implementation():
    """Every step of orc_exp_neg is a correctly rounded operation, so the CPU instruction and the C library's fma() must
    give the same bits (the property that makes the specification portable)."""
    t = exp_test_arguments()
    a, b = orc.exp_neg_array(t), orc.exp_neg_array(t, soft_fma=True)
    assert np.array_equal(a.view(np.int64), b.view(np.int64))
    assert a[np.isinf(t)].tolist() == [0.0] and (a >= 0).all() and (a <= 1).all()
    assert a[t >= 1024.0].max() == 0.0 and a[t == 0.0].min() == 1.0
    ulp = np.abs(a.view(np.int64) - np.exp(-t).view(np.int64))
    assert ulp.max() <= 1, ulp.max()


def test_specified_exp_is_the_host_libms_exp():
    """The pin of DisparityRefine's one libm call (CStereoMatching.cpp:665-666).  The specification restates glibc 2.35's
    published exp algorithm in the operation order of its FMA build, so on such a host (this image; the GPU box) orc_exp_neg
    and the C runtime's exp() agree in EVERY bit -- over the weights' range, the table-index switch points, the special-case
    range [512, 1024), the subnormal results and beyond.  Elsewhere (another libm, no FMA: glibc then runs the same algorithm
    with separate multiplies and adds) it is held to libm grade: within 1 ulp, last-bit disagreement <= 0.3 %.  A second,
    independent libm-grade exp (expl rounded to double, i.e. nearly correctly rounded) disagrees in < 0.1 % of the arguments."""
    rng = np.random.default_rng(3)
    t = np.concatenate([exp_test_arguments(), rng.random(3000000) * 30, rng.random(1000000) * 1100,
                        10.0 ** rng.uniform(-320, 3, 500000), 700.0 + rng.random(1000000) * 50])
    spec = orc.exp_neg_array(t)
    try:
        orc.set_exp_mode(1)
        libm = orc.exp_neg_array(t)
        orc.set_exp_mode(2)
        expl = orc.exp_neg_array(t)
    finally:
        orc.set_exp_mode(0)
    d_libm = spec.view(np.int64) != libm.view(np.int64)
    d_expl = spec.view(np.int64) != expl.view(np.int64)
    print("specified exp vs host libm exp: %d of %d differ; vs expl rounded: %d (%.4f %%)" % (d_libm.sum(), t.size, d_expl.sum(), 100 * d_expl.mean()))
    if host_libm_is_glibc_with_fma():
        assert d_libm.sum() == 0, t[d_libm][:10]
    else:
        assert d_libm.mean() <= 3e-3 and np.abs(spec.view(np.int64) - libm.view(np.int64)).max() <= 1
    assert d_expl.mean() < 1e-3 and np.abs(spec.view(np.int64) - expl.view(np.int64)).max() <= 1


def test_exp_table_is_derived_from_first_principles():
    """Both generated copies of the 128-entry table (oracle/exp_table.h, csrc/exp_table.h) are 2^(j/128) split as
    H (1 + T) -- re-derived here with 200-bit arithmetic, not copied from a libm."""
    import os
    import re
    import struct
    mp = pytest.importorskip("mpmath")
    mp.mp.prec = 200
    want = []
    for j in range(128):
        ex = mp.power(2, mp.mpf(j) / 128)
        H = float(ex)
        T = float((ex - mp.mpf(H)) / mp.mpf(H))
        assert abs(T) < 2.0 ** -53 and 1.0 <= H < 2.0
        want += [struct.unpack("<Q", struct.pack("<d", T))[0], struct.unpack("<Q", struct.pack("<d", H))[0] - (j << 45)]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for rel in ("oracle/exp_table.h", "reconstruction_amd/csrc/exp_table.h"):
        with open(os.path.join(root, rel)) as f:
            got = [int(v, 16) for v in re.findall(r"0x([0-9a-f]{16})ull", f.read())]
        assert got == want, rel
