"""The CPU oracle against golden vectors produced by the REAL reference: oracle/_ref/ref_probe links the
reference's own CStereoMatching.cpp / CManageData.cpp objects and its vendored Armadillo 4.200 (compiled
where they lie in the build container; oracle/ref_probe/make_golden.py is the generating script).
Covers what can run without the OpenCV library: the Armadillo primitives, CManageData::WindowToVec,
CStereoMatching::FindMargin, ::OrderConstraint and ::UniquenessContraint<short|double>.  Everything must
agree bit for bit -- same operations in the same order."""
import os

import numpy as np
import pytest

from oracle import oracle as orc

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "ref_probe_golden.npz"))


def _n(prefix):
    return len([k for k in G.files if k.startswith("in__" + prefix)])


@pytest.mark.parametrize("i", range(8))
def test_armadillo_mean_norm_dot(i):
    a, b = G["in__arma_vec_%d" % i], G["in__arma_vec_b_%d" % i]
    mean, norm, dot, norm_centered = G["ref__arma_out_%d" % i]
    assert orc.arma_mean(a) == mean
    assert orc.arma_norm2(a) == norm
    assert orc.arma_dot(a, b) == dot
    assert orc.arma_norm2(a - orc.arma_mean(a)) == norm_centered


@pytest.mark.parametrize("i", range(10))
def test_armadillo_median(i):
    assert orc.arma_median_int(G["in__median_in_%d" % i]) == int(G["ref__median_out_%d" % i][0])


def test_window_to_vec():
    img = G["in__w2v_img"]
    for i, (x, y, w) in enumerate(G["in__w2v_cases"]):
        ref = G["ref__w2v_out_%d" % i]
        n, u = orc.window_to_vec(img, int(x), int(y), int(w))
        assert n == ref[0], (i, n, ref[0])
        assert np.array_equal(u, ref[1:]), i
    # the flat window: norm 0 is returned as 1 (CManageData.cpp:89)
    assert G["ref__w2v_out_%d" % (len(G["in__w2v_cases"]) - 1)][0] == 1.0


N_FM, N_OC, N_UQ, N_NCC = _n("fm_mask_"), _n("oc_disp_"), _n("uq_p_"), _n("ncc_imgA_")


def test_fixture_holds_the_round3_cases():
    assert (N_FM, N_OC, N_UQ, N_NCC) == (8, 7, 6, 9)


@pytest.mark.parametrize("i", range(N_FM))
def test_find_margin(i):
    m = orc.find_margin(G["in__fm_mask_%d" % i], int(G["in__fm_r_%d" % i][0]))
    assert list(m.astuple()) == list(G["ref__fm_out_%d" % i])


@pytest.mark.parametrize("i", range(N_OC))
def test_order_constraint(i):
    out = orc.order_constraint(G["in__oc_disp_%d" % i], tuple(int(v) for v in G["in__oc_margin_%d" % i]))
    ref = G["ref__oc_out_%d" % i]
    assert np.array_equal(out, ref), "%d pixels differ" % (out != ref).sum()
    assert (ref != G["in__oc_disp_%d" % i]).sum() > 50  # the case really exercises the greedy removal


@pytest.mark.parametrize("i", range(N_UQ))
def test_uniqueness_three_passes(i):
    m = [int(v) for v in G["in__uq_margins_%d" % i]]
    d0, d1 = orc.uniqueness(G["in__uq_p_%d" % i], G["in__uq_q_%d" % i], tuple(m[:6]), tuple(m[6:]))
    assert np.array_equal(d0, G["ref__uq_out0_%d" % i])
    assert np.array_equal(d1, G["ref__uq_out1_%d" % i])
    assert (G["ref__uq_out0_%d" % i] != G["in__uq_p_%d" % i]).sum() > 100


def test_order_constraint_golden_has_a_long_crossing_component():
    """Cases 4 and 5 are C2-shaped: one thrown pixel ties ~2000 pixels of a row into a single crossing component."""
    for i in (4, 5):
        d = G["in__oc_disp_%d" % i]
        m = [int(v) for v in G["in__oc_margin_%d" % i]]
        row = d[m[0], m[2]:m[3] + 1].astype(np.int64)
        t = (row + np.arange(m[2], m[3] + 1))[row != -10000]
        cuts = np.nonzero(np.maximum.accumulate(t)[:-1] <= np.minimum.accumulate(t[::-1])[::-1][1:])[0]
        longest = np.diff(np.concatenate([[-1], cuts, [len(t) - 1]])).max()
        assert longest >= 1800, longest


def ncc_expected_disparity(scores, r, W):
    """The strict-`>`-from--1 scan of LowestLevelInitialMatch (.cpp:204-218) over the REFERENCE's scores (all masks 255,
    both margins = the r-frame): first maximum wins, a row of scores that never exceeds -1 keeps NOMATCH."""
    ny, nx, _ = scores.shape
    out = np.full((ny, W), -10000, np.int16)
    for y in range(ny):
        for x in range(nx):
            best, arg = -1.0, -1
            for c in range(nx):
                v = scores[y, x, c]
                if v > best:
                    best, arg = v, c
            if arg != -1:
                out[y, x + r] = arg - x
    return out


@pytest.mark.parametrize("i", range(N_NCC))
def test_lowest_level_match_against_reference_scores(i):
    """The oracle's LowestLevelInitialMatch picks, for every pixel, the candidate the reference's own fp64 scores
    (WindowToVec + arma::dot of the compiled reference, incl. exact and near ties) make the first maximum."""
    A, B, r = G["in__ncc_imgA_%d" % i], G["in__ncc_imgB_%d" % i], int(G["in__ncc_r_%d" % i][0])
    H, W = A.shape[:2]
    sc = G["ref__ncc_scores_%d" % i]
    want = ncc_expected_disparity(sc, r, W)
    mask = np.full((H, W), 255, np.uint8)
    mg = (r, H - 1 - r, r, W - 1 - r, W - 2 * r, H - 2 * r)
    got = orc.lowest_level_initial_match(A, B, mask, mask, r, mg, mg)
    assert np.array_equal(got[r:H - r], want)
    assert (got[:r] == -10000).all() and (got[H - r:] == -10000).all()
    srt = np.sort(sc, axis=2)
    gap = srt[:, :, -1] - srt[:, :, -2]
    if i in (2, 3, 6):      # periodic / saturated / binary textures: the maximum is shared -> the first-maximum rule decides
        assert (gap == 0).mean() > 0.05
    if i == 6:              # mathematically equal scores that differ in their last bits: the summation ORDER decides
        assert ((gap > 0) & (gap < 1e-11)).sum() >= 20


HL_CASES = ((1.0, 2), (-1.5, 3), (0.25, 5))


def ncc_expected_high_level(scores, r, H, W, pd, offset):
    """HighLevelInitialMatch (.cpp:255-301) with a constant parent disparity pd, all masks 255 and both margins = the
    r-frame: candidates [x + int(2 pd + 0.5) -+ offset] clipped to the other margin, first maximum of the REFERENCE's
    scores, strict `>` from -1."""
    out = np.full((H, W), -10000, np.int16)
    XL1, XR1 = r, W - 1 - r
    for y in range(r, H - r):
        for x in range(r, W - r):
            c = x + int(pd * 2 + 0.5)           # int(): truncation toward zero, as the C cast
            best, arg = -1.0, None
            for cand in range(max(c - offset, XL1), min(c + offset, XR1) + 1):
                v = scores[y - r, x - r, cand - r]
                if v > best:
                    best, arg = v, cand
            if arg is not None:
                out[y, x] = arg - x
    return out


@pytest.mark.parametrize("i", range(N_NCC))
def test_high_level_match_against_reference_scores(i):
    A, B, r = G["in__ncc_imgA_%d" % i], G["in__ncc_imgB_%d" % i], int(G["in__ncc_r_%d" % i][0])
    H, W = A.shape[:2]
    mask = np.full((H, W), 255, np.uint8)
    mg = (r, H - 1 - r, r, W - 1 - r, W - 2 * r, H - 2 * r)
    for pd, offset in HL_CASES:
        parent = np.full(((H + 1) // 2 + 1, (W + 1) // 2 + 1), pd, np.float64)
        got = orc.high_level_initial_match(A, B, mask, mask, r, offset, mg, mg, parent)
        assert np.array_equal(got, ncc_expected_high_level(G["ref__ncc_scores_%d" % i], r, H, W, pd, offset)), (i, pd, offset)


N_XI = _n("xi_imgA_")


@pytest.mark.parametrize("i", range(N_XI))
def test_refine_data_term_against_the_reference(i):
    """DisparityRefine's only non-trivial arithmetic besides exp -- the matching cost xi of 3x3x3 windows
    (CStereoMatching.cpp:624-629) -- against the compiled reference's own WindowToVec + arma::dot in its call-site
    expression (1 - dot / (normL * normR)) / 2, for every (own column, right-window left edge) pair of whole rows: 8-bit
    noise, the bench's band-limited texture, two- / three-level textures, flat and saturated windows (norm 0 -> 1).  The
    oracle's refine loop calls this very function (stereo_oracle.c: refine_xi).  Bit for bit."""
    A, B = G["in__xi_imgA_%d" % i], G["in__xi_imgB_%d" % i]
    got, ref = orc.refine_xi_table(A, B), G["ref__xi_table_%d" % i]
    assert got.shape == ref.shape
    assert np.array_equal(got.view(np.int64), ref.view(np.int64)), int((got.view(np.int64) != ref.view(np.int64)).sum())
