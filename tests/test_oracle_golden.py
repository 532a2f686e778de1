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


@pytest.mark.parametrize("i", range(4))
def test_find_margin(i):
    m = orc.find_margin(G["in__fm_mask_%d" % i], int(G["in__fm_r_%d" % i][0]))
    assert list(m.astuple()) == list(G["ref__fm_out_%d" % i])


@pytest.mark.parametrize("i", range(4))
def test_order_constraint(i):
    out = orc.order_constraint(G["in__oc_disp_%d" % i], tuple(int(v) for v in G["in__oc_margin_%d" % i]))
    ref = G["ref__oc_out_%d" % i]
    assert np.array_equal(out, ref), "%d pixels differ" % (out != ref).sum()
    assert (ref != G["in__oc_disp_%d" % i]).sum() > 50  # the case really exercises the greedy removal


@pytest.mark.parametrize("i", range(4))
def test_uniqueness_three_passes(i):
    m = [int(v) for v in G["in__uq_margins_%d" % i]]
    d0, d1 = orc.uniqueness(G["in__uq_p_%d" % i], G["in__uq_q_%d" % i], tuple(m[:6]), tuple(m[6:]))
    assert np.array_equal(d0, G["ref__uq_out0_%d" % i])
    assert np.array_equal(d1, G["ref__uq_out1_%d" % i])
    assert (G["ref__uq_out0_%d" % i] != G["in__uq_p_%d" % i]).sum() > 100
