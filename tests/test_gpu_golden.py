"""The reference's golden vectors straight in front of the HIP kernels, through the C ABI (no oracle in between):
tests/golden/ref_probe_golden.npz holds inputs and the outputs of the COMPILED reference (oracle/ref_probe links
CStereoMatching.cpp / CManageData.cpp and the vendored Armadillo where they lie; oracle/ref_probe/make_golden.py made
the fixture).  FindMargin (.cpp:1011-1038), OrderConstraint (.cpp:310-368), UniquenessContraint<short|double>
(.cpp:450-497) and the NCC argmax of LowestLevelInitialMatch (.cpp:202-218, scores = WindowToVec + arma::dot of the
reference, incl. exact and last-bit ties) must come out bit for bit.

The module also re-runs tests/test_oracle_golden.py under the gpu mark, so the GPU box's own toolchain shows
oracle == reference next to HIP == reference."""
import numpy as np
import pytest

import test_oracle_golden as cpu_side
from test_oracle_golden import G, N_FM, N_NCC, N_OC, N_UQ, N_XI, ncc_expected_disparity

pytestmark = pytest.mark.gpu
NOMATCH = -10000


@pytest.mark.parametrize("i", range(N_FM))
def test_hip_find_margin_equals_the_reference(ctx, i):
    m = ctx.find_margin(G["in__fm_mask_%d" % i], int(G["in__fm_r_%d" % i][0]))
    assert list(m.astuple()) == list(G["ref__fm_out_%d" % i])


@pytest.mark.parametrize("i", range(N_OC))
def test_hip_order_constraint_equals_the_reference(ctx, i):
    d = G["in__oc_disp_%d" % i]
    out = ctx.order_constraint(d, tuple(int(v) for v in G["in__oc_margin_%d" % i]))
    ref = G["ref__oc_out_%d" % i]
    assert np.array_equal(out, ref), "%d pixels differ" % (out != ref).sum()


@pytest.mark.parametrize("i", range(N_UQ))
def test_hip_uniqueness_equals_the_reference(ctx, i):
    m = [int(v) for v in G["in__uq_margins_%d" % i]]
    p, q = G["in__uq_p_%d" % i], G["in__uq_q_%d" % i]
    d0, d1 = ctx.uniqueness(p, q, tuple(m[:6]), tuple(m[6:]))
    assert d0.dtype == p.dtype
    assert np.array_equal(d0, G["ref__uq_out0_%d" % i]), (d0 != G["ref__uq_out0_%d" % i]).sum()
    assert np.array_equal(d1, G["ref__uq_out1_%d" % i]), (d1 != G["ref__uq_out1_%d" % i]).sum()


@pytest.mark.parametrize("i", range(N_NCC))
def test_hip_lowest_level_match_equals_the_reference_scores_argmax(ctx, i):
    """All masks 255, both margins = the r-frame: every pixel scans every candidate of its row; the reference's own fp64
    scores decide the expected column (first maximum of a strict `>` scan from -1)."""
    A, B, r = G["in__ncc_imgA_%d" % i], G["in__ncc_imgB_%d" % i], int(G["in__ncc_r_%d" % i][0])
    H, W = A.shape[:2]
    want = ncc_expected_disparity(G["ref__ncc_scores_%d" % i], r, W)
    mask = np.full((H, W), 255, np.uint8)
    mg = (r, H - 1 - r, r, W - 1 - r, W - 2 * r, H - 2 * r)
    got = ctx.initial_match(A, B, mask, mask, r, 2, mg, mg)
    bad = got[r:H - r] != want
    assert not bad.any(), "%d of %d pixels pick another column than the reference's scores" % (bad.sum(), bad.size)
    assert (got[:r] == NOMATCH).all() and (got[H - r:] == NOMATCH).all()


@pytest.mark.parametrize("i", range(N_NCC))
def test_hip_high_level_match_equals_the_reference_scores_argmax(ctx, i):
    """HighLevelInitialMatch (.cpp:255-301) on the same rows: a constant parent disparity p gives every pixel the
    interval [x + 2p - offset, x + 2p + offset] clipped to the other margin; the expected column is the first maximum
    of the REFERENCE's scores over that interval."""
    A, B, r = G["in__ncc_imgA_%d" % i], G["in__ncc_imgB_%d" % i], int(G["in__ncc_r_%d" % i][0])
    H, W = A.shape[:2]
    sc = G["ref__ncc_scores_%d" % i]
    mask = np.full((H, W), 255, np.uint8)
    mg = (r, H - 1 - r, r, W - 1 - r, W - 2 * r, H - 2 * r)
    for pd, offset in cpu_side.HL_CASES:
        parent = np.full(((H + 1) // 2 + 1, (W + 1) // 2 + 1), pd, np.float64)
        got = ctx.initial_match(A, B, mask, mask, r, offset, mg, mg, parent=parent)
        want = cpu_side.ncc_expected_high_level(sc, r, H, W, pd, offset)
        assert np.array_equal(got, want), (i, pd, offset, int((got != want).sum()))


# ---- oracle == reference, shown on the GPU box as well ---------------------------------------------------------------
test_oracle_armadillo_on_the_gpu_box = cpu_side.test_armadillo_mean_norm_dot
test_oracle_median_on_the_gpu_box = cpu_side.test_armadillo_median
test_oracle_window_to_vec_on_the_gpu_box = cpu_side.test_window_to_vec
@pytest.mark.parametrize("form", [0, 1, 2])
@pytest.mark.parametrize("i", range(N_XI))
def test_hip_refine_data_term_equals_the_reference(ctx, i, form):
    """The three device restatements of DisparityRefine's matching cost (k_refine.hip: refine_left + refine_cost_left of the
    first sweep, refine_data_term_packed = a lane per cache miss, refine_data_term_quad = four lanes per miss) against the
    COMPILED reference's own (1 - arma::dot(vecL, vecR) / (normL * normR)) / 2 (CStereoMatching.cpp:624-629) on whole rows,
    through the C ABI with no oracle in between: every bit, flat windows and equal-cost textures included."""
    A, B = G["in__xi_imgA_%d" % i], G["in__xi_imgB_%d" % i]
    ref = G["ref__xi_table_%d" % i]
    got = ctx.refine_xi(A, B, form)
    nw = ref.shape[2]
    for c in range(3):   # [c, y, x, col] = xi(x, y, col + c): compared wherever col + c is a window inside the row
        a, b = got[c, :, :, :nw - c], ref[:, :, c:]
        assert np.array_equal(a.view(np.int64), b.view(np.int64)), (form, c, int((a.view(np.int64) != b.view(np.int64)).sum()))


test_oracle_refine_data_term_on_the_gpu_box = cpu_side.test_refine_data_term_against_the_reference
test_oracle_find_margin_on_the_gpu_box = cpu_side.test_find_margin
test_oracle_order_constraint_on_the_gpu_box = cpu_side.test_order_constraint
test_oracle_uniqueness_on_the_gpu_box = cpu_side.test_uniqueness_three_passes
test_oracle_ncc_argmax_on_the_gpu_box = cpu_side.test_lowest_level_match_against_reference_scores
test_oracle_high_level_argmax_on_the_gpu_box = cpu_side.test_high_level_match_against_reference_scores


def test_hip_specified_exp_equals_the_oracle_bit_for_bit(ctx):
    """DisparityRefine's only transcendental: the device evaluation of the specified exp(-t) (k_refine.hip: exp_neg -- glibc
    2.35's table-driven algorithm, the table in LDS, the scale by an integer add on the table word, glibc's special case for
    t in [512, 1024)) against the oracle's over 1 M arguments incl. every table-index switch point, the special-case limits
    and the whole subnormal range -- equal bit patterns, in the general form and in the form the time-skewed kernel's
    common path uses (t < 512).  On a glibc / FMA host both are the C runtime's exp(-t) itself."""
    from oracle import oracle as orc
    from helpers import host_libm_is_glibc_with_fma
    from test_oracle_known_answers import exp_test_arguments
    t = exp_test_arguments()
    w = orc.exp_neg_array(t)
    for small in (False, True):
        g = ctx.exp_neg(t, small_form=small)
        bad = g.view(np.int64) != w.view(np.int64)
        assert not bad.any(), (small, int(bad.sum()), t[bad][:5], g[bad][:5], w[bad][:5])
    if host_libm_is_glibc_with_fma():
        orc.set_exp_mode(1)
        try:
            libm = orc.exp_neg_array(t)
        finally:
            orc.set_exp_mode(0)
        assert np.array_equal(g.view(np.int64), libm.view(np.int64))


def test_hip_unscaled_division_is_the_ieee_division_inside_its_guard(ctx):
    """k_refine_skew's common path divides WITHOUT the hardware sequence's operand scaling and fix-up steps (v_div_scale,
    v_div_fixup), under a guard: 2^-300 < |a| (numerators), denominators in [2^-300, 2^300].  Inside that range the two
    must be the same bits -- held against the device's own a / b AND against numpy's (IEEE round-to-nearest) division on
    4 M operand pairs: random mantissas over the admitted exponent range, the operand shapes of .cpp:669,671 (weights in
    (e^-200, 1], disparities, ws), exact quotients, and ratios one ulp either side of a rounding boundary."""
    rng = np.random.default_rng(20240917)
    n = 1 << 20

    def rnd(lo_exp, hi_exp, size, signed=True):
        m = rng.random(size) + 1.0
        e = rng.integers(lo_exp, hi_exp + 1, size)
        v = np.ldexp(m, e)
        return v * rng.choice([-1.0, 1.0], size) if signed else v

    cases = []
    cases.append((rnd(-299, 299, n), rnd(-299, 299, n, signed=False)))          # the whole admitted range
    wx, wy = np.exp(-rng.random(n) * 200.0), np.exp(-rng.random(n) * 200.0)     # .cpp:669: ds
    dsum1, dsum2 = rng.normal(0, 300, n), rng.normal(0, 300, n)
    cases.append((wx * dsum1 + wy * dsum2, 2 * (wx + wy)))
    pwp, ws = rng.random(n), np.ldexp(1.0, rng.integers(-20, 4, n))              # .cpp:671
    cases.append((rng.normal(0, 200, n) * pwp + ws * rng.normal(0, 200, n), pwp + ws))
    q = rnd(-40, 40, n)                                                          # exact and nearly exact quotients
    b = rnd(-100, 100, n, signed=False)
    a = q * b
    cases.append((np.nextafter(a, rng.choice([-np.inf, np.inf], n)), b))
    for a, b in cases:
        ok = (np.abs(a) > 2.0 ** -300) & (np.abs(a) < 2.0 ** 300) & (b > 2.0 ** -300) & (b < 2.0 ** 300)
        a, b = a[ok], b[ok]
        qf, qi = ctx.div_unscaled(a, b)
        want = a / b
        assert np.array_equal(qi.view(np.int64), want.view(np.int64))            # the device's a / b is IEEE
        bad = qf.view(np.int64) != want.view(np.int64)
        assert not bad.any(), (int(bad.sum()), a[bad][:4], b[bad][:4], qf[bad][:4], want[bad][:4])


def test_the_cloud_filters_trimmed_sqrt_is_sqrtf_on_every_float(ctx):
    """k_filter.hip: sqrtf_rn -- the correctly rounded float32 square root the k-nearest sums take (PCL: sqrt of the float32 squared
    distances, pcl::StatisticalOutlierRemoval behind CCloudOptimization.cpp:25-61) as the compiler's sequence without its denormal
    scaling and zero / infinity test -- against the device's sqrtf on EVERY non-negative float: the 1.88 G patterns from 2^-96 to
    FLT_MAX that take the trimmed sequence, zero, and the patterns below 2^-96, infinity and the NaNs that take sqrtf itself."""
    lo = 0x0f800000                       # 2^-96
    assert ctx.sqrt_check(lo, 0x7f800000 - lo) == 0
    assert ctx.sqrt_check(0, lo) == 0     # zero, the denormals, the small normals
    assert ctx.sqrt_check(0x7f800000, 1 << 16) == 0   # +inf and NaNs (bit-for-bit the same NaN)
    assert ctx.sqrt_check(0x80000000, 1 << 20) == 0   # -0 and negative denormals

