"""Edge cases of the restructured kernels (segment / chunk / component boundaries) against the oracle, through the C ABI.
All integer work: bit-exact."""
import numpy as np
import pytest

from oracle import oracle as orc
from reconstruction_amd import synth

pytestmark = pytest.mark.gpu
NOMATCH = -10000


def _maps(seed, H, W, p_nomatch=0.3, lo=-3, hi=4, p_mask=0.9):
    rng = np.random.default_rng(seed)
    d = rng.integers(lo, hi, size=(H, W)).astype(np.int16)
    d[rng.random((H, W)) < p_nomatch] = NOMATCH
    mask = np.where(rng.random((H, W)) < p_mask, 255, rng.integers(0, 255, size=(H, W))).astype(np.uint8)
    return d, mask


# margin heights around the 32 row segments of the vertical sweeps (empty segments, one-row segments, ragged last one)
# and widths around the 64-column chunks of the horizontal scans
@pytest.mark.parametrize("rows,cols", [(2, 70), (3, 64), (6, 65), (31, 63), (32, 128), (33, 129), (40, 200), (97, 321)])
def test_set_boundary_segment_and_chunk_edges(ctx, rows, cols):
    H, W = rows + 9, cols + 11
    for seed, pn in ((1, 0.2), (2, 0.8), (3, 0.0)):
        d, m = _maps(rows * 1000 + cols + seed, H, W, p_nomatch=pn)
        if seed == 3:
            m[:] = 255          # long unbroken runs: the scans' carries cross every chunk / segment
            d[:] = NOMATCH
            d[H // 2, 5 + cols // 2] = 3
        own = (4, 4 + rows - 1, 5, 5 + cols - 1, cols, rows)
        oth = (3, H - 3, 7, W - 4, W - 10, H - 5)
        s1, BLg, BRg = ctx.set_boundary_smooth(d, m, own, oth)
        s2, BLo, BRo = orc.set_boundary_smooth(d, m, own, oth)
        assert s1 == s2 == 0
        sel = np.zeros((H, W), bool)
        sel[own[0]:own[1] + 1, own[2]:own[3] + 1] = True
        assert np.array_equal(BLg[sel], BLo[sel]), (rows, cols, seed)
        assert np.array_equal(BRg[sel], BRo[sel]), (rows, cols, seed)


@pytest.mark.parametrize("n", [2, 63, 64, 65, 128, 129, 500])
def test_order_constraint_component_shapes(ctx, n):
    """Rows with exactly n valid pixels: sorted (no component), one outlier tying everything together (a component
    longer than the lane-per-component limit), many small components, and a flat run with ties."""
    W, H = n + 40, 12
    rng = np.random.default_rng(n)
    d = np.full((H, W), NOMATCH, np.int16)
    xs = np.arange(10, 10 + n)
    d[2, xs] = 3                                   # m = x + 3: strictly increasing
    d[3, xs] = 3
    d[3, xs[0]] = min(n + 20, 600)                 # one far-reaching outlier at the start
    d[4, xs] = 3
    d[4, xs[-1]] = -min(n + 20, 600)               # ... and one at the end (crosses everything before it)
    d[5, xs] = (rng.integers(0, 2, n) * 3).astype(np.int16)        # small local inversions
    d[6, xs] = (-xs + xs[0]).astype(np.int16)                       # m constant: no strict inversion at all
    d[7, xs] = (-xs + xs[0] + rng.integers(0, 3, n)).astype(np.int16)  # nearly flat: ties and crossings everywhere
    d[8, xs[::2]] = rng.integers(-8, 9, len(xs[::2])).astype(np.int16)  # holes between valid pixels
    own = (1, H - 2, 4, W - 5, W - 8, H - 2)
    a, b = ctx.order_constraint(d, own), orc.order_constraint(d, own)
    assert np.array_equal(a, b), [(y, int((a[y] != b[y]).sum())) for y in range(H) if (a[y] != b[y]).any()]


def test_smooth_and_median_at_margin_borders(ctx):
    for seed in range(3):
        H, W = 37, 151
        d, m = _maps(700 + seed, H, W, p_nomatch=0.25)
        # FindMargin keeps every margin at least MatchBlockRadius >= 1 pixels inside the image (.cpp:1011-1038)
        for own in ((1, H - 2, 1, W - 2, W - 2, H - 2), (2, H - 3, 3, W - 4, W - 6, H - 4), (5, 9, 70, 80, 11, 5)):
            assert np.array_equal(ctx.median_filter(d, m, own), orc.median_filter(d, m, own)), own
            assert np.array_equal(ctx.smooth_constraint(d, own), orc.smooth_constraint(d, own)), own


def test_rematch_sparse_path_random(ctx):
    """Rematch on a map with many holes: the per-row lists, the sparse NCC kernel and intervals of every width."""
    cfg = synth.config_small(160, 96, 2, radius=2, pair=11)
    img, msk = cfg.image, cfg.mask
    r = 2
    H, W = msk[0].shape
    mg0 = orc.find_margin(msk[0], r).astuple()
    mg1 = orc.find_margin(msk[1], r).astuple()
    rng = np.random.default_rng(5)
    for pn in (0.02, 0.3, 0.9):
        d = rng.integers(-6, 7, size=(H, W)).astype(np.int16)
        d[rng.random((H, W)) < pn] = NOMATCH
        so, do = orc.rematch(img[0], img[1], msk[0], msk[1], r, mg0, mg1, d)
        sg, dg = ctx.rematch(img[0], img[1], msk[0], msk[1], r, mg0, mg1, d)
        assert so == sg == 0
        assert np.array_equal(dg, do), (pn, int((dg != do).sum()))


@pytest.mark.parametrize("ksize", [1, 2, 5, 8, 9, 17, 40])
def test_cloud_erosion_with_holes(ctx, ksize):
    """Binary erosion test of DisparityToCloud on a mask with holes and border contact (batches of 8 span rows, the
    clean-block quick accept)."""
    rng = np.random.default_rng(ksize)
    H, W = 90, 140
    m = np.full((H, W), 255, np.uint8)
    m[:3] = 0
    m[:, -2:] = 7
    for _ in range(6):
        y, x = int(rng.integers(0, H)), int(rng.integers(0, W))
        m[y:y + int(rng.integers(1, 9)), x:x + int(rng.integers(1, 9))] = int(rng.integers(0, 255))
    e = orc.erode_ellipse(m, ksize)
    g = ctx.erode_ellipse_is255(m, ksize)
    assert np.array_equal(g == 255, e == 255)


# ---- DisparityRefine: margins around the kernel's 256-column x 4-row workgroups, random textures ------------------
@pytest.mark.parametrize("rows,cols,iters", [(3, 5, 7), (4, 9, 30), (6, 258, 12), (7, 255, 9), (11, 300, 30), (33, 64, 60)])
def test_refine_block_edges(ctx, rows, cols, iters):
    """Margins whose interior is 1 x 3 pixels up to just over one workgroup: every pixel count of the last row group,
    pixels with a single valid neighbour pair (modes 1 / 2), isolated ones (mode 0), both disparity signs.
    fp64, bit for bit (identical operation order, IEEE sqrt / division, the shared specified exp)."""
    H, W = rows + 6, cols + 8
    rng = np.random.default_rng(rows * 100 + cols)
    base = rng.integers(0, 256, size=(H, W + 8, 3)).astype(np.uint8)
    img0 = np.ascontiguousarray(base[:, 4:W + 4])
    img1 = np.ascontiguousarray(base[:, 2:W + 2])      # view 1 = view 0 shifted by 2: a real minimum near d = -2 / +2
    for sign in (+1, -1):
        d = (sign * 2 + rng.integers(-1, 2, size=(H, W))).astype(np.int16)
        d[rng.random((H, W)) < 0.25] = NOMATCH
        own = (3, 3 + rows - 1, 4, 4 + cols - 1, cols, rows)
        a = ctx.disparity_refine(d, img0, img1, iters, 0.03, own)
        b = orc.disparity_refine(d, img0, img1, iters, 0.03, own)
        na, nb = a == NOMATCH, b == NOMATCH
        assert np.array_equal(na, nb)
        assert np.array_equal(a, b), (rows, cols, sign, float(np.abs(a - b).max()))


@pytest.mark.parametrize("skew", [0, 4, 2])
def test_refine_with_weights_across_the_exps_whole_range(ctx, skew):
    """DisparityRefine's smoothness weights exp(-(|dE - dC| - |dW - dC|)^2) (CStereoMatching.cpp:665-666) over the whole range of
    the specified exp INSIDE the sweep kernels: disparity steps of 0 ... 45 pixels between neighbours give arguments from 0 to
    ~2000 -- the table path (t < 512: the time-skewed kernel's common form), glibc's special case for t in [512, 1024) (subnormal
    weights beyond 708), exact zeros beyond 1024 and with them `wx + wy == 0` -> the plain average of :667-668 -- in the single-sweep
    kernels and (skew = T sweeps per launch) in the time-skewed one from the first cached sweep on, where every such pixel fails the
    common row's guards and is redone by the general update.  Bit for bit against the oracle."""
    H, W, iters = 46, 300, 24
    rng = np.random.default_rng(77)
    base = rng.integers(0, 256, size=(H, W + 8, 3)).astype(np.uint8)
    img0 = np.ascontiguousarray(base[:, 4:W + 4])
    img1 = np.ascontiguousarray(base[:, 2:W + 2])
    d = (2 + rng.integers(-1, 2, size=(H, W))).astype(np.int16)
    jump = rng.random((H, W))
    d[jump < 0.10] += rng.integers(18, 30, size=int((jump < 0.10).sum())).astype(np.int16)      # t around 500 ... 900
    d[(jump >= 0.10) & (jump < 0.16)] -= rng.integers(30, 46, size=int(((jump >= 0.10) & (jump < 0.16)).sum())).astype(np.int16)  # t beyond 1024
    d[:, 100:140] += 40                                                                          # a whole band far away: both weights 0 along its edges
    d[rng.random((H, W)) < 0.05] = NOMATCH
    own = (3, H - 4, 4, W - 5, W - 8, H - 6)
    if skew:
        ctx.set_option("refine_skew_T", skew)
        ctx.set_option("refine_skew_from", 1)
        ctx.set_option("refine_skew_min_px", 0)
        ctx.set_option("refine_skew_rows", 16)
    try:
        a = ctx.disparity_refine(d, img0, img1, iters, 0.03, own)
    finally:
        ctx.set_option("refine_skew_T", 4)
        ctx.set_option("refine_skew_from", 22)
        ctx.set_option("refine_skew_min_px", 1000000)
        ctx.set_option("refine_skew_rows", 0)
    b = orc.disparity_refine(d, img0, img1, iters, 0.03, own)
    assert np.array_equal(a == NOMATCH, b == NOMATCH)
    assert np.array_equal(a, b), float(np.abs(a - b).max())
    assert np.isfinite(b[b != NOMATCH]).all() and np.abs(b[b != NOMATCH]).max() < 200


# ---- whole pairs at sizes between the small parity cases and C2 ---------------------------------------------------------
@pytest.mark.parametrize("which", ["c1", "five_levels", "c2_sample"])
def test_whole_pair_at_intermediate_sizes(ctx, which):
    """BASELINE's C1 (640x480, 3 levels), a 5-level pyramid (150 sweeps at the top: the refine transient AND its
    settled phase), and the 2048x1536 quarter-area sample of C2 that bench.py's cpu_baseline times, all against the
    whole-pair oracle: margins, point count, colours and both fp64 disparity maps identical bit for bit, XYZ included."""
    if which == "c1":
        cfg = synth.config_c1()
    elif which == "five_levels":
        cfg = synth.make_pair(512, 384, 5, radius=3, offset=2, pair=21, mask_kind="rect", mask_l0_width=16, holes=True,
                              occlude=True, name="s512x384_5levels")
    else:
        cfg = synth.config_c2_sample()
    ref = orc.match_pair(cfg)
    res = ctx.match_pair(cfg)
    assert res.margin == ref["margin"] and res.v_top == ref["v_top"] and res.n_points == ref["n_points"]
    for v in range(2):
        a, b = res.disparity[v], ref["disparity"][v]
        assert np.array_equal(a, b), (which, v, int((a != b).sum()), float(np.abs(a - b).max()))
    assert np.array_equal(res.bgr, ref["bgr"])
    fin = np.isfinite(ref["xyz"])
    assert np.array_equal(np.isfinite(res.xyz), fin)
    assert np.array_equal(res.xyz[fin], ref["xyz"][fin])
