"""NCC matchers on low-entropy images: exact and near ties between candidates.

The reference scores a candidate in fp64 with a fixed summation order (CManageData.cpp:81-90 WindowToVec,
op_dot_meat.hpp:20-55, CStereoMatching.cpp:207-218); when two candidates tie mathematically, its last-bit rounding
decides which one the strict '>' scan keeps.  The kernels filter with exact integer window sums and hand every
pixel that sees a (near) tie to a reference-order fp64 re-evaluation (k_match.hip: k_ncc_exact), so the chosen
COLUMN stays bit-exact on few-grey-level, saturated and periodic textures too -- the inputs on which
band-limited 8-bit noise (every other parity case) can never produce a tie."""
import numpy as np
import pytest

from oracle import oracle as orc
from reconstruction_amd import synth

from helpers import NOMATCH, diff_report, oracle_stages

pytestmark = pytest.mark.gpu


def _texture(kind, W, H, rng):
    if kind == "2level":       # two grey levels, iid per pixel (all channels equal)
        g = rng.choice(np.array([60, 190], np.uint8), size=(H, W))
        return np.repeat(g[:, :, None], 3, axis=2)
    if kind == "2level_rgb":   # two levels, iid per channel
        return rng.choice(np.array([0, 255], np.uint8), size=(H, W, 3))
    if kind == "3level":
        return rng.choice(np.array([10, 128, 250], np.uint8), size=(H, W, 3))
    if kind == "saturated":    # 8-bit noise with saturated (flat 0 / 255) blocks: zero-variance windows
        a = rng.integers(0, 256, size=(H, W, 3)).astype(np.uint8)
        for _ in range(10):
            y, x = int(rng.integers(0, H - 12)), int(rng.integers(0, W - 24))
            a[y:y + int(rng.integers(6, 12)), x:x + int(rng.integers(10, 24))] = 255 if rng.random() < 0.5 else 0
        return a
    if kind == "periodic":     # period 7 columns < search range: candidates one period apart tie exactly
        t = rng.integers(0, 256, size=(H, 7, 3)).astype(np.uint8)
        return np.tile(t, (1, (W + 6) // 7, 1))[:, :W]
    if kind == "periodic2d":   # doubly periodic 2-level pattern
        t = rng.choice(np.array([30, 220], np.uint8), size=(5, 6, 3))
        return np.tile(t, ((H + 4) // 5, (W + 5) // 6, 1))[:H, :W]
    raise ValueError(kind)


def make_case(kind, W, H, levels, radius, seed, shift, offset=2, flips=0):
    rng = np.random.default_rng(seed)
    top = 1 << (levels - 1)
    img0 = _texture(kind, W, H, rng)
    img1 = np.roll(img0, shift, axis=1).copy()          # exact integer shift: view 1 holds bit-identical windows
    for _ in range(flips):                               # a few changed pixels so that not every pixel ties
        y, x = int(rng.integers(0, H)), int(rng.integers(0, W))
        img1[y, x] = img0[int(rng.integers(0, H)), int(rng.integers(0, W))]
    m0 = np.zeros((H, W), np.uint8)
    b = (radius + 2) * top
    m0[b:H - b, b + abs(shift):W - b - abs(shift)] = 255
    m1 = np.roll(m0, shift, axis=1)
    Q, R, T = synth.pinhole_calibration(W, H, 0)
    return synth.PairConfig(width=W, height=H, pyr_levels=levels, radius=radius, ws=0.03, offset=offset, origin_width=W,
                            image=[img0, img1], mask=[m0, m1], Q=Q, R_final=R, T_final=T,
                            name="%s_%dx%d_r%d_s%d" % (kind, W, H, radius, seed))


CASES = [
    # kind, W, H, levels, radius, seed, shift
    ("2level", 80, 28, 1, 2, 1, 3),
    ("2level", 80, 28, 1, 2, 2, -4),
    ("2level", 96, 40, 1, 5, 3, 5),
    ("2level_rgb", 88, 30, 1, 2, 4, 2),
    ("3level", 80, 28, 1, 2, 5, -3),
    ("3level", 100, 44, 1, 5, 6, 4),
    ("saturated", 120, 48, 1, 2, 7, 6),
    ("saturated", 120, 56, 1, 5, 8, -5),
    ("periodic", 128, 36, 1, 2, 9, 3),
    ("periodic", 140, 48, 1, 5, 10, -2),
    ("periodic2d", 96, 40, 1, 2, 11, 4),
    # intervals wider than NCC_WIDE = 160 candidates: the one-workgroup-per-pixel kernel and its cross-lane reduction
    ("2level", 420, 40, 1, 2, 18, 5),
    ("periodic", 420, 36, 1, 3, 19, -3),
    ("3level", 440, 44, 1, 7, 20, 4),       # radius 7: 78 KB of dynamic LDS in k_ncc_wide
    ("2level", 150, 56, 1, 8, 21, 3),       # radius 8: the generic byte-wise kernel
    # two levels: the top level (the few-level texture itself) goes through HighLevelInitialMatch + Rematch
    ("2level", 160, 64, 2, 2, 12, 4),
    ("2level_rgb", 160, 64, 2, 2, 13, -6),
    ("3level", 176, 80, 2, 5, 14, 6),
    ("saturated", 192, 96, 2, 2, 15, 8),
    ("periodic", 192, 72, 2, 2, 16, -4),
    ("periodic2d", 192, 96, 3, 2, 17, 8),
]


@pytest.mark.parametrize("case", CASES, ids=lambda c: "%s_%dx%d_L%d_r%d_s%d" % c[:6])
def test_ncc_matchers_on_low_entropy_textures(ctx, case):
    kind, W, H, levels, radius, seed, shift = case
    cfg = make_case(kind, W, H, levels, radius, seed, shift, flips=W * H // 50)
    rec, fin = oracle_stages(cfg)
    imgs, msks = fin["imgs"], fin["msks"]
    r, off = cfg.radius, cfg.offset
    fails = []
    n_initial = n_rematch = 0
    for q in rec:
        k, st = q["level"], q["stage"]
        mg = q["mg"]
        tag = "%s L%d %s v%s" % (cfg.name, k, st, q.get("v"))
        if st == "initial":
            v = q["v"]; o = 1 - v
            g = ctx.initial_match(imgs[k][v], imgs[k][o], msks[k][v], msks[k][o], r, off, mg[v], mg[o], q["parent"])
            n_initial += int((q["out"] != NOMATCH).sum())
            if not np.array_equal(g, q["out"]):
                fails.append(diff_report(tag, g, q["out"]))
        elif st == "rematch":
            v = q["v"]; o = 1 - v
            s, g = ctx.rematch(imgs[k][v], imgs[k][o], msks[k][v], msks[k][o], r, mg[v], mg[o], q["inp"])
            assert s == 0
            n_rematch += int(((q["inp"] == NOMATCH) & (q["out"] != NOMATCH)).sum())
            if not np.array_equal(g, q["out"]):
                fails.append(diff_report(tag, g, q["out"]))
    assert n_initial > 0
    assert not fails, "\n".join(fails[:8])
    # and the whole pair (every stage downstream of a flipped column would differ)
    ref = orc.match_pair(cfg)
    res = ctx.match_pair(cfg)
    for v in range(2):
        assert np.array_equal(res.disparity[v], ref["disparity"][v]), diff_report("pair d%d" % v, res.disparity[v], ref["disparity"][v])
    assert res.n_points == ref["n_points"]


def test_perfectly_anticorrelated_candidate_follows_the_reference(ctx):
    """score == -1 mathematically (support window = 255 - reference window): the reference accepts it only if its
    rounded fp64 score exceeds the initial -1 (CStereoMatching.cpp:205,213)."""
    rng = np.random.default_rng(3)
    W, H, r = 72, 24, 2
    img0 = rng.choice(np.array([0, 255], np.uint8), size=(H, W, 3))
    img1 = 255 - img0
    m = np.zeros((H, W), np.uint8)
    m[4:H - 4, 20:28] = 255
    m1 = np.zeros_like(m)
    m1[4:H - 4, 20:21] = 255       # one candidate column only
    mg0, mg1 = orc.find_margin(m, r).astuple(), orc.find_margin(m1, r).astuple()
    want = orc.lowest_level_initial_match(img0, img1, m, m1, r, mg0, mg1)
    got = ctx.initial_match(img0, img1, m, m1, r, 2, mg0, mg1, None)
    assert np.array_equal(got, want), diff_report("anticorrelated", got, want)


WIDE_CASES = [c for c in CASES if c[1] >= 420] + [("saturated", 460, 40, 1, 5, 31, -6), ("2level_rgb", 700, 36, 1, 2, 32, 7)]


@pytest.mark.parametrize("wide_rows", [1, 2, 3], ids=["workgroup_per_pixel", "int8_row_gemm", "sliding_window_sums"])
@pytest.mark.parametrize("case", WIDE_CASES, ids=lambda c: "%s_%dx%d_r%d" % (c[0], c[1], c[2], c[4]))
def test_every_wide_row_kernel_picks_the_reference_columns(ctx, case, wide_rows):
    """Rows whose pixels scan more than NCC_WIDE candidates have three interchangeable kernels (option wide_rows: the
    workgroup-per-pixel scan, the int8 row GEMM on the matrix cores, the sliding window sums); on tie-ridden textures
    each of them must hand the same pixels to the reference-order re-evaluation and end on the oracle's columns."""
    kind, W, H, levels, radius, seed, shift = case
    cfg = make_case(kind, W, H, levels, radius, seed, shift, flips=W * H // 50)
    mask = [np.ascontiguousarray(m) for m in cfg.mask]
    mg = [orc.find_margin(mask[v], radius).astuple() for v in range(2)]
    ctx.set_option("wide_rows", wide_rows)
    try:
        for v in range(2):
            want = orc.lowest_level_initial_match(cfg.image[v], cfg.image[1 - v], mask[v], mask[1 - v], radius, mg[v], mg[1 - v])
            got = ctx.initial_match(cfg.image[v], cfg.image[1 - v], mask[v], mask[1 - v], radius, 2, mg[v], mg[1 - v], None)
            assert (mg[1 - v][3] - mg[1 - v][2] + 1) > 160          # every pixel is a wide pixel
            assert np.array_equal(got, want), diff_report("wide_rows=%d v%d" % (wide_rows, v), got, want)
    finally:
        ctx.set_option("wide_rows", 0)
