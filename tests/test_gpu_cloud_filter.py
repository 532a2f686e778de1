"""the per-pair cloud filter (CCloudOptimization::filter, CloudOptimization/CCloudOptimization.cpp:82-121 --
StatisticalOutlierRemoval k = 100 / 1 sigma, radius-2.5 normals, turn toward CamCenter) on the GPU against
oracle/cloud_oracle.c, the brute-force restatement of PCL 1.7.2's published algorithms (parity unpinned: PCL is not in
the reference tree).  Kept indices, the per-point mean distances' statistics: bit-exact.  Normals: 1e-6 (the device's
atan2 / cos / sin differ from the host's in the last bits), sign exact."""
import numpy as np
import pytest
import torch

from oracle import oracle as orc
from reconstruction_amd import synth

pytestmark = pytest.mark.gpu


def surface_cloud(n, seed, extent=60.0, outliers=200, duplicates=50):
    """A bumpy depth-map-like patch (anisotropic sampling as a perspective camera gives it), far and near outliers,
    exact duplicates, an isolated cluster smaller than k."""
    rng = np.random.default_rng(seed)
    u = rng.random((n, 2)) * [extent, 0.7 * extent] - [extent / 2, 0.35 * extent]
    z = 600.0 + 6.0 * np.sin(u[:, 0] / 9.0) * np.cos(u[:, 1] / 7.0) + rng.normal(0, 0.03, n)
    xyz = np.c_[u * (z[:, None] / 600.0), z]
    idx = rng.choice(n, outliers, replace=False)
    xyz[idx] += rng.normal(0, 1.0, (outliers, 3)) * rng.choice([0.5, 3.0, 40.0], (outliers, 1))
    idx = rng.choice(n, duplicates, replace=False)
    xyz[idx] = xyz[rng.choice(n, duplicates)]
    xyz[:30] = [200.0, 150.0, 900.0] + rng.normal(0, 0.2, (30, 3))     # an island of 30 < k points
    xyz = xyz.astype(np.float32)
    xyz[40:44] = [[np.inf, 1.0, 2.0], [np.nan, np.nan, np.nan], [0.0, -np.inf, 5.0], [1.0, 2.0, np.nan]]   # d == 0 gives 1/0 (.cpp:745)
    return xyz


@pytest.mark.parametrize("n,seed,k", [(50000, 1, 100), (6000, 2, 100), (3000, 3, 20)])
def test_sor_and_normals_equal_the_oracle(ctx, n, seed, k):
    xyz = surface_cloud(n, seed)
    cam = np.array([30.0, -20.0, 0.0], np.float32)
    keep_o, dist_o, (mean_o, std_o, thr_o) = orc.sor_filter(xyz, k, 1.0)
    kept, nrm, st = ctx.filter_cloud(xyz, k, 1.0, 2.5, cam)
    assert (st["mean"], st["stddev"], st["threshold"]) == (mean_o, std_o, thr_o)      # same distances, same order of summation
    assert np.array_equal(kept, np.nonzero(keep_o)[0])
    assert 0.5 * n < len(kept) < n and st["exhaustive"] >= 30          # the island needs the exhaustive search
    nrm_o = orc.cloud_normals(xyz[keep_o], 2.5, cam)
    nan_o = np.isnan(nrm_o[:, 0])
    assert np.array_equal(np.isnan(nrm[:, 0]), nan_o)
    assert set(range(40, 44)) <= set(kept.tolist()) and nan_o.sum() >= 4    # non-finite points: kept (distance 0), NaN normal -- as PCL
    ok = ~nan_o
    # well-conditioned neighbourhoods: 1e-6; the cross-product eigenvector of a nearly isotropic neighbourhood (few
    # points) is ill-conditioned in the formula itself
    err = np.abs(nrm[ok, :3] - nrm_o[ok, :3]).max(axis=1)
    assert (err < 1e-6).mean() > 0.999 and err.max() < 1e-3, (float((err < 1e-6).mean()), float(err.max()))
    assert np.abs(nrm[ok, 3] - nrm_o[ok, 3]).max() < 1e-6
    assert np.all(np.einsum("ij,ij->i", nrm[ok, :3], cam[None] - xyz[keep_o][ok]) >= 0)   # turned toward CamCenter


def test_filter_of_a_matched_pair_on_device_equals_the_host_entry(ctx):
    """rsm_filter_last_cloud (the cloud never leaves the GPU; output = the RCCL payload) against rsm_filter_cloud on the
    downloaded cloud, and the reference-shaped CloudOptimization sink behind StereoMatching.MatchAllLayer."""
    from reconstruction_amd import Camera, CloudOptimization, ManageData, StereoMatching
    from reconstruction_amd.dist import unpack_records
    cfg = synth.config_small(320, 192, 3, radius=2, pair=4, mask_l0_width=60, border_l0=4)
    res = ctx.match_pair(cfg)
    assert res.n_points > 20000
    cam = np.array([0.0, 0.0, 0.0], np.float32)
    # the synthetic rig's units: ~0.4 per pixel at Z ~ 1000; a radius of 2.5 holds ~100 points
    kept, nrm, st = ctx.filter_cloud(res.xyz, 100, 1.0, 2.5, cam)
    rec = torch.empty((res.n_points, 16), dtype=torch.uint8, device="cuda:0")
    nd = torch.empty((res.n_points, 4), dtype=torch.float32, device="cuda:0")
    m, st2 = ctx.filter_last_cloud(rec.data_ptr(), nd.data_ptr(), res.n_points, 100, 1.0, 2.5, cam)
    # (the statistics' bits; "exhaustive" -- how many queries ended in the whole-cloud search -- is a diagnostic of the route taken:
    # the device entry decides most queries on the pixel lattice and starts its grid ladder elsewhere)
    assert m == len(kept) and 0 < m < res.n_points and all(st2[k] == st[k] for k in ("mean", "stddev", "threshold"))
    xyz16, bgr16 = unpack_records(rec[:m])
    assert np.array_equal(xyz16, res.xyz[kept].astype(np.float32)) and np.array_equal(bgr16, res.bgr[kept])
    assert np.array_equal(nd[:m].cpu().numpy(), nrm, equal_nan=True)
    # reference-shaped sink
    top = 1 << (cfg.pyr_levels - 1)
    data = ManageData(cam=[[Camera(camID=0, image=cfg.image[0], mask=cfg.mask[0], CamCenter=cam),
                            Camera(camID=1, image=cfg.image[1], mask=cfg.mask[1], CamCenter=cam)]],
                      m_PyrmNum=cfg.pyr_levels, m_LowestLevelSize=(cfg.width // top, cfg.height // top),
                      m_OriginSize=(cfg.width, cfg.height), rectified=[dict(Q=cfg.Q, R_final=cfg.R_final, T_final=cfg.T_final)])
    opt = CloudOptimization(ctx)
    opt.Init(100, 1, 50, 2, 2.5, data, False)        # CReconstruction.cpp:18
    sm = StereoMatching(0)
    sm.Init(data, opt, 2, 0.03)                       # CReconstruction.cpp:17
    sm.Verbose = 0
    sm.MatchAllLayer()
    fx, fn = opt.cloud_normals[0]
    assert np.array_equal(fx, res.xyz[kept].astype(np.float32)) and np.array_equal(fn, nrm, equal_nan=True)


PAIRS_W = [dict(width=320, height=192, levels=3, radius=2, pair=4, mask_l0_width=60, border_l0=4),
           dict(width=512, height=384, levels=5, radius=3, offset=2, pair=21, mask_l0_width=16, holes=True, occlude=True),
           dict(width=384, height=256, levels=3, radius=2, pair=2, mask_kind="ellipse"),
           dict(width=320, height=160, levels=2, radius=4, pair=7, mask_l0_width=100, border_l0=6, occlude=True),
           dict(width=256, height=192, levels=3, radius=2, pair=5, mask_l0_width=40, border_l0=4, holes=True),
           # large disparities: f / d = 6, the depth resolution of a close-range rig like the reference's -- a THIN sheet
           dict(width=512, height=256, levels=3, radius=2, pair=6, mask_l0_width=70, border_l0=4, d0_l0=25.0, amp_l0=2.0, holes=True)]


def _filter_sig(ctx, k, cam, window):
    ctx.set_option("filter_window", window)
    try:
        rec, nrm, st = ctx.filter_last_cloud_host(k, 1.0, 2.5, cam)
        info = ctx.filter_last_info()
    finally:
        ctx.set_option("filter_window", 1)
    return (rec.tobytes(), nrm.tobytes(), (st["mean"], st["stddev"], st["threshold"]), len(rec)), info


@pytest.mark.parametrize("case", range(len(PAIRS_W)))
@pytest.mark.parametrize("k", [100, 30])
def test_pixel_window_pass_gives_the_generic_searchs_results(ctx, case, k):
    """rsm_filter_last_cloud's pixel-window k-nearest pass (round 5: the cloud of a matched pair is a depth map, so the k
    nearest of a point lie inside a pixel window around its own pixel whenever the (k+1)-th smallest distance found there is
    below the distance to every ray outside the window -- decided per point, a query per lane; the rest goes to the grid
    ladder) against the same call with the pass switched off (every query through the generic grid search): the same
    surviving points, the same normals, the same statistics, to the bit -- for the probed radius and for every instantiated
    one (7, 12, 16, 20, 24), with every combination of the passes behind the tile pass (`filter_list`: the 49 x 49 and 81 x 81
    list passes a thread per query, the wave passes over 80 ... pixels, the whole-chip search over the lattice copy), on rectangular and elliptic masks, holes, occlusions (depth edges), both disparity signs,
    rotated rigs (R_final != I for pair > 0), thick sheets (small disparities: depth noise of many pixel spacings) and a thin
    one (large disparities), k = 100 and 30."""
    cfg = synth.config_small(**PAIRS_W[case])
    res = ctx.match_pair(cfg)
    assert res.n_points > 5000
    cam = (3.0 * case, -2.0, 1.0)
    want, info0 = _filter_sig(ctx, k, cam, 0)
    assert info0["radius"] == 0 and info0["points"] == res.n_points and info0["kept"] == want[3] and 0 < want[3] < res.n_points
    ctx.set_option("filter_list", 0)   # the tile pass alone in front of the ladder
    try:
        got, info = _filter_sig(ctx, k, cam, 12)
    finally:
        ctx.set_option("filter_list", 23)
    assert got == want and info["radius"] == 12
    for fl in (1, 2, 3, 5, 7, 15, 31):   # the passes behind the tile pass one by one, thread form and wave form, with and without the wave passes at 80 ... pixels (23 = the default, below)
        ctx.set_option("filter_list", fl)
        try:
            got, info = _filter_sig(ctx, k, cam, 16)
        finally:
            ctx.set_option("filter_list", 23)
        assert got == want, (case, k, "filter_list", fl)
    for window in (1, 7, 12, 16, 20, 24):   # (default: what the tile pass leaves over gets the 24-pixel window a thread each)
        got, info = _filter_sig(ctx, k, cam, window)
        print("case %d k %d window %d: %d points, radius %d, %d (%.1f %%) left to the ladder" % (case, k, window, res.n_points, info["radius"], info["undecided"],
                                                                                     100.0 * info["undecided"] / res.n_points))
        assert got == want, (case, k, window)
        if window > 1:
            assert info["radius"] == window
    # the probed radius is remembered by the context (no probe launches on the following calls, a fresh probe every 8th and after
    # any set_option): ten calls in a row, the same bits every time
    ctx.set_option("filter_window", 1)
    for rep in range(10):
        rec, nrm, st = ctx.filter_last_cloud_host(k, 1.0, 2.5, cam)
        assert (rec.tobytes(), nrm.tobytes(), (st["mean"], st["stddev"], st["threshold"]), len(rec)) == want, (case, k, "repeat", rep)
    thin = "d0_l0" in PAIRS_W[case]
    if thin and k == 100:   # the thin sheet: the probe settles for the small window, which decides nearly everything
        got, info = _filter_sig(ctx, k, cam, 1)
        assert info["radius"] == 7 and info["undecided"] < 0.15 * res.n_points, info


@pytest.mark.parametrize("case", range(len(PAIRS_W)))
@pytest.mark.parametrize("radius", [2.5, 9.0])
def test_normals_on_the_pixel_lattice_equal_the_grid_normals(ctx, case, radius):
    """Round 6: the radius search of the normals runs over a pixel window of the lattice copy (the removed points blanked) whenever
    the bound that decides the k-nearest windows says a window of at most `filter_normals_window` pixels holds every point within the
    radius; otherwise, and for a generic cloud, over a grid of radius-cells.  Both orders of summation on the same neighbourhoods:
    the same NaN pattern (fewer than 3 neighbours), the same normals to 1e-6 (the nine sums are exact in double for the few dozen
    float products of a neighbourhood, so in practice the same bits) -- for the reference's radius 2.5 (a pixel or two on these rigs,
    often less than one: no normals at all) and for a wider one (windows up to the lattice's border)."""
    cfg = synth.config_small(**PAIRS_W[case])
    res = ctx.match_pair(cfg)
    cam = (3.0 * case, -2.0, 1.0)
    out = {}
    try:
        for wmax in (0, 8, 40):
            ctx.set_option("filter_normals_window", wmax)
            rec, nrm, st = ctx.filter_last_cloud_host(100, 1.0, radius, cam)
            out[wmax] = (rec.tobytes(), nrm.copy(), ctx.filter_last_info())
    finally:
        ctx.set_option("filter_normals_window", 8)
    info0, info8, info40 = out[0][2], out[8][2], out[40][2]
    need = info40["normals_need"]
    print("case %d radius %.1f: widest window a normal needs %d pixels; used %d / %d / %d" % (case, radius, need, info0["normals_window"], info8["normals_window"], info40["normals_window"]))
    assert info0["normals_window"] == 0 and need >= -1   # (-1: a grid level of the k-nearest search took the lattice copy's arena space)
    assert info8["normals_window"] == (max(need, 1) if 0 <= need <= 8 else 0) and info40["normals_window"] == (max(need, 1) if 0 <= need <= 40 else 0)
    for wmax in (8, 40):
        assert out[wmax][0] == out[0][0]                                   # the same surviving points
        a, b = out[wmax][1], out[0][1]
        assert np.array_equal(np.isnan(a), np.isnan(b))
        ok = ~np.isnan(b[:, 0])
        print("   window limit %d: %d of %d points have a normal, largest difference %.1e" % (wmax, ok.sum(), len(b), np.abs(a[ok] - b[ok]).max() if ok.any() else 0.0))
        assert not ok.any() or np.abs(a[ok] - b[ok]).max() < 1e-6, (case, wmax)


@pytest.mark.parametrize("case", range(len(PAIRS_W)))
def test_wave_and_workgroup_forms_of_the_wide_passes_agree(ctx, case):
    """The passes behind the tile and list passes take a wave per listed query (k_sor_window_wave) or, once few queries are left, four
    waves per query with a longer selection list (k_sor_window_wg, round 6): never (`filter_wg_max` 0), by the default rule, always --
    with the list passes on, off, and in the wave form themselves (so that thousands of queries reach either form): the same bits."""
    ctx.match_pair(synth.config_small(**PAIRS_W[case]))
    cam = (1.0, 2.0 * case, -1.0)
    want, _ = _filter_sig(ctx, 100, cam, 0)
    try:
        for fl in (23, 4, 28, 31):
            ctx.set_option("filter_list", fl)
            for wg_max in (0, 2048, 1 << 30):
                ctx.set_option("filter_wg_max", wg_max)
                got, info = _filter_sig(ctx, 100, cam, 12)
                assert got == want, (case, fl, wg_max)
    finally:
        ctx.set_option("filter_list", 23)
        ctx.set_option("filter_wg_max", 2048)


def test_some_test_pair_takes_the_lattice_normals(ctx):
    """... and the window form is what the default settings run on at least one of the test rigs (the others need wider windows
    than the default 8 pixels: close-range rigs whose pixel spacing is a small fraction of the radius)."""
    used = []
    for case in range(len(PAIRS_W)):
        ctx.match_pair(synth.config_small(**PAIRS_W[case]))
        ctx.filter_last_cloud_host(100, 1.0, 2.5, (0.0, 0.0, 0.0))
        used.append(ctx.filter_last_info()["normals_window"])
    print("normals windows of the test pairs:", used)
    assert any(w > 0 for w in used), used


def test_pixel_window_pass_against_the_brute_force_oracle(ctx):
    """... and against oracle/cloud_oracle.c (PCL's statistical outlier removal restated by brute force) on a pair small enough
    for O(n^2): the kept set and the statistics' bits."""
    cfg = synth.config_small(width=256, height=192, levels=3, radius=2, pair=5, mask_l0_width=40, border_l0=4, holes=True)
    res = ctx.match_pair(cfg)
    xyz = res.xyz.astype(np.float32)
    keep_o, dist_o, (mean_o, std_o, thr_o) = orc.sor_filter(xyz, 100, 1.0)
    rec, nrm, st = ctx.filter_last_cloud_host(100, 1.0, 2.5, (0.0, 0.0, 0.0))
    assert ctx.filter_last_info()["window"]
    assert (st["mean"], st["stddev"], st["threshold"]) == (mean_o, std_o, thr_o)
    want = xyz[keep_o]
    assert len(rec) == len(want) and np.array_equal(np.stack([rec["x"], rec["y"], rec["z"]], 1), want, equal_nan=True)


@pytest.mark.parametrize("n,k", [(150, 100), (101, 100), (50, 100), (7, 100), (2, 5), (1, 3)])
def test_tiny_clouds_follow_the_oracle(ctx, n, k):
    """Clouds around and below k + 1 points (every query ends in the whole-cloud search; with fewer than k + 1 points the
    sum runs over the neighbours that exist, as oracle/cloud_oracle.c states PCL's loop), a non-finite point among them,
    a single point (PCL's variance divides by n - 1 = 0: the NaN threshold removes nothing)."""
    rng = np.random.default_rng(100 + n)
    xyz = (rng.normal(0, 3.0, (n, 3)) + [0, 0, 500]).astype(np.float32)
    if n >= 7:
        xyz[3] = [np.nan, 1.0, 2.0]
    cam = np.array([1.0, 2.0, 0.0], np.float32)
    keep_o, dist_o, (mean_o, std_o, thr_o) = orc.sor_filter(xyz, k, 1.0)
    kept, nrm, st = ctx.filter_cloud(xyz, k, 1.0, 2.5, cam)
    for a, b in ((st["mean"], mean_o), (st["stddev"], std_o), (st["threshold"], thr_o)):
        assert a == b or (np.isnan(a) and np.isnan(b)), (st, mean_o, std_o, thr_o)
    assert np.array_equal(kept, np.nonzero(keep_o)[0])
    nrm_o = orc.cloud_normals(xyz[keep_o], 2.5, cam)
    assert np.array_equal(np.isnan(nrm[:, 0]), np.isnan(nrm_o[:, 0]))
    ok = ~np.isnan(nrm_o[:, 0])
    if ok.any():
        assert np.abs(nrm[ok, :3] - nrm_o[ok, :3]).max() < 1e-3


def test_filter_of_an_empty_and_an_all_nonfinite_cloud(ctx):
    kept, nrm, st = ctx.filter_cloud(np.zeros((0, 3), np.float32), 100, 1.0, 2.5, (0, 0, 0))
    assert len(kept) == 0 and len(nrm) == 0
    xyz = np.full((9, 3), np.nan, np.float32)
    kept, nrm, st = ctx.filter_cloud(xyz, 100, 1.0, 2.5, (0, 0, 0))
    assert list(kept) == list(range(9)) and np.isnan(nrm).all()      # distance 0 for every point, nothing exceeds a NaN threshold
