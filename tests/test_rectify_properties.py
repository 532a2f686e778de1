"""Rectify (CStereoMatching.cpp:117-168) held to what the GEOMETRY demands, not to a second copy of itself.

OpenCV's source is absent from the reference tree, so the product's host routine (csrc/rectify_host.cpp, through the C ABI:
rsm_stereo_rectify) and the oracle's (oracle/rectify_oracle.c) are both the builder's transcription of cv::stereoRectify
(flags = 0, alpha = -1): comparing them bit for bit (tests/test_rectify_cpu.py) guards against regressions, it cannot say
either is RIGHT.  These tests can: for random calibrations and poses BOTH implementations must produce a rectification in
which (1) the new rotations are rotations, (2) any 3-D point lands on the same row of both rectified images to < 1e-9
pixels, (3) the rectified baseline lies along x and P2(0,3) = f * Tx_new -- the sign convention the reference's flip of
Q(3,2) at :138 relies on --, (4) Q reprojects (u, v, u1 - u2) to the point's rectified coordinates to < 1e-9 relative, and
(5) the whole Rectify contract :132-145 closes: with the level scale applied, Q's sign flipped and the reference's
disparity convention (x_other - x), DisparityToCloud's formula (:744-749) followed by R_final * X + T_final returns the
WORLD point, and cam[j].P = scaled P * Extrinsic_final projects world points straight onto the scaled rectified image j.
The row stays "partial" in the coverage table whatever these say: no OpenCV binary exists here to pin the arithmetic
(the projection of the image corners through float32, the rounding of the new principal point) bit for bit."""
import numpy as np
import pytest

from oracle import oracle as orc
from reconstruction_amd import stereo_rectify as product_stereo_rectify

IMPLS = [("oracle", orc.stereo_rectify), ("product", product_stereo_rectify)]


def _rot(rx, ry, rz):
    cx, sx, cy, sy, cz, sz = np.cos(rx), np.sin(rx), np.cos(ry), np.sin(ry), np.cos(rz), np.sin(rz)
    Rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
    Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    return Rz @ Ry @ Rx


def random_rig(rng):
    """Two pinhole cameras looking roughly the same way, the second one mostly to one side (either side) of the first:
    K[2] 3x3, E[2] 3x4 = [R | t] with x_cam = R X_world + t, the image size."""
    ow, oh = int(rng.integers(400, 4200)), int(rng.integers(300, 3200))
    f = rng.uniform(0.8, 2.0) * ow
    K = [np.array([[f * rng.uniform(0.97, 1.03), 0, ow / 2 + rng.uniform(-30, 30)],
                   [0, f * rng.uniform(0.97, 1.03), oh / 2 + rng.uniform(-30, 30)], [0, 0, 1.0]]) for _ in range(2)]
    R0 = _rot(*rng.uniform(-0.5, 0.5, 3))
    R1 = _rot(*rng.uniform(-0.12, 0.12, 3)) @ R0
    C0 = rng.uniform(-500, 500, 3)
    side = 1.0 if rng.random() < 0.5 else -1.0
    C1 = C0 + R0.T @ np.array([side * rng.uniform(40, 400), rng.uniform(-15, 15), rng.uniform(-15, 15)])
    E = [np.hstack([R0, (-R0 @ C0)[:, None]]), np.hstack([R1, (-R1 @ C1)[:, None]])]
    return K, E, (ow, oh)


def relative_pose(E):
    R = E[1][:, :3] @ E[0][:, :3].T                  # .cpp:125
    T = -R @ E[0][:, 3] + E[1][:, 3]                 # .cpp:126
    return R, T


def world_points(rng, E, n=12):
    """points in front of both cameras, 3 to 40 baselines away"""
    R0, t0 = E[0][:, :3], E[0][:, 3]
    base = np.linalg.norm(relative_pose(E)[1])
    pts = []
    for _ in range(n):
        Xc0 = np.array([rng.uniform(-0.3, 0.3), rng.uniform(-0.3, 0.3), 1.0]) * rng.uniform(3, 40) * base
        pts.append(R0.T @ (Xc0 - t0))
    return pts


@pytest.mark.parametrize("name,impl", IMPLS)
def test_rectification_is_a_rectification_for_random_poses(name, impl):
    rng = np.random.default_rng(2026)
    worst_row, worst_q = 0.0, 0.0
    for trial in range(60):
        K, E, size = random_rig(rng)
        R, T = relative_pose(E)
        R1, R2, P1, P2, Q = impl(K[0], K[1], size, R, T)
        # (1) rotations
        for Rn in (R1, R2):
            assert np.abs(Rn @ Rn.T - np.eye(3)).max() < 1e-13 and abs(np.linalg.det(Rn) - 1) < 1e-13, (name, trial)
        # (3) one focal length, one row of principal points, no skew; the baseline along x, P2(0,3) = f * Tx_new
        f = P1[0, 0]
        assert P1[1, 1] == f and P2[0, 0] == f and P2[1, 1] == f and P1[1, 2] == P2[1, 2], (name, trial)
        assert P1[0, 1] == 0 and P2[0, 1] == 0 and np.all(P1[:, 3] == 0) and P2[1, 3] == 0 and P2[2, 3] == 0
        t_new = R2 @ T
        assert abs(t_new[1]) < 1e-12 * abs(t_new[0]) and abs(t_new[2]) < 1e-12 * abs(t_new[0]), (name, trial, t_new)
        assert abs(P2[0, 3] - f * t_new[0]) < 1e-12 * abs(P2[0, 3]), (name, trial)
        # Q as cv::stereoRectify leaves it (before the reference's flip): Q(3,2) = -1 / Tx_new, Q(3,3) = (cx1 - cx2) / Tx_new
        assert abs(Q[3, 2] + 1.0 / t_new[0]) < 1e-12 * abs(Q[3, 2])
        assert abs(Q[3, 3] - (P1[0, 2] - P2[0, 2]) / t_new[0]) < 1e-12 * max(abs(Q[3, 3]), 1e-300) + 1e-18
        assert Q[0, 3] == -P1[0, 2] and Q[1, 3] == -P1[1, 2] and Q[2, 3] == f
        for Xw in world_points(rng, E):
            Xc0 = E[0][:, :3] @ Xw + E[0][:, 3]
            Xc1 = E[1][:, :3] @ Xw + E[1][:, 3]
            assert abs(np.abs(R @ Xc0 + T - Xc1).max()) < 1e-9 * np.abs(Xc1).max()        # (the test's own pose algebra)
            Xr = R1 @ Xc0                                                                  # rectified camera-1 coordinates
            p1 = P1 @ np.append(Xr, 1.0)
            p2 = P2[:, :3] @ (R2 @ Xc1)
            u1, v1, u2, v2 = p1[0] / p1[2], p1[1] / p1[2], p2[0] / p2[2], p2[1] / p2[2]
            # (2) epipolar lines are rows
            worst_row = max(worst_row, abs(v1 - v2))
            assert abs(v1 - v2) < 1e-9, (name, trial, v1, v2)
            # ... and P2 (whose fourth column speaks camera-1 coordinates) agrees with the second camera's own projection
            p2b = P2 @ np.append(Xr, 1.0)
            assert abs(p2b[0] / p2b[2] - u2) < 1e-9 * max(1.0, abs(u2))
            # (4) Q reprojects pixel + disparity to the rectified point
            h = Q @ np.array([u1, v1, u1 - u2, 1.0])
            err = np.abs(h[:3] / h[3] - Xr).max() / np.abs(Xr).max()
            worst_q = max(worst_q, err)
            assert err < 1e-9, (name, trial, err)
    print("%s: worst row misalignment %.2e px, worst Q reprojection %.2e relative" % (name, worst_row, worst_q))


def rectify_contract(K, E, size, lowest_w, levels, impl):
    """CStereoMatching::Rectify's outputs (:125-145) from a stereo_rectify implementation: Q with the sign flip, R_final,
    T_final, cam[j].P = (rows 0..1 scaled) P_j * Extrinsic_final, the level scale."""
    R, T = relative_pose(E)
    R1, R2, P1, P2, Q = impl(K[0], K[1], size, R, T)
    R0, t0 = E[0][:, :3], E[0][:, 3]
    R_final = R0.T @ R1.T                  # :132
    T_final = -R0.T @ t0                   # :133
    Ext = np.zeros((4, 4))
    Ext[3, 3] = 1
    Ext[:3, :3] = R_final.T
    Ext[:3, 3] = -R_final.T @ T_final
    Q = Q.copy()
    Q[3, 2] = -Q[3, 2]                     # :138
    scale = float(lowest_w) / size[0] * (1 << (levels - 1))   # :140
    P = []
    for Pj in (P1, P2):
        Pj = Pj.copy()
        Pj[:2] *= scale                    # :143
        P.append(Pj @ Ext)                 # :145
    return Q, R_final, T_final, P, scale


@pytest.mark.parametrize("name,impl", IMPLS)
def test_the_whole_rectify_contract_returns_world_points(name, impl):
    """Project a world point with cam[0].P and cam[1].P (what Rectify leaves for CloudOptimization, :145), form the
    reference's disparity x_other - x at the working resolution, reproject with DisparityToCloud's formula (:697-699, :744-749:
    Q's fourth column times the scale, the flipped Q(3,2)) and apply R_final, T_final (:749): the world point, to 1e-9."""
    rng = np.random.default_rng(77)
    worst = 0.0
    for trial in range(40):
        K, E, size = random_rig(rng)
        levels = int(rng.integers(2, 6))
        lowest_w = max(16, int(size[0] * rng.uniform(0.4, 1.0)) >> (levels - 1))
        Q, R_final, T_final, P, scale = rectify_contract(K, E, size, lowest_w, levels, impl)
        assert np.abs(R_final @ R_final.T - np.eye(3)).max() < 1e-13
        q = Q.copy()
        q[:, 3] *= scale                                   # :697-699
        for Xw in world_points(rng, E):
            p0 = P[0] @ np.append(Xw, 1.0)
            p1 = P[1] @ np.append(Xw, 1.0)
            x0, y0, x1, y1 = p0[0] / p0[2], p0[1] / p0[2], p1[0] / p1[2], p1[1] / p1[2]
            assert abs(y0 - y1) < 1e-9 * max(1.0, abs(y0)), (name, trial)
            d = x1 - x0                                    # the reference's disparity: x_match_in_other - x
            iW = 1.0 / (q[3, 3] + q[3, 2] * d)             # :744
            F = np.array([(q[0, 3] + x0) * iW, (y0 + q[1, 3]) * iW, q[2, 3] * iW])   # :745-747
            back = R_final @ F + T_final                   # :749
            err = np.abs(back - Xw).max() / max(np.abs(Xw).max(), np.linalg.norm(relative_pose(E)[1]))
            worst = max(worst, err)
            assert err < 1e-9, (name, trial, err)
    print("%s: worst world reprojection %.2e relative" % (name, worst))


def test_oracle_rectify_pair_plan_equals_the_contract():
    """The oracle's whole-pair routine (orc.rectify_pair: what the GPU path is compared with) outputs exactly the contract's
    Q / R_final / T_final / P built from its own stereo_rectify."""
    from reconstruction_amd import synth
    raw = synth.make_raw_pair()
    r = orc.rectify_pair(raw["K"], raw["E"], raw["origin"], raw["lowest"], raw["pyr_levels"], raw["image"], raw["mask"])
    Q, R_final, T_final, P, _ = rectify_contract(raw["K"], raw["E"], raw["origin"], raw["lowest"][0], raw["pyr_levels"], orc.stereo_rectify)
    assert np.allclose(r["Q"], Q, rtol=1e-14, atol=0) and np.allclose(r["R_final"], R_final, rtol=0, atol=1e-13)
    assert np.allclose(r["T_final"], T_final, rtol=1e-14, atol=1e-12)
    for j in range(2):
        assert np.allclose(r["P"][j], P[j], rtol=1e-12, atol=1e-9)
