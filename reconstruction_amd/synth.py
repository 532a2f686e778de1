"""Synthetic rectified stereo pairs.

The reference's demo inputs (config_ETH.yml / config_myself.yml, calibration,
images) are git-ignored upstream and absent, so every workload here is a seeded
synthetic stand-in of the same shape: band-limited random texture for view 0,
view 1 = view 0 resampled by a smooth disparity field plus small integer noise,
rectangular or elliptical masks, pinhole calibration giving Q / R_final /
T_final in the form CStereoMatching::Rectify leaves them
(CStereoMatching.cpp:128-138).

Disparity convention (CStereoMatching.cpp:221): disp[x] = x_match_in_other - x.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field

import numpy as np

NOMATCH = -10000


@dataclass
class PairConfig:
    """One stereo pair's inputs + parameters, as CStereoMatching sees them."""

    width: int                 # top level = LowestLevelSize * 2^(N-1)
    height: int
    pyr_levels: int            # m_PyrmNum
    radius: int = 2            # MatchBlockRadius (CReconstruction.cpp:17)
    ws: float = 0.03           # m_ws (CReconstruction.cpp:17)
    offset: int = 2            # m_offset (CStereoMatching.h:47)
    origin_width: int = 0      # m_OriginSize.width (0 -> width, scale = 1)
    image: list = field(default_factory=list)   # [2] uint8 HxWx3 BGR
    mask: list = field(default_factory=list)    # [2] uint8 HxW
    Q: np.ndarray | None = None                 # 4x4 fp64 (after the :138 sign flip)
    R_final: np.ndarray | None = None           # 3x3
    T_final: np.ndarray | None = None           # 3
    verbose: int = 0
    true_disparity: np.ndarray | None = None    # HxW fp64, view0 -> view1 (diagnostics only)
    name: str = ""


def _binomial_blur(a: np.ndarray, passes: int = 2) -> np.ndarray:
    k = np.array([1, 4, 6, 4, 1], dtype=np.float64) / 16.0
    for _ in range(passes):
        p = np.pad(a, ((0, 0), (2, 2), (0, 0)), mode="reflect")
        a = sum(k[i] * p[:, i:i + a.shape[1]] for i in range(5))
        p = np.pad(a, ((2, 2), (0, 0), (0, 0)), mode="reflect")
        a = sum(k[i] * p[i:i + a.shape[0]] for i in range(5))
    return a


def make_texture(width: int, height: int, seed: int, contrast: float = 3.0) -> np.ndarray:
    rng = np.random.default_rng(seed)
    a = rng.integers(0, 256, size=(height, width, 3)).astype(np.float64)
    a = _binomial_blur(a, 2)
    a = 127.5 + contrast * (a - 127.5)
    return np.clip(np.rint(a), 0, 255).astype(np.uint8)


def warp_view(img0: np.ndarray, disp: np.ndarray, noise_seed: int | None) -> np.ndarray:
    """view1(x, y) = bilinear(view0, x - disp(x, y), y), fp64 then rounded, + iid noise in [-2, 2]."""
    H, W, _ = img0.shape
    xs = np.arange(W, dtype=np.float64)[None, :] - disp
    xs = np.clip(xs, 0.0, W - 1.0)
    x0 = np.floor(xs).astype(np.int64)
    x1 = np.minimum(x0 + 1, W - 1)
    f = (xs - x0)[..., None]
    rows = np.arange(H)[:, None]
    a = img0[rows, x0].astype(np.float64)
    b = img0[rows, x1].astype(np.float64)
    out = a * (1.0 - f) + b * f
    if noise_seed is not None:
        rng = np.random.default_rng(noise_seed)
        out = out + rng.integers(-2, 3, size=out.shape)
    return np.clip(np.rint(out), 0, 255).astype(np.uint8)


def pinhole_calibration(width: int, height: int, pair: int = 0):
    """Q after the reference's sign flip (:138), R_final, T_final."""
    f = 1.2 * width
    cx, cy = width / 2.0, height / 2.0
    B = 100.0
    Q = np.array([[1, 0, 0, -cx], [0, 1, 0, -cy], [0, 0, 0, f], [0, 0, -1.0 / B, 0]], dtype=np.float64)
    if pair == 0:
        R = np.eye(3)
        T = np.zeros(3)
    else:
        th = math.radians(36.0 * pair)
        R = np.array([[math.cos(th), 0, math.sin(th)], [0, 1, 0], [-math.sin(th), 0, math.cos(th)]])
        T = np.array([600.0 * math.sin(th), 0.0, 600.0 * (1 - math.cos(th))])
    return Q, R, T


def make_pair(width: int, height: int, pyr_levels: int, radius: int = 2, offset: int = 2,
              ws: float = 0.03, pair: int = 0, mask_kind: str = "rect", mask_l0_width: int | None = None,
              d0_l0: float = 3.0, amp_l0: float = 2.0, noise: bool = True, holes: bool = False,
              border_l0: int = 8, name: str = "", occlude: bool = False) -> PairConfig:
    """Seeded synthetic pair.

    mask_kind "rect": 255 inside a centred rectangle `mask_l0_width * 2^(N-1)` wide
    (default: half the image width), full height minus `border_l0 * 2^(N-1)` rows top and
    bottom (C2/C5 style).  "ellipse": centred ellipse (C1/C3 style).  "full": all 255.
    View-1's mask is view-0's shifted by round(d0).  d0's sign alternates with `pair`.
    """
    assert width % (1 << (pyr_levels - 1)) == 0 and height % (1 << (pyr_levels - 1)) == 0
    s = 1000 + pair
    top = 1 << (pyr_levels - 1)
    sign = 1.0 if pair % 2 == 0 else -1.0
    d0 = sign * d0_l0 * top
    amp = amp_l0 * top
    assert amp * 2 * math.pi / width < 0.5, "ordering constraint needs |dd/dx| < 0.5"
    img0 = make_texture(width, height, s)
    xx = np.arange(width, dtype=np.float64)[None, :]
    yy = np.arange(height, dtype=np.float64)[:, None]
    disp1 = d0 + amp * np.sin(2 * math.pi * xx / width) * np.cos(2 * math.pi * yy / height)
    img1 = warp_view(img0, disp1, s + 500 if noise else None)
    if occlude:  # unrelated texture in two bands of view 1: forces NOMATCH / rematch / constraint kills
        other = make_texture(width, height, s + 700)
        img1[height // 3:height // 3 + max(4, height // 10), width // 4:3 * width // 4] = \
            other[height // 3:height // 3 + max(4, height // 10), width // 4:3 * width // 4]
        img1[:, width // 2:width // 2 + max(3, width // 40)] = other[:, width // 2:width // 2 + max(3, width // 40)]

    m0 = np.zeros((height, width), dtype=np.uint8)
    if mask_kind == "rect":
        mw = (mask_l0_width if mask_l0_width is not None else (width // top) // 2) * top
        x_lo = (width - mw) // 2
        b = border_l0 * top
        m0[b:height - b, x_lo:x_lo + mw] = 255
    elif mask_kind == "ellipse":
        ex = (xx - width / 2.0) / (0.36 * width)
        ey = (yy - height / 2.0) / (0.40 * height)
        m0[(ex * ex + ey * ey) <= 1.0] = 255
    elif mask_kind == "full":
        m0[:] = 255
    else:
        raise ValueError(mask_kind)
    if holes:
        rng = np.random.default_rng(s + 900)
        for _ in range(6):
            hx = int(rng.integers(width // 4, 3 * width // 4))
            hy = int(rng.integers(height // 4, 3 * height // 4))
            hw = int(rng.integers(2, 4)) * top
            m0[hy:hy + hw, hx:hx + hw] = 0
    sh = int(round(d0))
    m1 = np.zeros_like(m0)
    if sh >= 0:
        m1[:, sh:] = m0[:, :width - sh]
    else:
        m1[:, :width + sh] = m0[:, -sh:]
    Q, R, T = pinhole_calibration(width, height, pair)
    # disparity seen from view 0: d(x0) with x0 + d = x1 and x0 = x1 - disp1(x1); first-order field
    return PairConfig(width=width, height=height, pyr_levels=pyr_levels, radius=radius, ws=ws, offset=offset,
                      origin_width=width, image=[img0, img1], mask=[m0, m1], Q=Q, R_final=R, T_final=T,
                      true_disparity=disp1, name=name or f"synth{width}x{height}_p{pair}")


# --- the BASELINE.json configurations -------------------------------

def config_c1(pair: int = 0) -> PairConfig:
    """C1: ETH-style single pair 640x480, 5x5 NCC, 64 disparities at L0 (CPU plumbing case)."""
    return make_pair(640, 480, 3, radius=2, offset=2, pair=pair, mask_kind="rect", mask_l0_width=64,
                     name="C1_640x480_r2_d64")


def config_c2(pair: int = 0) -> PairConfig:
    """C2: single pair 4096x3072 (12.6 MP), 11x11 NCC, 128 disparities at L0, 5 levels."""
    return make_pair(4096, 3072, 5, radius=5, offset=2, pair=pair, mask_kind="rect", mask_l0_width=128,
                     name="C2_4096x3072_r5_d128")


def config_c2_sample(pair: int = 0) -> PairConfig:
    """Bounded CPU sample of C2: same 5 levels / 11x11 / offset 2, 1/4 of the area (2048x1536)."""
    return make_pair(2048, 1536, 5, radius=5, offset=2, pair=pair, mask_kind="rect", mask_l0_width=64,
                     border_l0=6, d0_l0=2.0, amp_l0=1.0, name="C2s_2048x1536_r5_d64")


def config_c3(pair: int = 0) -> PairConfig:
    """C3: one pair of the 10-camera portrait rig, 3072x4096, 11x11 NCC, ellipse mask."""
    return make_pair(3072, 4096, 5, radius=5, offset=2, pair=pair, mask_kind="ellipse",
                     name=f"C3_3072x4096_r5_p{pair}")


def config_c3_shipped(pair: int = 0) -> PairConfig:
    """C3': the 10-camera rig at the scale the reference ships (BatchProcess/main.cpp:59-61: lowest level 160x240,
    4 levels -> 1280x1920), 5x5 NCC (CReconstruction.cpp:17), ellipse mask."""
    return make_pair(1280, 1920, 4, radius=2, offset=2, pair=pair, mask_kind="ellipse",
                     name=f"C3s_1280x1920_r2_p{pair}")


def config_c5_reduced(pair: int = 0) -> PairConfig:
    """C5's geometry (15x15 NCC, 4 levels, 256 candidates at the lowest level) on a 2560x768 top level, small
    enough for the CPU oracle."""
    return make_pair(2560, 768, 4, radius=7, offset=2, pair=pair, mask_kind="rect", mask_l0_width=256, border_l0=10,
                     name=f"C5r_2560x768_r7_p{pair}")


def config_c5(pair: int = 0) -> PairConfig:
    """C5: synthetic 16-view stress: 4096x3072, 15x15 NCC, 256 disparities at L0, 4 levels."""
    return make_pair(4096, 3072, 4, radius=7, offset=2, pair=pair, mask_kind="rect", mask_l0_width=256,
                     name=f"C5_4096x3072_r7_p{pair}")


def config_small(width=96, height=64, levels=2, radius=2, offset=2, pair=0, mask_kind="rect",
                 holes=False, **kw) -> PairConfig:
    """Small seeded cases for parity tests."""
    return make_pair(width, height, levels, radius=radius, offset=offset, pair=pair, mask_kind=mask_kind,
                     holes=holes, **kw)


# --- raw (unrectified) pairs for CStereoMatching::Rectify ---------------------------------------

def _rot(rx, ry, rz):
    cx, sx, cy, sy, cz, sz = math.cos(rx), math.sin(rx), math.cos(ry), math.sin(ry), math.cos(rz), math.sin(rz)
    Rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
    Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    return Rz @ Ry @ Rx


def make_raw_pair(origin=(640, 480), lowest=(80, 60), pyr_levels=3, seed=7, baseline=60.0, depth=1500.0,
                  tilt=(0.01, -0.03, 0.008), mask_border=40):
    """Two calibrated pinhole views of a textured fronto-parallel plane Z = depth (world = camera-0 frame up to a
    small tilt), as raw images + masks + calibration (K 3x3, E 3x4 = [R | t], x_cam = R X + t).  After
    Rectify the true disparity is constant: f_rect * baseline / depth * scale."""
    ow, oh = origin
    f = 1.1 * ow
    K0 = np.array([[f, 0, ow / 2.0 + 3.0], [0, f * 1.01, oh / 2.0 - 2.0], [0, 0, 1.0]])
    K1 = np.array([[f * 0.99, 0, ow / 2.0 - 4.0], [0, f * 1.0, oh / 2.0 + 1.5], [0, 0, 1.0]])
    R0 = _rot(0.004, 0.006, -0.003)
    R1 = _rot(*tilt)
    C0 = np.array([0.0, 0.0, 0.0])
    C1 = np.array([baseline, 1.5, -2.0])
    E0 = np.hstack([R0, (-R0 @ C0)[:, None]])
    E1 = np.hstack([R1, (-R1 @ C1)[:, None]])
    tex = make_texture(2 * ow, 2 * oh, seed, contrast=3.0).astype(np.float64)
    s = 2 * ow / (depth / f * ow * 1.8)      # world units -> texture pixels
    imgs, msks = [], []
    uu, vv = np.meshgrid(np.arange(ow, dtype=np.float64), np.arange(oh, dtype=np.float64))
    for K, R, Cc in ((K0, R0, C0), (K1, R1, C1)):
        rays = np.stack([(uu - K[0, 2]) / K[0, 0], (vv - K[1, 2]) / K[1, 1], np.ones_like(uu)], -1) @ R  # R^T applied
        lam = (depth - Cc[2]) / rays[..., 2]
        X = Cc[0] + lam * rays[..., 0]
        Y = Cc[1] + lam * rays[..., 1]
        tx = np.clip(X * s + ow, 0, 2 * ow - 1.001)
        ty = np.clip(Y * s + oh, 0, 2 * oh - 1.001)
        x0 = np.floor(tx).astype(np.int64); y0 = np.floor(ty).astype(np.int64)
        fx = (tx - x0)[..., None]; fy = (ty - y0)[..., None]
        v = (tex[y0, x0] * (1 - fx) * (1 - fy) + tex[y0, x0 + 1] * fx * (1 - fy) +
             tex[y0 + 1, x0] * (1 - fx) * fy + tex[y0 + 1, x0 + 1] * fx * fy)
        imgs.append(np.clip(np.rint(v), 0, 255).astype(np.uint8))
        m = np.zeros((oh, ow), np.uint8)
        m[mask_border:oh - mask_border, mask_border:ow - mask_border] = 255
        msks.append(m)
    return dict(K=[K0, K1], E=[E0, E1], origin=origin, lowest=lowest, pyr_levels=pyr_levels, image=imgs, mask=msks,
                baseline=baseline, depth=depth)
