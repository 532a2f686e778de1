#!/usr/bin/env python
"""Generates tests/golden/ref_probe_golden.npz: seeded inputs + the outputs the REAL reference produces
for them (oracle/_ref/ref_probe = the reference's own CStereoMatching.cpp / CManageData.cpp objects and its
vendored Armadillo 4.200, compiled where they lie).  Runs only where /root/reference exists; the .npz it
writes is the committed fixture, this script is how it was made.

    make -C oracle/ref_probe && python oracle/ref_probe/make_golden.py
"""
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
import blob  # noqa: E402

NOMATCH = -10000


def build_inputs():
    rng = np.random.default_rng(20260929)
    a = {}
    # Armadillo primitives: lengths the path uses (27, 75 = 5x5x3, 363 = 11x11x3, 675) and edge lengths
    for i, n in enumerate([1, 2, 3, 26, 27, 75, 363, 675]):
        v = rng.integers(0, 256, n).astype(np.float64)
        if i == 3:
            v[:] = 17.0  # flat vector -> zero norm after mean removal
        a["arma_vec_%d" % i] = v
        a["arma_vec_b_%d" % i] = rng.normal(0, 50, n)
    meds = [[2, 4, 5, 9], [-9, -5, -4, -2], [3], [7, -1], [1, 2, 3], [5, 5, 6, 6, 7], [-3, -3, 8, 9, 10, 11],
            [-7, 2, -7, 2], [0, -1], [100, -100, 3, 4, 5, 6]]
    for i, m in enumerate(meds):
        a["median_in_%d" % i] = np.array(m, np.int32)
    # WindowToVec
    H, W = 30, 44
    img = rng.integers(0, 256, (H, W, 3)).astype(np.uint8)
    img[4:9, 10:15] = 93  # a flat 5x5 window (norm 0 -> 1)
    cases = []
    for w in (3, 5, 11):
        for _ in range(8):
            cases.append([int(rng.integers(0, W - w)), int(rng.integers(0, H - w)), w])
    cases.append([10, 4, 5])
    a["w2v_img"] = img
    a["w2v_cases"] = np.array(cases, np.int32)
    # FindMargin
    masks = []
    m = np.zeros((40, 60), np.uint8); m[8:30, 12:50] = 255; m[15, 3] = 255; m[2, 20] = 255; masks.append((m, 2))
    m = np.zeros((40, 60), np.uint8); masks.append((m, 3))                       # empty -> inverted defaults
    m = np.full((33, 47), 255, np.uint8); masks.append((m, 5))                   # full
    m = (rng.random((50, 70)) < 0.03).astype(np.uint8) * 255; m[m == 0] = rng.integers(0, 255, (m == 0).sum()); masks.append((m, 4))
    for i, (mk, r) in enumerate(masks):
        a["fm_mask_%d" % i] = mk
        a["fm_r_%d" % i] = np.array([r], np.int32)
    # OrderConstraint: smooth field + outliers + ties
    for i in range(4):
        Hh, Ww = 20, 90 + 10 * i
        d = (np.round(4 * np.sin(np.arange(Ww) / 9.0))[None, :] + rng.integers(-1, 2, (Hh, Ww))).astype(np.int16)
        out = rng.random((Hh, Ww)) < 0.04 * (i + 1)
        d[out] = rng.integers(-25, 26, out.sum())
        d[rng.random((Hh, Ww)) < 0.2] = NOMATCH
        XL, XR = 3 + i, Ww - 4
        a["oc_disp_%d" % i] = d
        a["oc_margin_%d" % i] = np.array([2, Hh - 3, XL, XR, XR - XL + 1, Hh - 4], np.int32)
    # UniquenessContraint_<short> / <double>
    for i in range(4):
        Hh, Ww = 24, 120
        p = rng.integers(-4, 5, (Hh, Ww)).astype(np.int16)
        q = rng.integers(-4, 5, (Hh, Ww)).astype(np.int16)
        ys, xs = np.nonzero(np.ones_like(p))
        for y, x in list(zip(ys, xs))[::2]:
            t = x + int(p[y, x])
            if 0 <= t < Ww:
                q[y, t] = -p[y, x] + int(rng.integers(-2, 3))
        p[rng.random((Hh, Ww)) < 0.15 + 0.1 * i] = NOMATCH
        q[rng.random((Hh, Ww)) < 0.2] = NOMATCH
        own = [3, Hh - 4, 8, Ww - 9, Ww - 16, Hh - 6]
        oth = [3, Hh - 4, 10, Ww - 12, Ww - 21, Hh - 6]
        if i >= 2:  # double flavour
            pf = np.where(p == NOMATCH, float(NOMATCH), p + rng.normal(0, 0.45, p.shape))
            qf = np.where(q == NOMATCH, float(NOMATCH), q + rng.normal(0, 0.45, q.shape))
            a["uq_p_%d" % i], a["uq_q_%d" % i] = pf, qf
        else:
            a["uq_p_%d" % i], a["uq_q_%d" % i] = p, q
        a["uq_margins_%d" % i] = np.array(own + oth, np.int32)
    build_inputs_round3(a)
    build_inputs_round5(a)
    return a


def build_inputs_round5(a):
    """Round 5: DisparityRefine's data term (.cpp:624-629) -- 3x3x3 windows of whole rows, every (own column, right-window
    left edge) pair: 8-bit noise, a band-limited texture with its shifted + noisy partner (what the bench data look like),
    two- and three-level textures (equal costs reached through different summation orders), flat and saturated regions
    (norm 0 -> 1 on either side, on both)."""
    rng = np.random.default_rng(20261001)
    kinds = ["random", "smooth_shifted", "two_level", "three_level", "flat_regions", "saturated"]
    for i, kind in enumerate(kinds):
        Hh, Ww = 6, 64
        if kind == "random":
            A = rng.integers(0, 256, (Hh, Ww, 3))
            B = rng.integers(0, 256, (Hh, Ww, 3))
        elif kind == "smooth_shifted":
            base = rng.integers(0, 256, (Hh + 4, Ww + 12, 3)).astype(np.float64)
            for _ in range(2):
                base = (base[:, :-4] + 4 * base[:, 1:-3] + 6 * base[:, 2:-2] + 4 * base[:, 3:-1] + base[:, 4:]) / 16
                base = (base[:-2] + 2 * base[1:-1] + base[2:]) / 4
            A = np.round(base[:Hh, :Ww])
            B = np.round(base[:Hh, 3:Ww + 3]) + rng.integers(-2, 3, (Hh, Ww, 3))
        elif kind == "two_level":
            A = rng.choice([0, 255], (Hh, Ww, 1)).repeat(3, 2)
            B = rng.choice([0, 255], (Hh, Ww, 1)).repeat(3, 2)
        elif kind == "three_level":
            A = rng.choice([10, 100, 250], (Hh, Ww, 3))
            B = np.where(rng.random((Hh, Ww, 3)) < 0.7, np.roll(A, 2, axis=1), rng.choice([10, 100, 250], (Hh, Ww, 3)))
        elif kind == "flat_regions":
            A = rng.integers(0, 256, (Hh, Ww, 3)); A[:, 10:20] = 93; A[:, 40:44] = 0
            B = rng.integers(0, 256, (Hh, Ww, 3)); B[:, 15:30] = 93; B[:, 50:] = 255
        else:
            A = rng.integers(200, 300, (Hh, Ww, 3))
            B = rng.integers(-40, 60, (Hh, Ww, 3))
        a["xi_imgA_%d" % i] = np.clip(A, 0, 255).astype(np.uint8)
        a["xi_imgB_%d" % i] = np.clip(B, 0, 255).astype(np.uint8)


def build_inputs_round3(a):
    """Round 3: inputs shaped like what the C2 workload produces (own generator: the entries above keep their bytes)."""
    rng = np.random.default_rng(20260930)
    # FindMargin at the radii of C2 / C5 (5, 7): isolated 255 pixels inside and outside the r-frame, values 254 nearby
    for i, (r, Hh, Ww) in enumerate([(5, 64, 97), (7, 71, 120), (5, 40, 40), (7, 31, 200)], start=4):
        m = rng.integers(0, 255, (Hh, Ww)).astype(np.uint8)          # never 255
        if i != 6:
            m[r + 6:Hh - r - 9, r + 11:Ww - r - 4] = 255
        m[r - 1, Ww // 2] = 255                                      # just outside the scanned frame: ignored
        m[Hh - r, 3] = 255
        m[r, r] = 255 if i == 5 else 254                             # the frame's corner pixel
        m[Hh // 2, Ww - r - 1] = 255                                 # last scanned column
        a["fm_mask_%d" % i] = m
        a["fm_r_%d" % i] = np.array([r], np.int32)
    # OrderConstraint rows with ONE long crossing component (C2's top level: an outlier ties ~2000 pixels together)
    for i, (Ww, nrows) in enumerate([(2100, 3), (2100, 3), (900, 4)], start=4):
        x = np.arange(Ww)
        d = np.round(6 * np.sin(x / 70.0) + 3 * np.sin(x / 13.0))[None, :].repeat(nrows + 4, 0).astype(np.int16)
        d += rng.integers(-1, 2, d.shape).astype(np.int16)
        for y in range(d.shape[0]):
            if i == 4:      # an early pixel thrown far to the right: crosses everything it jumps over
                d[y, 20 + y] = 2000
            elif i == 5:    # a late pixel thrown far to the left, plus a second long jump nested inside the first
                d[y, Ww - 30 - y] = -1990
                d[y, 300] = 1200
            else:           # several medium jumps and equal targets (ties of the crossing count)
                for k in range(6):
                    d[y, 60 + 130 * k] = 400 - 50 * k
                d[y, 500:520] = (519 - np.arange(500, 520)).astype(np.int16) + 500 - 500  # all land on column 519
        d[rng.random(d.shape) < 0.05] = NOMATCH
        XL, XR = 5, Ww - 6
        a["oc_disp_%d" % i] = d
        a["oc_margin_%d" % i] = np.array([2, 2 + nrows - 1, XL, XR, XR - XL + 1, nrows], np.int32)
    # UniquenessContraint<double>: values within 1e-12 of the int(p + 0.5) switch points k - 0.5, both signs, and of
    # the |q + p| < 2 decision
    for i in range(4, 6):
        Hh, Ww = 16, 160
        base = rng.integers(-5, 6, (Hh, Ww)).astype(np.float64)
        eps = rng.choice([0.0, 1e-12, -1e-12, 2.0 ** -40, -2.0 ** -40, 1e-9, -1e-9], (Hh, Ww))
        p = base - 0.5 + eps                                         # p + 0.5 = k + eps
        q = np.zeros((Hh, Ww))
        for y in range(Hh):
            for x in range(Ww):
                q[y, x] = -p[y, max(0, min(Ww - 1, x - int(rng.integers(-5, 6))))] + rng.choice([0.0, 2.0, -2.0, 2.0 - 1e-12, -2.0 + 1e-12, 1.3])
        p[rng.random((Hh, Ww)) < 0.1] = NOMATCH
        q[rng.random((Hh, Ww)) < 0.15] = NOMATCH
        own = [2, Hh - 3, 9, Ww - 10, Ww - 18, Hh - 4]
        oth = [2, Hh - 3, 11, Ww - 13, Ww - 23, Hh - 4]
        a["uq_p_%d" % i], a["uq_q_%d" % i] = p, q
        a["uq_margins_%d" % i] = np.array(own + oth, np.int32)
    # NCC scores of whole rows exactly as the matchers compute them (.cpp:202-211: vecL /= normL;
    # dot(vecL, vecR) / normR): random, two-grey-level, periodic and saturated textures (near and exact ties)
    for i, (r, kind) in enumerate([(2, "random"), (2, "two_level"), (5, "periodic"), (5, "saturated"), (7, "two_level"), (1, "random"),
                                   (1, "two_level_gray"), (2, "three_level"), (1, "three_level")]):
        Hh, Ww = 2 * r + 1 + 3, 72
        if kind == "random":
            A = rng.integers(0, 256, (Hh, Ww, 3))
            B = np.roll(A, 3, axis=1) + rng.integers(-6, 7, A.shape)
        elif kind == "two_level":
            A = rng.choice([40, 200], (Hh, Ww, 1)).repeat(3, 2)
            B = np.roll(A, -2, axis=1)
            B[:, ::7] = 200
        elif kind == "two_level_gray":      # r = 1: few distinct score values, reached through different summation orders
            A = rng.choice([0, 255], (Hh, Ww, 1)).repeat(3, 2)
            B = rng.choice([0, 255], (Hh, Ww, 1)).repeat(3, 2)
        elif kind == "three_level":
            A = rng.choice([10, 100, 250], (Hh, Ww, 3))
            B = np.where(rng.random((Hh, Ww, 3)) < 0.8, np.roll(A, 2, axis=1), rng.choice([10, 100, 250], (Hh, Ww, 3)))
        elif kind == "periodic":
            col = rng.integers(0, 256, (Hh, 8, 3))
            A = np.tile(col, (1, Ww // 8, 1))
            B = np.roll(A, 5, axis=1)
        else:
            A = rng.integers(0, 256, (Hh, Ww, 3))
            A[:, 20:45] = 255
            B = np.roll(A, 4, axis=1)
            B[:, 50:] = 0
        a["ncc_imgA_%d" % i] = np.clip(A, 0, 255).astype(np.uint8)
        a["ncc_imgB_%d" % i] = np.clip(B, 0, 255).astype(np.uint8)
        a["ncc_r_%d" % i] = np.array([r], np.int32)


def main():
    probe = os.path.join(ROOT, "oracle", "_ref", "ref_probe")
    if not os.path.exists(probe):
        subprocess.check_call(["make", "-s", "-C", HERE, "all"])
    inputs = build_inputs()
    with tempfile.TemporaryDirectory() as td:
        fi, fo = os.path.join(td, "in.blob"), os.path.join(td, "out.blob")
        blob.write(fi, inputs)
        subprocess.check_call([probe, fi, fo])
        outputs = blob.read(fo)
    dst = os.path.join(ROOT, "tests", "golden", "ref_probe_golden.npz")
    np.savez_compressed(dst, **{"in__" + k: v for k, v in inputs.items()}, **{"ref__" + k: v for k, v in outputs.items()})
    print("wrote", dst, os.path.getsize(dst), "bytes;", len(outputs), "reference outputs")


if __name__ == "__main__":
    main()
