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
    return a


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
