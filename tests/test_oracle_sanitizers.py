"""the CPU restatement under AddressSanitizer + UndefinedBehaviorSanitizer (make -C oracle asan).
The reference reads beyond rows in two places (CStereoMatching.cpp:628 the refine data term's right window,
:492 q[boundary_L + 1]) and its truncating int() casts see negative disparities (:286, :625); the restatement emulates
those reads on flat buffers (oracle/stereo_oracle.c).  Whole pairs that reach those places -- matches at the margin
edge, negative disparities drifting outward, holes and occlusions, radius 7 -- must run clean under the sanitizers AND give
the bytes of the ordinary -O3 build (exact-size heap blocks in the driver: one byte outside any buffer is reported)."""
import os
import struct
import subprocess

import numpy as np
import pytest

from oracle import oracle as orc
from reconstruction_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DRIVER = os.path.join(ROOT, "oracle", "_asan", "asan_driver")

CASES = {
    # full-width masks: the margins ARE the r-frame, so intervals and refine windows run into the image border
    "matches_at_the_margin_edge": dict(width=96, height=64, levels=2, radius=2, pair=0, mask_kind="full", border_l0=0),
    # odd pair index = negative view-0 disparities; large drift toward the left border
    "negative_disparities_drifting_outward": dict(width=128, height=64, levels=3, radius=2, pair=1, mask_kind="full",
                                                  d0_l0=4.0, amp_l0=1.0, border_l0=0),
    "holes_and_occlusion_radius5": dict(width=160, height=96, levels=2, radius=5, offset=4, pair=3, holes=True, occlude=True,
                                        border_l0=7),
    "radius7_three_levels": dict(width=192, height=96, levels=3, radius=7, pair=2, mask_l0_width=40, border_l0=8),
}


@pytest.fixture(scope="module")
def driver():
    r = subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "asan"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    return DRIVER


@pytest.mark.parametrize("name", list(CASES))
def test_whole_pair_is_clean_under_asan_and_ubsan(driver, tmp_path, name):
    cfg = synth.config_small(**CASES[name])
    W, H = cfg.width, cfg.height
    fin, fout = tmp_path / "in.bin", tmp_path / "out.bin"
    with open(fin, "wb") as f:
        f.write(struct.pack("<6i", W, H, cfg.pyr_levels, cfg.radius, cfg.offset, cfg.origin_width or W))
        f.write(struct.pack("<d", cfg.ws))
        for a in (cfg.Q, cfg.R_final, cfg.T_final):
            f.write(np.ascontiguousarray(a, np.float64).tobytes())
        for v in range(2):
            f.write(np.ascontiguousarray(cfg.image[v], np.uint8).tobytes())
        for v in range(2):
            f.write(np.ascontiguousarray(cfg.mask[v], np.uint8).tobytes())
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0:halt_on_error=1", UBSAN_OPTIONS="print_stacktrace=1",
               OMP_NUM_THREADS="4")
    r = subprocess.run([driver, str(fin), str(fout)], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0 and "Sanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-3000:]
    raw = open(fout, "rb").read()
    st, = struct.unpack_from("<i", raw, 0)
    margins = struct.unpack_from("<12i", raw, 4)
    n_points, v_top = struct.unpack_from("<2q", raw, 52)
    off = 68
    d = [np.frombuffer(raw, np.float64, W * H, off + v * W * H * 8).reshape(H, W) for v in range(2)]
    off += 2 * W * H * 8
    xyz = np.frombuffer(raw, np.float64, 3 * n_points, off).reshape(-1, 3)
    bgr = np.frombuffer(raw, np.uint8, 3 * n_points, off + 24 * n_points).reshape(-1, 3)
    ref = orc.match_pair(cfg)       # the ordinary -O3 shared library
    assert st == ref["status"] == 0
    assert list(margins) == list(ref["margin"][0]) + list(ref["margin"][1])
    assert n_points == ref["n_points"] > 0 and v_top == ref["v_top"]
    for v in range(2):
        assert np.array_equal(d[v], ref["disparity"][v])            # -O1 + sanitizers == -O3, bit for bit
        assert (d[v] != -10000).sum() > 50
    assert np.array_equal(xyz, ref["xyz"], equal_nan=True) and np.array_equal(bgr, ref["bgr"])
    if name == "negative_disparities_drifting_outward":
        assert np.median(d[0][d[0] != -10000]) < -4
