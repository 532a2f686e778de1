"""The control that turns "DisparityRefine amplifies last-bit differences of exp" from an argument into a measurement
(CStereoMatching.cpp:665-666).  The reference calls its C runtime's exp, whose last bit is unspecified.  The
oracle can evaluate the smoothness weights three ways: (0) the fully specified exp the kernels share (a degree-13 fma
Horner chain, within 1 ulp: its last bit differs from glibc's in ~9 % of the arguments), (1) the host libm's exp, (2) the
host libm's long-double expl rounded to double -- a second libm-grade exp that differs from (1) in a few arguments per
hundred thousand.  On the 5-level occluded 512x384 pair (the one case whose HIP result misses north_star's 1e-3 against
the libm oracle for a handful of pixels), measured here and asserted:

  * libm vs expl -- two exps NEITHER of which is ours, nearly identical functions -- still end up visibly apart: dozens
    of pixels differ by more than 1e-9 (up to ~1e-6) after the level's 150 sweeps, although each sweep is accurate to
    1e-16: the iteration amplifies a single last-bit difference by ten orders of magnitude;
  * specified vs either libm drifts more (hundreds of pixels, worst 7e-3) in proportion to how often its last bit differs,
    and stays inside the same envelope: >= 99.9 % of the pixels within 1e-3, none above 5e-2;
  * the pixels that drift are the same ill-conditioned ones whichever two exps are compared (overlap far beyond chance);
  * the NOMATCH sets and point counts are identical in all three.
So the residual against the libm oracle measures the reference's own sensitivity to ITS runtime's exp, not an error of the
kernels: a reference rebuilt against another C runtime differs from itself the same way.

CPU only (oracle against oracle).  The numbers land in tests/golden/exp_control_stats.json when RSM_WRITE_EXP_CONTROL=1."""
import json
import os

import numpy as np

from oracle import oracle as orc
from reconstruction_amd import synth

NOMATCH = -10000
CASE = dict(width=512, height=384, levels=5, radius=3, offset=2, pair=21, mask_l0_width=16, holes=True, occlude=True)


def _run(mode, cfg):
    orc.set_exp_mode(mode)
    try:
        return orc.match_pair(cfg)
    finally:
        orc.set_exp_mode(0)


def _cmp(a, b):
    out = []
    for v in range(2):
        x, y = a["disparity"][v], b["disparity"][v]
        nx, ny = x == NOMATCH, y == NOMATCH
        both = ~nx & ~ny
        rel = np.zeros(x.shape)
        rel[both] = np.abs(x[both] - y[both]) / np.maximum(1.0, np.abs(y[both]))
        out.append(dict(valid=int(both.sum()), nomatch_mismatch=int((nx != ny).sum()), above_1e3=int((rel > 1e-3).sum()),
                        above_1e9=int((rel > 1e-9).sum()), max_rel=float(rel.max()), drift=rel > 1e-9))
    return out


def test_two_libm_grade_exps_disagree_like_the_specified_one_does():
    cfg = synth.config_small(**CASE)
    r = {m: _run(m, cfg) for m in (0, 1, 2)}
    assert r[0]["n_points"] == r[1]["n_points"] == r[2]["n_points"] > 1000
    pairs = {"specified_vs_libm": _cmp(r[0], r[1]), "specified_vs_expl": _cmp(r[0], r[2]), "libm_vs_expl": _cmp(r[1], r[2])}
    stats = {k: [{kk: vv for kk, vv in s.items() if kk != "drift"} for s in v] for k, v in pairs.items()}
    print(json.dumps(stats, indent=1))
    for k, v in pairs.items():
        for s in v:
            assert s["nomatch_mismatch"] == 0, k                      # the NOMATCH sets never depend on the exp
            assert s["above_1e3"] <= 2e-3 * s["valid"] and s["max_rel"] < 5e-2, (k, s["above_1e3"], s["max_rel"])
    # how often the three exps differ in their last bit, on the argument range of the weights
    t = np.random.default_rng(7).uniform(0.0, 30.0, 400000)
    e0 = orc.exp_neg_array(t)
    orc.set_exp_mode(1)
    e1 = orc.exp_neg_array(t)
    orc.set_exp_mode(2)
    e2 = orc.exp_neg_array(t)
    orc.set_exp_mode(0)
    rate = {"specified_vs_libm": float((e0 != e1).mean()), "specified_vs_expl": float((e0 != e2).mean()), "libm_vs_expl": float((e1 != e2).mean())}
    stats["last_bit_disagreement_rate"] = rate
    print(rate)
    assert np.abs(e0 - e1).max() <= np.spacing(e1).max() and rate["libm_vs_expl"] < 1e-3 < rate["specified_vs_libm"] < 0.2
    ctrl = pairs["libm_vs_expl"]
    # two nearly identical exps, neither of them ours, still drift apart visibly: amplification by ~10 orders of magnitude
    assert sum(s["above_1e9"] for s in ctrl) > 50 and max(s["max_rel"] for s in ctrl) > 1e-7
    for name in ("specified_vs_libm", "specified_vs_expl"):
        ours = pairs[name]
        # ours differs in its last bit ~1000 times more often and drifts more, but nowhere near in proportion: the same few
        # ill-conditioned pixels carry the drift whichever two exps are compared (overlap far beyond chance)
        assert sum(s["above_1e9"] for s in ours) <= 20 * sum(s["above_1e9"] for s in ctrl), name
        for v in range(2):
            a, b = ours[v]["drift"], ctrl[v]["drift"]
            if a.sum() and b.sum():
                chance = a.mean() * b.mean() * a.size
                assert (a & b).sum() > 10 * chance, (name, v, int((a & b).sum()), chance)
    if os.environ.get("RSM_WRITE_EXP_CONTROL") == "1":
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "exp_control_stats.json")
        with open(path, "w") as f:
            json.dump({"case": CASE, "modes": {"specified": 0, "libm": 1, "expl": 2}, "stats": stats}, f, indent=1)
