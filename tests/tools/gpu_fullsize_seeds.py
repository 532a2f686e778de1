"""One-off validation: further full-size workloads (other seeds / disparity signs, C5) against the whole-pair CPU oracle."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle import oracle as orc
from reconstruction_amd import Context, synth
ctx = Context(0)
for name, cfg in (("C2 pair 1", synth.config_c2(pair=1)), ("C2 pair 2", synth.config_c2(pair=2)), ("C3 pair 4", synth.config_c3(pair=4)),
                  ("C5 pair 0", synth.config_c5(pair=0))):
    t0 = time.time(); ref = orc.match_pair(cfg); t1 = time.time()
    res = ctx.match_pair(cfg)
    ok = res.margin == ref["margin"] and res.v_top == ref["v_top"] and res.n_points == ref["n_points"]
    for v in range(2):
        ok = ok and np.array_equal(res.disparity[v], ref["disparity"][v])
    fin = np.isfinite(ref["xyz"])
    ok = ok and np.array_equal(np.isfinite(res.xyz), fin) and np.array_equal(res.xyz[fin], ref["xyz"][fin]) and np.array_equal(res.bgr, ref["bgr"])
    print("%s: %s (oracle %.0f s, v_top %d, points %d)" % (name, "bit-identical" if ok else "MISMATCH", t1 - t0, res.v_top, res.n_points), flush=True)
