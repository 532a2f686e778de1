"""Development aid: the 10-pair C3 rig on one GPU through rsm_match_pairs (host buffers in and out, PCIe included)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from reconstruction_amd import Context, match_pairs, synth
cfgs = [synth.config_c3(pair=p) for p in range(10)]
import sys
CASES = ((1, True, False), (2, True, False), (2, False, False), (3, False, False), (2, True, True), (3, True, True), (2, False, True), (3, False, True))
if len(sys.argv) > 1:  # e.g. 4,0,1 5,0,1 : contexts, disparity maps too, page-locked outputs
    CASES = tuple(tuple(int(v) for v in a.split(",")) for a in sys.argv[1:])
for nctx, disp, pinned in CASES:
    pool = [Context(0) for _ in range(nctx)]
    match_pairs(pool, cfgs[:nctx], want_cloud=True, want_disparity=disp, pinned=pinned)
    tm = {}
    res, st = match_pairs(pool, cfgs, want_cloud=True, want_disparity=disp, timing=tm, pinned=pinned)
    dt = tm["call_s"]
    v = sum(r.v_top for r in res)
    print("C3 10 pairs, %d context(s), %s host buffers in/out (%s): %.1f ms per pair, %.1f Mdisp/s (PCIe-inclusive), statuses %s"
          % (nctx, "page-locked OUTPUT," if pinned else "pageable", "cloud + both fp64 disparity maps" if disp else "cloud only", dt / 10 * 1e3, v / dt / 1e6, set(st)), flush=True)
    del res
    for c in pool:
        c.close()
