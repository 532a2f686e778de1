"""Development aid: the NCC kernels alone (rsm_bench_ncc) at SURVEY 8(d)'s two points, per wide-row kernel."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from reconstruction_amd import Context
W, H = 4096, 3072
ctx = Context(0)
for wr, name in ((3, "sliding window sums"), (2, "int8 row GEMM (MFMA)"), (1, "one workgroup per pixel")):
    ctx.set_option("wide_rows", wr)
    for r, cands in ((5, 129), (7, 257), (5, 257), (7, 1025)):
        if wr == 1 and cands > 257:
            continue
        ms = ctx.bench_ncc(W, H, r, cands, iters=2)
        px = (W - 2 * r) * (H - 2 * r)
        print("wide_rows=%d (%s): %dx%d / %d candidates: %.2f ms per launch = %.1f GDE/s" % (wr, name, 2 * r + 1, 2 * r + 1, cands, ms, px * cands / ms / 1e6), flush=True)
