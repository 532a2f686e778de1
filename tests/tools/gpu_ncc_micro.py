"""Development aid: the NCC kernels alone (rsm_bench_ncc) at SURVEY 8(d)'s two points, per wide-row kernel."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from reconstruction_amd import Context
W, H = 4096, 3072
ctx = Context(0)
# the metric's own point (11x11, 129 candidates) and its neighbours: rows of pixels with long intervals take a row kernel from
# ncc_mid candidates on (default 64); ncc_mid = 160 keeps them in the 5-candidate band kernel (round 3's behaviour)
for mid in (160, 64, 32, 0):
    ctx.set_option("ncc_mid", mid)
    for r, cands in ((5, 129), (5, 65), (5, 33), (2, 65)):
        ms = ctx.bench_ncc(W, H, r, cands, iters=2)
        px = (W - 2 * r) * (H - 2 * r)
        print("ncc_mid=%d: %dx%d / %d candidates: %.2f ms per launch = %.1f GDE/s" % (mid, 2 * r + 1, 2 * r + 1, cands, ms, px * cands / ms / 1e6), flush=True)
ctx.set_option("ncc_mid", 0)
for wr, name in ((0, "per row: sliding sums up to 512 candidates, int8 row GEMM beyond"),):
    ctx.set_option("wide_rows", wr)
    for r, cands in ((5, 129), (7, 257), (5, 257), (7, 1025)):
        ms = ctx.bench_ncc(W, H, r, cands, iters=2)
        px = (W - 2 * r) * (H - 2 * r)
        print("wide_rows=%d (%s): %dx%d / %d candidates: %.2f ms per launch = %.1f GDE/s" % (wr, name, 2 * r + 1, 2 * r + 1, cands, ms, px * cands / ms / 1e6), flush=True)
for wr, name in ((3, "sliding window sums"), (2, "int8 row GEMM (MFMA)"), (1, "one workgroup per pixel")):
    ctx.set_option("wide_rows", wr)
    for r, cands in ((5, 129), (7, 257), (5, 257), (7, 1025)):
        if wr == 1 and cands > 257:
            continue
        ms = ctx.bench_ncc(W, H, r, cands, iters=2)
        px = (W - 2 * r) * (H - 2 * r)
        print("wide_rows=%d (%s): %dx%d / %d candidates: %.2f ms per launch = %.1f GDE/s" % (wr, name, 2 * r + 1, 2 * r + 1, cands, ms, px * cands / ms / 1e6), flush=True)
