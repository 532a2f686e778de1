"""CPU analysis (oracle only): how the refine data-term cache keys int(d - 1.5) of a level's pixels move over the sweeps,
and how well a key could be predicted ahead of its first use.  Usage: python tests/tools/analyze_keys.py [W H levels nsweeps]"""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from oracle import oracle as orc
from reconstruction_amd import synth
from helpers import oracle_stages

W, H, N, NS = (int(a) for a in (sys.argv[1:5] + [640, 480, 3, 60][len(sys.argv) - 1:]))
cfg = synth.config_small(W, H, N, radius=2, offset=2, pair=1, holes=False)
t0 = time.time()
rec, fin = oracle_stages(cfg)
r = [q for q in rec if q["stage"] == "refine" and q["level"] == N - 1 and q["v"] == 0][0]
d16, mg = r["inp"], r["mg"][0]
im = fin["imgs"][N - 1]
print("stages %.1fs; level %dx%d margin %s" % (time.time() - t0, d16.shape[1], d16.shape[0], mg))
states = [d16.astype(np.float64)]
for n in range(1, NS + 1):
    states.append(orc.disparity_refine(d16, im[0], im[1], n, cfg.ws, mg))
print("sweeps %.1fs" % (time.time() - t0))
YL, YR, XL, XR = mg[0], mg[1], mg[2], mg[3]
sl = (slice(YL + 1, YR), slice(XL + 1, XR))
valid = states[0][sl] != -10000
keys = [np.trunc(s[sl] - 1.5).astype(np.int64) for s in states]  # key used BY sweep n+1 is keys[n]
fr = [s[sl] - 1.5 - k for s, k in zip(states, keys)]
npx = int(valid.sum())
print("interior valid pixels", npx)
# two-way cache simulation (way = key & 1), as the kernels do
way = [np.full(valid.shape, -99999, np.int64), np.full(valid.shape, -99999, np.int64)]
PREFILL = int(os.environ.get("PREFILL", "0"))  # 1: the first sweep also installs the neighbour key in the direction of its update
seen_sets = [set() for _ in range(0)]
tot_miss = 0
first_touch = 0
touched = {}
hist = np.zeros(valid.shape + (0,), np.int64)
keyhist = []  # list of arrays: all keys ever installed per pixel (for first-touch classification)
installed = np.zeros(valid.shape, np.int64)
everkeys = [np.full(valid.shape, -99999, np.int64) for _ in range(8)]
nkeys = np.zeros(valid.shape, np.int64)
pred_ok = pred_tot = 0
for n in range(NS):
    k = keys[n]
    w = k & 1
    cur = np.where(w == 1, way[1], way[0])
    miss = valid & (cur != k)
    # was this key ever installed before (conflict miss) or first touch?
    ever = np.zeros(valid.shape, bool)
    for e in everkeys:
        ever |= (e == k)
    ft = miss & ~ever
    # prediction made from the state one sweep earlier (n-1): neighbour by the fractional part
    if n >= 1:
        kp = keys[n - 1]
        pred = np.where(fr[n - 1] > 0.5, kp + 1, kp - 1)
        pm = miss & (kp != k)
        pred_tot += int(pm.sum())
        pred_ok += int((pm & (pred == k)).sum())
    nm = int(miss.sum())
    tot_miss += nm
    first_touch += int(ft.sum())
    if n < 12 or n % 10 == 0:
        print("sweep %3d: misses %8d (%.3f%%), first-touch %8d, |dk|==1 among misses %.3f" %
              (n + 1, nm, 100.0 * nm / npx, int(ft.sum()),
               float((np.abs(k - keys[n - 1]) == 1)[miss].mean()) if n >= 1 and nm else 0.0))
    # install
    if PREFILL and n == 0:
        k2 = np.where(states[1][sl] > states[0][sl], k + 1, k - 1)
        w2 = k2 & 1
        way[0] = np.where(valid & (w2 == 0), k2, way[0])
        way[1] = np.where(valid & (w2 == 1), k2, way[1])
        everkeys[7][valid] = k2[valid]
    way[0] = np.where(miss & (w == 0), k, way[0])
    way[1] = np.where(miss & (w == 1), k, way[1])
    slot = np.minimum(nkeys, 7)
    for i, e in enumerate(everkeys):
        sel = ft & (slot == i)
        e[sel] = k[sel]
    nkeys += ft
print("total misses %d (%.2f per pixel), first-touch %d, distinct keys per pixel: mean %.2f, hist %s" %
      (tot_miss, tot_miss / npx, first_touch, nkeys[valid].mean(), np.bincount(nkeys[valid])[:8]))
print("fractional-part neighbour prediction one sweep ahead: %d / %d = %.3f" % (pred_ok, pred_tot, pred_ok / max(1, pred_tot)))
# prediction made ONCE after the first sweep (state 1): which neighbour key is the first new key the pixel needs?
k1 = keys[1]
firstnew = np.full(valid.shape, -99999, np.int64)
when = np.full(valid.shape, -1, np.int64)
for n in range(2, NS):
    sel = valid & (firstnew == -99999) & (keys[n] != k1) & (keys[n] != keys[0])
    firstnew[sel] = keys[n][sel]
    when[sel] = n
has = valid & (firstnew != -99999)
pred1 = np.where(fr[1] > 0.5, k1 + 1, k1 - 1)
print("pixels needing a third key within %d sweeps: %d (%.1f%%); predicted from state 1 by the fractional part: %.3f" %
      (NS, int(has.sum()), 100.0 * has.sum() / npx, float((pred1 == firstnew)[has].mean()) if has.any() else 0))
