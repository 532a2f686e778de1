"""CPU analysis (oracle only): simulates the data-term cache of DisparityRefine under a nearest-neighbour prefetch policy
with T sweeps per launch, and counts the misses that remain inside the launches.
Usage: python tests/tools/analyze_prefetch.py W H levels nsweeps T [every]"""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from oracle import oracle as orc
from reconstruction_amd import synth
from helpers import oracle_stages

W, H, N, NS, T = (int(a) for a in sys.argv[1:6])
EVERY = int(sys.argv[6]) if len(sys.argv) > 6 else 0   # 1: requests from every sweep of a launch, 0: from its last sweep only
cache = "/tmp/keys_%d_%d_%d_%d.npz" % (W, H, N, NS)
if os.path.exists(cache):
    z = np.load(cache)
    states, mg = list(z["states"]), tuple(z["mg"])
else:
    cfg = synth.config_small(W, H, N, radius=2, offset=2, pair=1, holes=False)
    rec, fin = oracle_stages(cfg)
    r = [q for q in rec if q["stage"] == "refine" and q["level"] == N - 1 and q["v"] == 0][0]
    d16, mg = r["inp"], r["mg"][0]
    im = fin["imgs"][N - 1]
    states = [d16.astype(np.float64)]
    for n in range(1, NS + 1):
        states.append(orc.disparity_refine(d16, im[0], im[1], n, cfg.ws, mg))
    np.savez_compressed(cache, states=np.array(states), mg=np.array(mg))
YL, YR, XL, XR = mg[0], mg[1], mg[2], mg[3]
sl = (slice(YL + 1, YR), slice(XL + 1, XR))
valid = states[0][sl] != -10000
npx = int(valid.sum())
keys = [np.trunc(s[sl] - 1.5).astype(np.int64) for s in states]
fr = [s[sl] - 1.5 - k for s, k in zip(states, keys)]
way = [np.full(valid.shape, -99999, np.int64), np.full(valid.shape, -99999, np.int64)]

def resident(k):
    return np.where((k & 1) == 1, way[1], way[0]) == k

def install(k, sel):
    way[0][sel & ((k & 1) == 0)] = k[sel & ((k & 1) == 0)]
    way[1][sel & ((k & 1) == 1)] = k[sel & ((k & 1) == 1)]

def wanted(n):
    """keys a pixel in state n will want: its own key and the nearer neighbour"""
    k = keys[n]
    nn = np.where(fr[n] > 0.5, k + 1, k - 1)
    return k, nn

# sweep 1 = k_refine_first: own key + requests from state 1
install(keys[0], valid)
tot_pref = 0
k, nn = wanted(1)
for q in (k, nn):
    need = valid & ~resident(q)
    tot_pref += int(need.sum())
    install(q, need)
print("pixels %d; after the first sweep: %d prefetched entries (%.2f per pixel)" % (npx, tot_pref, tot_pref / npx))
tot_miss = 0
n = 1
while n < NS:
    m_launch = 0
    reqs = []
    for t in range(T):
        if n >= NS:
            break
        k = keys[n]
        miss = valid & ~resident(k)
        m_launch += int(miss.sum())
        install(k, miss)
        n += 1
        if EVERY or t == T - 1 or n == NS:
            reqs.append(n)
    p_launch = 0
    for q in reqs:
        for kk in wanted(q):
            need = valid & ~resident(kk)
            p_launch += int(need.sum())
            install(kk, need)
    tot_miss += m_launch
    tot_pref += p_launch
    if n <= 42 or n % 20 < T:
        print("launch ending at sweep %3d: in-launch misses %7d (%.4f%% of pixel-sweeps), prefetches after it %7d" %
              (n, m_launch, 100.0 * m_launch / (npx * T), p_launch))
print("total in-launch misses %d (%.3f per pixel), total prefetches %d (%.2f per pixel)" % (tot_miss, tot_miss / npx, tot_pref, tot_pref / npx))
