"""Shared test helpers: the oracle's MatchOneLayer sequence with every stage boundary captured."""
from __future__ import annotations

import numpy as np

from oracle import oracle as orc

NOMATCH = -10000


def build_pyramid(cfg):
    """ConstructPyrm with the oracle's pyrDown. Returns imgs[k][v], msks[k][v] (k = 0 lowest)."""
    N = cfg.pyr_levels
    imgs = [None] * N
    msks = [None] * N
    imgs[N - 1] = [np.ascontiguousarray(cfg.image[0]), np.ascontiguousarray(cfg.image[1])]
    msks[N - 1] = [np.ascontiguousarray(cfg.mask[0]), np.ascontiguousarray(cfg.mask[1])]
    for k in range(N - 2, -1, -1):
        imgs[k] = [orc.pyr_down(imgs[k + 1][v]) for v in range(2)]
        msks[k] = [orc.pyr_down(msks[k + 1][v]) for v in range(2)]
    return imgs, msks


def oracle_stages(cfg, max_levels=None):
    """Runs the oracle stage by stage (the 17-call sequence of CStereoMatching.cpp:36-113) and records
    every stage's inputs and outputs.  Returns (records, final) where records is a list of dicts
    {level, stage, v, ...} and final = dict(disparity=[d0, d1], margin=[m0, m1], imgs, msks)."""
    N = cfg.pyr_levels
    r, off, ws = cfg.radius, cfg.offset, cfg.ws
    imgs, msks = build_pyramid(cfg)
    rec = []
    dd = [None, None]
    mg = None
    for k in range(N if max_levels is None else min(N, max_levels)):
        mg = [orc.find_margin(msks[k][v], r).astuple() for v in range(2)]
        im, mk = imgs[k], msks[k]
        ds = [None, None]
        for v in range(2):
            o = 1 - v
            if k == 0:
                ds[v] = orc.lowest_level_initial_match(im[v], im[o], mk[v], mk[o], r, mg[v], mg[o])
                rec.append(dict(level=k, stage="initial", v=v, parent=None, out=ds[v].copy()))
            else:
                ds[v] = orc.high_level_initial_match(im[v], im[o], mk[v], mk[o], r, off, mg[v], mg[o], dd[v])
                rec.append(dict(level=k, stage="initial", v=v, parent=dd[v].copy(), out=ds[v].copy()))
        for v in range(2):
            i = ds[v].copy()
            ds[v] = orc.smooth_constraint(ds[v], mg[v])
            rec.append(dict(level=k, stage="smooth", v=v, inp=i, out=ds[v].copy()))
        for v in range(2):
            i = ds[v].copy()
            ds[v] = orc.order_constraint(ds[v], mg[v])
            rec.append(dict(level=k, stage="order", v=v, inp=i, out=ds[v].copy()))
        i0, i1 = ds[0].copy(), ds[1].copy()
        ds[0], ds[1] = orc.uniqueness(ds[0], ds[1], mg[0], mg[1])
        rec.append(dict(level=k, stage="uniq16", inp=[i0, i1], out=[ds[0].copy(), ds[1].copy()]))
        for v in range(2):
            o = 1 - v
            i = ds[v].copy()
            st, BL, BR = orc.set_boundary_smooth(ds[v], mk[v], mg[v], mg[o])
            assert st == 0
            rec.append(dict(level=k, stage="setb", v=v, inp=i, BL=BL, BR=BR))
            st, ds[v] = orc.rematch(im[v], im[o], mk[v], mk[o], r, mg[v], mg[o], ds[v])
            assert st == 0
            rec.append(dict(level=k, stage="rematch", v=v, inp=i, out=ds[v].copy()))
        i0, i1 = ds[0].copy(), ds[1].copy()
        ds[0], ds[1] = orc.uniqueness(ds[0], ds[1], mg[0], mg[1])
        rec.append(dict(level=k, stage="uniq16", inp=[i0, i1], out=[ds[0].copy(), ds[1].copy()]))
        for v in range(2):
            i = ds[v].copy()
            ds[v] = orc.median_filter(ds[v], mk[v], mg[v])
            rec.append(dict(level=k, stage="median", v=v, inp=i, out=ds[v].copy()))
        iters = 30 + 30 * k
        nd = [None, None]
        for v in range(2):
            nd[v] = orc.disparity_refine(ds[v], im[v], im[1 - v], iters, ws, mg[v])
            rec.append(dict(level=k, stage="refine", v=v, inp=ds[v].copy(), iters=iters, out=nd[v].copy()))
        i0, i1 = nd[0].copy(), nd[1].copy()
        nd[0], nd[1] = orc.uniqueness(nd[0], nd[1], mg[0], mg[1])
        rec.append(dict(level=k, stage="uniq64", inp=[i0, i1], out=[nd[0].copy(), nd[1].copy()]))
        dd = nd
        for q in rec:
            if q["level"] == k:
                q.setdefault("mg", mg)
    return rec, dict(disparity=dd, margin=mg, imgs=imgs, msks=msks)


def cloud_scale(cfg):
    top = 1 << (cfg.pyr_levels - 1)
    return float(cfg.width // top) / (cfg.origin_width or cfg.width) * top


def diff_report(name, a, b):
    a = np.asarray(a); b = np.asarray(b)
    bad = np.argwhere(a != b)
    msg = "%s: %d / %d elements differ" % (name, len(bad), a.size)
    for idx in bad[:8]:
        idx = tuple(int(i) for i in idx)
        msg += "\n   at %s: got %s expected %s" % (idx, a[idx], b[idx])
    return msg


def libm_exp_stats(res_disp, ref_disp):
    """HIP result (specified exp) against the oracle run with the HOST libm's exp -- the situation of the reference,
    whose C runtime's exp has an unspecified last bit (CStereoMatching.cpp:665-666).  Returns per view:
    (pixels valid in both, NOMATCH-set mismatches, count above 1e-3 relative, max relative error)."""
    out = []
    for v in range(2):
        a, b = np.asarray(res_disp[v]), np.asarray(ref_disp[v])
        na, nb = a == NOMATCH, b == NOMATCH
        both = ~na & ~nb
        rel = np.abs(a[both] - b[both]) / np.maximum(1.0, np.abs(b[both]))
        out.append(dict(valid=int(both.sum()), nomatch_mismatch=int((na != nb).sum()), above_1e3=int((rel > 1e-3).sum()),
                        above_1e9=int((rel > 1e-9).sum()), max_rel=float(rel.max()) if rel.size else 0.0))
    return out


def host_libm_is_glibc_with_fma():
    """Whether the host libm's exp is the routine the specification restates, run the way it restates it: glibc >= 2.28
    (the table-driven exp) on an x86-64 CPU with FMA3 (glibc's ifunc then picks __exp_fma)."""
    import ctypes
    import platform
    if platform.machine() != "x86_64":
        return False
    try:
        gnu = ctypes.CDLL(None).gnu_get_libc_version
        gnu.restype = ctypes.c_char_p
        major, minor = (int(v) for v in gnu().decode().split(".")[:2])
    except Exception:
        return False
    with open("/proc/cpuinfo") as f:
        flags = f.read()
    return (major, minor) >= (2, 28) and " fma " in flags.replace("\n", " ")
