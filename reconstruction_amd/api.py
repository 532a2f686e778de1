"""Host-side mirror of the reference's call surface for the CStereoMatching path.

`StereoMatching` keeps the names, argument meaning and error behaviour of
reconstruction/CStereoMatching.h:35-48 (Init / MatchAllLayer, public fields Q, R_final, T_final,
margin, MatchBlockRadius, m_ws, m_offset, Verbose); `ManageData` / `Camera` carry the fields of
CManageData.h:16-40 this path reads and writes.  All compute goes through the C ABI of
include/rsm.h (librsm_mi355.so, hand-written HIP for gfx950) -- there is no CPU path here.

MatAllLayer either rectifies on the GPU (cameras carry calibration + raw image / mask
files or arrays, `ManageData.rectified[pair]` is absent) or starts from rectified inputs (each
`ManageData.cam[pair][v]` holds what Rectify leaves in `.image` / `.mask`, .cpp:154-158, and
`ManageData.rectified[pair]` the Q / R_final / T_final it computes, .cpp:128-138).
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field

import numpy as np

from . import _lib
from ._lib import Boundary, FilterParams, NOMATCH, PairIn, PairOut, RectifyIn, RectifyOut, RsmError  # noqa: F401


class _Pinned:
    """Owner of one rsm_host_alloc block (freed with the last array that views it)."""
    def __init__(self, nbytes):
        self._lib = _lib.load()
        self.ptr = self._lib.rsm_host_alloc(C.c_size_t(max(1, nbytes)))
        if not self.ptr:
            raise RsmError(-3, "rsm_host_alloc(%d) failed" % nbytes)

    def __del__(self):
        try:
            self._lib.rsm_host_free(C.c_void_p(self.ptr))
        except Exception:
            pass


def host_empty(shape, dtype=np.float64):
    """numpy array in page-locked host memory (rsm_host_alloc): uploads from / downloads into it are single DMAs."""
    shape = (shape,) if np.isscalar(shape) else tuple(shape)
    dt = np.dtype(dtype)
    n = int(np.prod(shape)) * dt.itemsize
    own = _Pinned(n)
    buf = (C.c_uint8 * max(1, n)).from_address(own.ptr)
    buf._owner = own  # keeps the block alive as long as any view of `buf` lives
    return np.frombuffer(buf, dtype=dt, count=int(np.prod(shape))).reshape(shape)


def _filled(a, v):
    a[...] = v
    return a


def _u8(a):
    return np.ascontiguousarray(a, dtype=np.uint8)


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _bd(t) -> Boundary:
    if isinstance(t, Boundary):
        return t
    return Boundary(*t)


def stereo_rectify(K1, K2, size, R, T):
    """cv::stereoRectify(K1, 0, K2, 0, size, R, T, flags=0, alpha=-1) through the C ABI (host fp64, no GPU needed)."""
    a = [np.ascontiguousarray(x, np.float64) for x in (K1, K2, R, T)]
    R1 = np.zeros((3, 3)); R2 = np.zeros((3, 3)); P1 = np.zeros((3, 4)); P2 = np.zeros((3, 4)); Q = np.zeros((4, 4))
    st = _lib.load().rsm_stereo_rectify(_p(a[0]), _p(a[1]), int(size[0]), int(size[1]), _p(a[2]), _p(a[3]),
                                        _p(R1), _p(R2), _p(P1), _p(P2), _p(Q))
    if st != 0:
        raise RsmError(st, "rsm_stereo_rectify")
    return R1, R2, P1, P2, Q


def write_ply(path, xyz, bgr, normals=None):
    """cloud%d.ply of DisparityToCloud (.cpp:723-729,754-756) through the C ABI (host-only, no GPU needed).
    With `normals` ([n,4]: nx, ny, nz, curvature -- pcl::PointNormal's fields, what filter() accumulates in
    cloud_normals, CCloudOptimization.cpp:123) the same records are followed by four float properties."""
    xyz = np.ascontiguousarray(xyz, np.float64).reshape(-1, 3)
    bgr = _u8(bgr).reshape(-1, 3)
    assert len(xyz) == len(bgr)
    if normals is not None:
        nrm = np.ascontiguousarray(normals, np.float32).reshape(-1, 4)
        assert len(nrm) == len(xyz)
        rec = np.zeros(len(xyz), dtype=[("xyz", "<f4", 3), ("bgr", "u1", 3), ("n", "<f4", 4)])
        rec["xyz"], rec["bgr"], rec["n"] = xyz.astype(np.float32), bgr, nrm
        with open(path, "wb") as f:
            f.write(("ply\nformat binary_little_endian 1.0\nelement vertex %d\nproperty float x\nproperty float y\n"
                     "property float z\nproperty uchar blue\nproperty uchar green\nproperty uchar red\nproperty float nx\n"
                     "property float ny\nproperty float nz\nproperty float curvature\nend_header\n" % len(xyz)).encode())
            f.write(rec.tobytes())
        return
    st = _lib.load().rsm_write_ply(str(path).encode(), _p(xyz), _p(bgr), C.c_int64(len(xyz)))
    if st != 0:
        raise RsmError(st, "rsm_write_ply(%s)" % path)


@dataclass
class PairResult:
    disparity: list            # [2] float64 HxW (NOMATCH = -10000)
    margin: list               # [2] (YL, YR, XL, XR, width, height)
    n_points: int
    xyz: np.ndarray            # n_points x 3 float64, InsertPoint order
    bgr: np.ndarray            # n_points x 3 uint8
    v_top: int


class Context:
    """One rsm_ctx = one GPU. Not re-entrant (like CStereoMatching)."""

    def __init__(self, device: int = 0):
        self._lib = _lib.load()
        h = C.c_void_p()
        st = self._lib.rsm_create(C.byref(h), int(device))
        if st != 0:
            raise RsmError(st, "rsm_create(device=%d) failed -- is an MI355X visible? (no CPU fallback)" % device)
        self._h = h
        self.device = device
        self._keep = None

    def close(self):
        if getattr(self, "_h", None):
            self._lib.rsm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _chk(self, st):
        if st != 0:
            raise RsmError(st, (self._lib.rsm_last_error(self._h) or b"").decode())

    # ---- whole pair ------------------------------------------------------------------------------
    @staticmethod
    def _pair_in(cfg, imgs=None, msks=None):
        pin = PairIn()
        imgs = imgs or [_u8(cfg.image[0]), _u8(cfg.image[1])]
        msks = msks or [_u8(cfg.mask[0]), _u8(cfg.mask[1])]
        H, W = msks[0].shape
        assert imgs[0].shape == (H, W, 3) and imgs[1].shape == (H, W, 3) and msks[1].shape == (H, W)
        assert (W, H) == (cfg.width, cfg.height)
        for v in range(2):
            pin.image[v] = imgs[v].ctypes.data
            pin.mask[v] = msks[v].ctypes.data
        pin.width, pin.height, pin.pyr_levels = W, H, cfg.pyr_levels
        pin.radius, pin.ws, pin.offset = cfg.radius, cfg.ws, cfg.offset
        pin.origin_width = cfg.origin_width or W
        pin.Q[:] = list(np.asarray(cfg.Q, np.float64).ravel())
        pin.R_final[:] = list(np.asarray(cfg.R_final, np.float64).ravel())
        pin.T_final[:] = list(np.asarray(cfg.T_final, np.float64).ravel())
        pin.verbose = int(getattr(cfg, "verbose", 0))
        return pin, (imgs, msks)

    def upload_pair(self, cfg):
        pin, keep = self._pair_in(cfg)
        self._chk(self._lib.rsm_upload_pair(self._h, C.byref(pin)))
        self._shape = (cfg.height, cfg.width)

    def upload_pair_device(self, cfg, image_ptrs, mask_ptrs):
        """image_ptrs / mask_ptrs: device addresses (e.g. torch.Tensor.data_ptr()) on this ctx's GPU."""
        pin, _ = self._pair_in(cfg)
        for v in range(2):
            pin.image[v] = int(image_ptrs[v])
            pin.mask[v] = int(mask_ptrs[v])
        self._chk(self._lib.rsm_upload_pair_device(self._h, C.byref(pin)))
        self._shape = (cfg.height, cfg.width)

    def rectify_pair(self, K, E, origin_size, lowest_size, pyr_levels, images, masks, radius=2, ws=0.03, offset=2,
                     verbose=0, want_images=True):
        """CStereoMatching::Rectify for one pair on the GPU (.cpp:117-168); the rectified pair stays resident, so
        run_pair() can follow.  K / E: [2] 3x3 / 3x4, images / masks: raw BGR / grey arrays of origin size
        (width, height).  Returns dict(image, mask, Q, R_final, T_final, P, size)."""
        rin, rout = RectifyIn(), RectifyOut()
        imgs = [_u8(i) for i in images]
        msks = [_u8(m) for m in masks]
        ow, oh = int(origin_size[0]), int(origin_size[1])
        for v in range(2):
            assert imgs[v].shape == (oh, ow, 3) and msks[v].shape == (oh, ow)
            rin.K[v][:] = list(np.asarray(K[v], np.float64).ravel())
            rin.E[v][:] = list(np.asarray(E[v], np.float64).ravel())
            rin.image[v] = imgs[v].ctypes.data
            rin.mask[v] = msks[v].ctypes.data
        rin.origin_width, rin.origin_height = ow, oh
        rin.lowest_width, rin.lowest_height, rin.pyr_levels = int(lowest_size[0]), int(lowest_size[1]), int(pyr_levels)
        W, H = rin.lowest_width << (pyr_levels - 1), rin.lowest_height << (pyr_levels - 1)
        rimg = [np.zeros((H, W, 3), np.uint8) for _ in range(2)] if want_images else [None, None]
        rmsk = [np.zeros((H, W), np.uint8) for _ in range(2)] if want_images else [None, None]
        if want_images:
            for v in range(2):
                rout.image[v] = rimg[v].ctypes.data
                rout.mask[v] = rmsk[v].ctypes.data
        self._chk(self._lib.rsm_rectify_pair(self._h, C.byref(rin), int(radius), C.c_double(ws), int(offset), int(verbose),
                                             C.byref(rout)))
        self._shape = (H, W)
        return dict(image=rimg, mask=rmsk, Q=np.array(rout.Q).reshape(4, 4), R_final=np.array(rout.R_final).reshape(3, 3),
                    T_final=np.array(rout.T_final), P=[np.array(rout.P[v]).reshape(3, 4) for v in range(2)], size=(W, H))

    def rect_map(self, A, R, newA, W, H):
        A, R, newA = (np.ascontiguousarray(x, np.float64) for x in (A, R, newA))
        m1 = np.zeros((H, W, 2), np.int16); m2 = np.zeros((H, W), np.uint16)
        self._chk(self._lib.rsm_stage_rect_map(self._h, _p(A), _p(R), _p(newA), W, H, _p(m1), _p(m2)))
        return m1, m2

    def remap_linear(self, src, map1, map2):
        src = _u8(src); Hs, Ws = src.shape[:2]; ch = 1 if src.ndim == 2 else src.shape[2]
        H, W = map2.shape
        m1 = np.ascontiguousarray(map1, np.int16); m2 = np.ascontiguousarray(map2, np.uint16)
        dst = np.zeros((H, W) + (() if src.ndim == 2 else (ch,)), np.uint8)
        self._chk(self._lib.rsm_stage_remap(self._h, _p(src), Ws, Hs, ch, _p(m1), _p(m2), W, H, _p(dst)))
        return dst

    def erode_ellipse_gray(self, mask, ksize):
        mask = _u8(mask); H, W = mask.shape
        dst = np.zeros_like(mask)
        self._chk(self._lib.rsm_stage_erode_gray(self._h, _p(mask), W, H, ksize, _p(dst)))
        return dst

    def run_pair(self):
        self._chk(self._lib.rsm_run_pair(self._h))

    def alloc_result(self, pinned=False, want_cloud=True, want_disparity=True) -> PairResult:
        """Result buffers for download_pair(into=...) sized for ANY cloud of the resident pair's size (W*H points: the
        point count depends on the data), page-locked when `pinned`."""
        H, W = self._shape
        new = (lambda shape, dt: host_empty(shape, dt)) if pinned else (lambda shape, dt: np.zeros(shape, dt))
        d = [new((H, W), np.float64), new((H, W), np.float64)] if want_disparity else [None, None]
        cap = W * H if want_cloud else 0
        xyz_buf, bgr_buf = new((cap, 3), np.float64), new((cap, 3), np.uint8)
        res = PairResult(disparity=d, margin=[None, None], n_points=0, xyz=xyz_buf[:0], bgr=bgr_buf[:0], v_top=0)
        res._xyz_buf, res._bgr_buf = xyz_buf, bgr_buf
        return res

    def download_pair(self, want_cloud=True, want_disparity=True, pinned=False, into=None) -> PairResult:
        """pinned: the results land in page-locked arrays (host_empty) instead of pageable ones; into: a PairResult whose
        buffers are reused (no allocation) -- one from alloc_result() (capacity W*H points: fits every cloud) or from an
        earlier download (capacity = that cloud's size; a larger cloud raises RsmError, nothing is truncated).  n_points,
        v_top, margin and the xyz / bgr views of `into` are refreshed from this download."""
        H, W = self._shape
        if into is not None:
            xyz_buf = getattr(into, "_xyz_buf", None)
            bgr_buf = getattr(into, "_bgr_buf", None)
            if xyz_buf is None:
                xyz_buf, bgr_buf = into.xyz, into.bgr
            pout = PairOut()
            if want_disparity:
                for v in range(2):
                    dv = into.disparity[v]
                    if dv is None or dv.shape != (H, W) or dv.dtype != np.float64 or not dv.flags.c_contiguous:
                        raise RsmError(-1, "download_pair(into=): disparity[%d] must be a C-contiguous float64 %dx%d array" % (v, H, W))
                    pout.disparity[v] = dv.ctypes.data
            cap = int(xyz_buf.shape[0]) if want_cloud else 0
            if want_cloud:
                if bgr_buf.shape[0] != cap or xyz_buf.dtype != np.float64 or bgr_buf.dtype != np.uint8:
                    raise RsmError(-1, "download_pair(into=): xyz / bgr buffers disagree")
                pout.max_points = cap
                pout.xyz = xyz_buf.ctypes.data if cap else None
                pout.bgr = bgr_buf.ctypes.data if cap else None
            self._chk(self._lib.rsm_download_pair(self._h, C.byref(pout)))
            n = int(pout.n_points)
            if want_cloud and n > cap:
                raise RsmError(-1, "download_pair(into=): the cloud has %d points, the buffers hold %d (use Context.alloc_result())" % (n, cap))
            into.n_points, into.v_top = n, int(pout.v_top)
            into.margin = [pout.margin[0].astuple(), pout.margin[1].astuple()]
            if want_cloud:
                into._xyz_buf, into._bgr_buf = xyz_buf, bgr_buf
                into.xyz, into.bgr = xyz_buf[:n], bgr_buf[:n]
            return into
        new = (lambda shape, dt: host_empty(shape, dt)) if pinned else (lambda shape, dt: np.zeros(shape, dt))
        d = [new((H, W), np.float64), new((H, W), np.float64)] if want_disparity else [None, None]
        pout = PairOut()
        if want_disparity:
            pout.disparity[0] = d[0].ctypes.data
            pout.disparity[1] = d[1].ctypes.data
        # first call learns n_points, second copies exactly that many
        pout.max_points = 0
        self._chk(self._lib.rsm_download_pair(self._h, C.byref(pout)))
        n = int(pout.n_points)
        xyz = new((n, 3), np.float64) if want_cloud else np.zeros((n, 3), np.float64)
        bgr = new((n, 3), np.uint8) if want_cloud else np.zeros((n, 3), np.uint8)
        if want_cloud and n > 0:
            p2 = PairOut()
            p2.max_points = n
            p2.xyz = xyz.ctypes.data
            p2.bgr = bgr.ctypes.data
            self._chk(self._lib.rsm_download_pair(self._h, C.byref(p2)))
        return PairResult(disparity=d, margin=[pout.margin[0].astuple(), pout.margin[1].astuple()],
                          n_points=n, xyz=xyz, bgr=bgr, v_top=int(pout.v_top))

    POINT16 = np.dtype([("x", np.float32), ("y", np.float32), ("z", np.float32), ("b", np.uint8), ("g", np.uint8), ("r", np.uint8), ("pad", np.uint8)])

    def download_points16(self, pinned=False) -> np.ndarray:
        """The last cloud as rsm_point16 records (float xyz = InsertPoint's cast, CCloudOptimization.cpp:61, + BGR), packed on
        the GPU and downloaded through rsm_pair_out.points16: a structured array of n_points records."""
        n = self.n_points
        rec = (host_empty((max(n, 1),), self.POINT16) if pinned else np.zeros(max(n, 1), self.POINT16))
        pout = PairOut()
        pout.max_points = n
        pout.points16 = rec.ctypes.data
        self._chk(self._lib.rsm_download_pair(self._h, C.byref(pout)))
        return rec[:n]

    def filter_last_info(self) -> dict:
        """What the last filter_last_cloud[_host] did: whether the pixel-window pass ran, how many queries it left to the grid search."""
        v = (C.c_int64 * 4)()
        self._chk(self._lib.rsm_filter_last_info(self._h, v))
        w = (C.c_int64 * 2)(0, -1)
        if hasattr(self._lib, "rsm_filter_last_normals_info"):   # (absent from the older libraries the A/B scripts load)
            self._chk(self._lib.rsm_filter_last_normals_info(self._h, w))
        return dict(window=bool(v[0]), radius=int(v[0]), undecided=int(v[1]), points=int(v[2]), kept=int(v[3]), normals_window=int(w[0]), normals_need=int(w[1]))

    def filter_last_cloud_host(self, mean_k=100, std_mul=1.0, normal_radius=2.5, cam_center=(0.0, 0.0, 0.0), want_normals=True):
        """rsm_filter_last_cloud_host: the per-pair filter on the GPU, its output -- the surviving points as rsm_point16 records
        and their oriented normals (nx, ny, nz, curvature) -- downloaded.  Returns (records, normals or None, stats dict)."""
        n = self.n_points
        rec = np.zeros(max(n, 1), self.POINT16)
        nrm = np.zeros((max(n, 1), 4), np.float32) if want_normals else None
        m = C.c_int64()
        st = (C.c_double * 4)()
        prm = self._filter_params(mean_k, std_mul, normal_radius, cam_center)
        self._chk(self._lib.rsm_filter_last_cloud_host(self._h, C.byref(prm), _p(rec), _p(nrm) if want_normals else None,
                                                       C.c_int64(n), C.byref(m), st))
        k = int(m.value)
        return rec[:k], (nrm[:k] if want_normals else None), dict(mean=st[0], stddev=st[1], threshold=st[2], exhaustive=int(st[3]))

    def match_pair(self, cfg, want_cloud=True) -> PairResult:
        self.upload_pair(cfg)
        self.run_pair()
        return self.download_pair(want_cloud=want_cloud)

    def result_device(self):
        """(disparity0_ptr, disparity1_ptr, n_points, xyz_ptr, bgr_ptr) device addresses of the last run."""
        d0, d1, xyz, bgr = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_void_p()
        n = C.c_int64()
        self._chk(self._lib.rsm_result_device(self._h, C.byref(d0), C.byref(d1), C.byref(n), C.byref(xyz), C.byref(bgr)))
        return d0.value, d1.value, int(n.value), xyz.value, bgr.value

    def export_cloud_device(self, xyz_ptr, bgr_ptr, max_points):
        """D2D copy of the last cloud into caller-owned device buffers (addresses, e.g. tensor.data_ptr())."""
        self._chk(self._lib.rsm_export_cloud_device(self._h, C.c_void_p(xyz_ptr or None), C.c_void_p(bgr_ptr or None),
                                                    C.c_int64(max_points)))

    def pack_cloud16(self, dst_ptr, max_points) -> int:
        """rsm_pack_cloud16: the last cloud as 16-byte point records (float xyz + BGR) into a caller-owned device
        buffer (address); returns the number of records written."""
        n = C.c_int64()
        self._chk(self._lib.rsm_pack_cloud16(self._h, C.c_void_p(dst_ptr or None), C.c_int64(max_points), C.byref(n)))
        return int(n.value)

    # ---- per-pair cloud filter (CCloudOptimization::filter, CloudOptimization/CCloudOptimization.cpp:82-121) ----
    @staticmethod
    def _filter_params(mean_k, std_mul, normal_radius, cam_center):
        prm = FilterParams()
        prm.sor_mean_k, prm.sor_std_mul, prm.normal_radius = int(mean_k), float(std_mul), float(normal_radius)
        prm.cam_center[:] = [float(v) for v in np.asarray(cam_center, np.float64).ravel()[:3]]
        return prm

    def filter_cloud(self, xyz, mean_k=100, std_mul=1.0, normal_radius=2.5, cam_center=(0.0, 0.0, 0.0)):
        """StatisticalOutlierRemoval + radius-search normals turned toward cam_center on a host cloud (n x 3, cast to
        float32 as InsertPoint does).  Returns (kept_index int32 [m], normals float32 [m,4] = nx, ny, nz, curvature,
        stats dict)."""
        xyz = np.ascontiguousarray(xyz, np.float32).reshape(-1, 3)
        n = len(xyz)
        kept = np.zeros(max(n, 1), np.int32)
        nrm = np.zeros((max(n, 1), 4), np.float32)
        m = C.c_int64()
        st = (C.c_double * 4)()
        prm = self._filter_params(mean_k, std_mul, normal_radius, cam_center)
        self._chk(self._lib.rsm_filter_cloud(self._h, _p(xyz), C.c_int64(n), C.byref(prm), _p(kept), _p(nrm), C.byref(m), st))
        return kept[:m.value].copy(), nrm[:m.value].copy(), dict(mean=st[0], stddev=st[1], threshold=st[2], exhaustive=int(st[3]))

    def filter_last_cloud(self, points_ptr, normals_ptr, max_points, mean_k=100, std_mul=1.0, normal_radius=2.5,
                          cam_center=(0.0, 0.0, 0.0)):
        """The same on the last run's cloud without leaving the GPU: surviving points as 16-byte records and their
        normals into caller-owned device buffers (addresses).  Returns (n_kept, stats dict)."""
        m = C.c_int64()
        st = (C.c_double * 4)()
        prm = self._filter_params(mean_k, std_mul, normal_radius, cam_center)
        self._chk(self._lib.rsm_filter_last_cloud(self._h, C.byref(prm), C.c_void_p(points_ptr or None), C.c_void_p(normals_ptr or None),
                                                  C.c_int64(max_points), C.byref(m), st))
        return int(m.value), dict(mean=st[0], stddev=st[1], threshold=st[2], exhaustive=int(st[3]))

    @property
    def n_points(self):
        return self.result_device()[2]

    def set_option(self, name: str, value: int):
        self._chk(self._lib.rsm_set_option(self._h, name.encode(), C.c_longlong(value)))

    # ---- measurement -----------------------------------------------------------------------------
    def profile_enable(self, on=True):
        """True / 1: events around every stage and every 8th launch of the dominant kernel; 2: the latter only."""
        self._chk(self._lib.rsm_profile_enable(self._h, int(on)))

    def profile_get(self):
        n = self._lib.rsm_profile_stage_count()
        ms = (C.c_double * n)()
        launches = (C.c_int64 * n)()
        byt = (C.c_double * n)()
        self._chk(self._lib.rsm_profile_get(self._h, ms, launches, byt))
        return {self._lib.rsm_profile_stage_name(i).decode(): {"ms": ms[i], "launches": int(launches[i]), "bytes": byt[i]}
                for i in range(n)}

    def bench_ncc(self, W, H, r, cands, iters=5) -> float:
        ms = C.c_double()
        self._chk(self._lib.rsm_bench_ncc(self._h, W, H, r, cands, iters, C.byref(ms)))
        return ms.value

    # ---- per-stage entry points (parity tests) -----------------------------------------------------
    def find_margin(self, mask, r):
        mask = _u8(mask); H, W = mask.shape
        m = Boundary()
        self._chk(self._lib.rsm_stage_find_margin(self._h, _p(mask), W, H, r, C.byref(m)))
        return m

    def pyr_down(self, src):
        src = _u8(src); H, W = src.shape[:2]
        ch = 1 if src.ndim == 2 else src.shape[2]
        dst = np.zeros(((H + 1) // 2, (W + 1) // 2) + (() if src.ndim == 2 else (ch,)), np.uint8)
        self._chk(self._lib.rsm_stage_pyr_down(self._h, _p(src), W, H, ch, _p(dst)))
        return dst

    def erode_ellipse_is255(self, mask, ksize):
        mask = _u8(mask); H, W = mask.shape
        dst = np.zeros_like(mask)
        self._chk(self._lib.rsm_stage_erode_ellipse(self._h, _p(mask), W, H, ksize, _p(dst)))
        return dst

    def initial_match(self, img_own, img_oth, mask_own, mask_oth, r, offset, own, oth, parent=None):
        img_own, img_oth, mask_own, mask_oth = map(_u8, (img_own, img_oth, mask_own, mask_oth))
        H, W = mask_own.shape
        d = np.zeros((H, W), np.int16)
        if parent is None:
            pp, Wp, Hp = None, 0, 0
        else:
            parent = np.ascontiguousarray(parent, np.float64)
            Hp, Wp = parent.shape
            pp = _p(parent)
        self._chk(self._lib.rsm_stage_initial_match(self._h, _p(img_own), _p(img_oth), _p(mask_own), _p(mask_oth),
                                                    W, H, r, offset, C.byref(_bd(own)), C.byref(_bd(oth)),
                                                    pp, Wp, Hp, _p(d)))
        return d

    def smooth_constraint(self, disp, own):
        d = np.array(disp, dtype=np.int16, order="C"); H, W = d.shape
        self._chk(self._lib.rsm_stage_smooth(self._h, _p(d), W, H, C.byref(_bd(own))))
        return d

    def order_constraint(self, disp, own):
        d = np.array(disp, dtype=np.int16, order="C"); H, W = d.shape
        self._chk(self._lib.rsm_stage_order(self._h, _p(d), W, H, C.byref(_bd(own))))
        return d

    def uniqueness_pass(self, p, q, own, oth):
        if np.asarray(p).dtype == np.float64:
            p = np.array(p, dtype=np.float64, order="C"); q = np.ascontiguousarray(q, np.float64)
            fn = self._lib.rsm_stage_uniqueness_pass_f64
        else:
            p = np.array(p, dtype=np.int16, order="C"); q = np.ascontiguousarray(q, np.int16)
            fn = self._lib.rsm_stage_uniqueness_pass_s16
        H, W = p.shape
        self._chk(fn(self._h, _p(p), _p(q), W, H, C.byref(_bd(own)), C.byref(_bd(oth))))
        return p

    def uniqueness(self, d0, d1, m0, m1):
        """UniquenessContraint<T> (.cpp:450-461): three passes."""
        d0 = self.uniqueness_pass(d0, d1, m0, m1)
        d1 = self.uniqueness_pass(d1, d0, m1, m0)
        d0 = self.uniqueness_pass(d0, d1, m0, m1)
        return d0, d1

    def set_boundary_smooth(self, disp, mask_own, own, oth):
        d = np.ascontiguousarray(disp, np.int16); mask_own = _u8(mask_own); H, W = d.shape
        BL = np.zeros((H, W), np.int16); BR = np.zeros((H, W), np.int16)
        st = self._lib.rsm_stage_set_boundary(self._h, _p(d), _p(mask_own), W, H, C.byref(_bd(own)),
                                              C.byref(_bd(oth)), _p(BL), _p(BR))
        if st not in (0, _lib.RSM_E_DEGENERATE_MARGIN):
            self._chk(st)
        return st, BL, BR

    def rematch(self, img_own, img_oth, mask_own, mask_oth, r, own, oth, disp):
        img_own, img_oth, mask_own, mask_oth = map(_u8, (img_own, img_oth, mask_own, mask_oth))
        d = np.array(disp, dtype=np.int16, order="C"); H, W = d.shape
        st = self._lib.rsm_stage_rematch(self._h, _p(img_own), _p(img_oth), _p(mask_own), _p(mask_oth), W, H, r,
                                         C.byref(_bd(own)), C.byref(_bd(oth)), _p(d))
        if st not in (0, _lib.RSM_E_DEGENERATE_MARGIN):
            self._chk(st)
        return st, d

    def median_filter(self, disp, mask_own, own):
        d = np.array(disp, dtype=np.int16, order="C"); mask_own = _u8(mask_own); H, W = d.shape
        self._chk(self._lib.rsm_stage_median(self._h, _p(d), _p(mask_own), W, H, C.byref(_bd(own))))
        return d

    def disparity_refine(self, disp, img_own, img_oth, iterations, ws, own):
        d = np.ascontiguousarray(disp, np.int16); img_own = _u8(img_own); img_oth = _u8(img_oth)
        H, W = d.shape
        out = np.zeros((H, W), np.float64)
        self._chk(self._lib.rsm_stage_refine(self._h, _p(d), _p(img_own), _p(img_oth), W, H, iterations,
                                             C.c_double(ws), C.byref(_bd(own)), _p(out)))
        return out

    def exp_neg(self, t, small_form: bool = False):
        """The specified exp(-t) of DisparityRefine's smoothness weights, evaluated on the device (small_form: through the
        time-skewed kernel's common-path form for arguments below 512)."""
        t = np.ascontiguousarray(t, np.float64).ravel()
        out = np.zeros(t.shape, np.float64)
        fn = self._lib.rsm_stage_exp_neg_small if small_form else self._lib.rsm_stage_exp_neg
        self._chk(fn(self._h, _p(t), C.c_int64(t.size), _p(out)))
        return out

    def refine_xi(self, img_own, img_oth, form: int = 0):
        """DisparityRefine's matching costs xi (CStereoMatching.cpp:624-629) as the device computes them: array [3, H-2, W-2, W-2],
        [c, y-1, x-1, col] = xi(x, y, col + c); form 0 / 1 / 2 = the first sweep's / lane-per-miss / four-lanes-per-miss routine."""
        img_own, img_oth = _u8(img_own), _u8(img_oth)
        H, W = img_own.shape[:2]
        out = np.zeros((3, H - 2, W - 2, W - 2), np.float64)
        self._chk(self._lib.rsm_stage_refine_xi(self._h, _p(img_own), _p(img_oth), W, H, int(form), _p(out)))
        return out

    def div_unscaled(self, a, b):
        """(q_fast, q_ieee): the time-skewed refine kernel's division without operand scaling beside the device's a / b."""
        a = np.ascontiguousarray(a, np.float64).ravel()
        b = np.ascontiguousarray(b, np.float64).ravel()
        assert a.shape == b.shape
        qf = np.zeros(a.shape, np.float64)
        qi = np.zeros(a.shape, np.float64)
        self._chk(self._lib.rsm_stage_div_unscaled(self._h, _p(a), _p(b), C.c_int64(a.size), _p(qf), _p(qi)))
        return qf, qi

    def sqrt_check(self, first_bits, n):
        """How many of the n floats with bit patterns first_bits .. first_bits + n - 1 the cloud filter's trimmed sqrtf gets wrong."""
        m = C.c_int64()
        self._chk(self._lib.rsm_stage_sqrt_check(self._h, C.c_uint32(int(first_bits)), C.c_int64(int(n)), C.byref(m)))
        return int(m.value)

    def disparity_to_cloud(self, disp, mask_org, img_own, Q, scale, R, T, own):
        d = np.ascontiguousarray(disp, np.float64); mask_org = _u8(mask_org); img_own = _u8(img_own)
        H, W = d.shape
        Q = np.ascontiguousarray(Q, np.float64); R = np.ascontiguousarray(R, np.float64)
        T = np.ascontiguousarray(T, np.float64)
        cap = W * H
        xyz = np.zeros((cap, 3), np.float64); bgr = np.zeros((cap, 3), np.uint8)
        n = C.c_int64()
        self._chk(self._lib.rsm_stage_cloud(self._h, _p(d), _p(mask_org), _p(img_own), W, H, _p(Q), C.c_double(scale),
                                            _p(R), _p(T), C.byref(_bd(own)), _p(xyz), _p(bgr), C.c_int64(cap),
                                            C.byref(n)))
        return xyz[:n.value].copy(), bgr[:n.value].copy()


def run_pairs(ctxs, repeats=1):
    """rsm_run_pairs[_repeat]: rsm_run_pair on several DIFFERENT contexts concurrently (pairs already resident),
    each context `repeats` times back to back."""
    lib = _lib.load()
    arr = (C.c_void_p * len(ctxs))(*[c._h for c in ctxs])
    st = lib.rsm_run_pairs_repeat(arr, len(ctxs), int(repeats))
    if st != 0:
        msgs = [(lib.rsm_last_error(c._h) or b"").decode() for c in ctxs]
        raise RsmError(st, "; ".join(m for m in msgs if m))


def _pairs_io(cfgs, want_cloud, want_disparity, pinned):
    """rsm_pair_in / rsm_pair_out arrays + the host buffers behind them for a list of pair configs."""
    n = len(cfgs)
    full = (lambda shape, v, dt=np.float64: _filled(host_empty(shape, dt), v)) if pinned else (lambda shape, v, dt=np.float64: np.full(shape, v, dt))
    ins = (PairIn * max(n, 1))()
    outs = (PairOut * max(n, 1))()
    keep, bufs = [], []
    for p, cfg in enumerate(cfgs):
        pin, k = Context._pair_in(cfg)
        ins[p] = pin
        keep.append(k)
        H, W = cfg.height, cfg.width
        d = [full((H, W), 0.0), full((H, W), 0.0)] if want_disparity else [None, None]   # full(): pages touched now
        xyz = full((W * H, 3), 0.0) if want_cloud else None
        bgr = full((W * H, 3), 0, np.uint8) if want_cloud else None
        if want_disparity:
            outs[p].disparity[0] = d[0].ctypes.data
            outs[p].disparity[1] = d[1].ctypes.data
        if want_cloud:
            outs[p].max_points = W * H
            outs[p].xyz = xyz.ctypes.data
            outs[p].bgr = bgr.ctypes.data
        bufs.append((d, xyz, bgr))
    return ins, outs, keep, bufs


def _pairs_results(n, outs, bufs, status, want_cloud):
    res = []
    for p in range(n):
        if status[p] != 0:
            res.append(None)
            continue
        d, xyz, bgr = bufs[p]
        m = int(outs[p].n_points)
        res.append(PairResult(disparity=d, margin=[outs[p].margin[0].astuple(), outs[p].margin[1].astuple()], n_points=m,
                              xyz=xyz[:m] if want_cloud else np.zeros((0, 3)),
                              bgr=bgr[:m] if want_cloud else np.zeros((0, 3), np.uint8), v_top=int(outs[p].v_top)))
    return res


def match_pairs(ctxs, cfgs, want_cloud=True, want_disparity=True, timing=None, pinned=False):
    """rsm_match_pairs: the pair loop of MatchAllLayer (.cpp:17-33) for a list of pair configs over a pool of
    contexts (same or different GPUs), pairs in flight together.  Returns (results in pair order, statuses).
    timing (optional dict) receives "call_s": seconds inside the C call alone (output buffers pre-faulted, the
    slicing of the results outside).  pinned: the output buffers are page-locked (host_empty)."""
    lib = _lib.load()
    n = len(cfgs)
    ins, outs, keep, bufs = _pairs_io(cfgs, want_cloud, want_disparity, pinned)
    status = (C.c_int * max(n, 1))()
    arr = (C.c_void_p * len(ctxs))(*[c._h for c in ctxs])
    import time as _time
    t0 = _time.perf_counter()
    lib.rsm_match_pairs(arr, len(ctxs), ins, outs, n, status)
    if timing is not None:
        timing["call_s"] = _time.perf_counter() - t0
    return _pairs_results(n, outs, bufs, status, want_cloud), list(status)[:n]


def match_pairs_multi_gpu(cfgs, n_gpus=0, pairs_in_flight=2, want_cloud=True, want_disparity=True):
    """rsm_match_pairs_multi_gpu: the same loop sharded over the GPUs of this node from ONE process; the
    library creates and destroys its own contexts (context i on GPU i % n_gpus, `pairs_in_flight` per GPU).
    n_gpus = 0: every visible GPU.  Returns (results in pair order -- None for a failed pair --, statuses, return code)."""
    lib = _lib.load()
    n = len(cfgs)
    ins, outs, keep, bufs = _pairs_io(cfgs, want_cloud, want_disparity, False)
    status = (C.c_int * max(n, 1))()
    rc = lib.rsm_match_pairs_multi_gpu(ins, n, int(n_gpus), int(pairs_in_flight), outs, status)
    return _pairs_results(n, outs, bufs, status, want_cloud), list(status)[:n], int(rc)


# ---------------------------------------------------------------------------------------------------
# Mirror of the reference's data / matching classes
# ---------------------------------------------------------------------------------------------------
@dataclass
class Camera:
    """struct camera (CManageData.h:16-26), the fields this path touches."""
    camID: int = 0
    image: np.ndarray | None = None   # rectified top-level BGR (what Rectify stores, .cpp:154)
    mask: np.ndarray | None = None    # rectified + eroded mask (.cpp:156-158)
    P: np.ndarray | None = None       # 3x4 projection (.cpp:143-145), carried through untouched
    bound: tuple | None = None        # written by MatchAllLayer (.cpp:27-28)
    image_name: str = ""
    mask_name: str = ""
    MatIntrinsics: np.ndarray | None = None   # 3x3 (CManageData.cpp:59)
    MatExtrinsics: np.ndarray | None = None   # 3x4 (CManageData.cpp:60)
    CamCenter: np.ndarray | None = None       # -R^T t as float32 (CManageData.cpp:61-62)


@dataclass
class ManageData:
    """CManageData (CManageData.h:28-61), the members MatchAllLayer reads."""
    cam: list = field(default_factory=list)          # cam[pair][0..1] -> Camera
    m_PyrmNum: int = 4
    m_LowestLevelSize: tuple = (160, 240)            # (width, height)
    m_OriginSize: tuple = (0, 0)                     # (width, height)
    isoutput: int = 0
    rectified: list = field(default_factory=list)    # per pair: dict(Q=4x4, R_final=3x3, T_final=3)

    @property
    def m_CampairNum(self):
        return len(self.cam)


class _PairCfg:
    pass


class StereoMatching:
    """CStereoMatching (CStereoMatching.h:35-71) on MI355X.

    Init(data, CloudOptimization, radii=2, ws=0.5, disparity_offset=2) and MatchAllLayer() keep the
    reference's signatures; per point `CloudOptimization.InsertPoint(point3)` is called in the
    reference's row-major order (or `InsertPoints(xyz)` once per pair when the sink offers it),
    then `CloudOptimization.filter(CamPair)` (.cpp:29-31).
    """

    def __init__(self, device: int = 0):
        self._ctx = Context(device)
        self.m_data = None
        self.m_CloudOptimization = None
        self.MatchBlockRadius = 2
        self.m_ws = 0.5
        self.m_offset = 2
        self.Q = None
        self.R_final = None
        self.T_final = None
        self.margin = [None, None]
        self.Verbose = 1
        self.disparity = [None, None]   # last pair's fp64 disparity maps (the reference keeps them local)

    def Init(self, data, CloudOptimization=None, radii=2, ws=0.5, disparity_offset=2):
        self.m_data = data
        self.m_CloudOptimization = CloudOptimization
        self.MatchBlockRadius = radii
        self.m_ws = ws
        self.m_offset = disparity_offset
        self.Verbose = 1

    def MatchAllLayer(self):
        data = self.m_data
        top = 1 << (data.m_PyrmNum - 1)
        W, H = data.m_LowestLevelSize[0] * top, data.m_LowestLevelSize[1] * top  # .cpp:120
        for CamPair in range(data.m_CampairNum):
            cams = data.cam[CamPair]
            if self.Verbose >= 1:
                print("processing pair %d: cam %d and cam %d..." % (CamPair + 1, cams[0].camID, cams[1].camID))
            pre_rectified = CamPair < len(data.rectified) and data.rectified[CamPair] is not None
            if not pre_rectified:
                # Rectify(CamPair, Q), .cpp:20 / :117-168: raw images + calibration -> rectified pair on the GPU
                from . import config as _cfg
                if self.Verbose >= 1:
                    print("\trectifying...")
                raw, rawm = [], []
                for j in range(2):
                    img = getattr(cams[j], "raw_image", None)
                    if img is None:
                        img = _cfg.imread_bgr(cams[j].image_name)
                    if img is None:
                        print("read image %s error" % cams[j].image_name)   # .cpp:147-151: silent return
                        return
                    msk = getattr(cams[j], "raw_mask", None)
                    if msk is None:
                        msk = _cfg.imread_gray(cams[j].mask_name)
                    if msk is None:
                        print("read image %s error" % cams[j].mask_name)
                        return
                    raw.append(img)
                    rawm.append(msk)
                rect = self._ctx.rectify_pair([cams[0].MatIntrinsics, cams[1].MatIntrinsics],
                                              [cams[0].MatExtrinsics, cams[1].MatExtrinsics], data.m_OriginSize,
                                              data.m_LowestLevelSize, data.m_PyrmNum, raw, rawm)
                for j in range(2):
                    cams[j].image, cams[j].mask, cams[j].P = rect["image"][j], rect["mask"][j], rect["P"][j]
                    if getattr(data, "isoutput", 0):   # .cpp:159-166: "%d_%d.jpg" of the rectified image
                        _cfg.imwrite_bgr("%d_%d.jpg" % (CamPair, cams[j].camID), cams[j].image)
            else:
                if cams[0].image is None or cams[1].image is None:
                    print("read image %s error" % (cams[0].image_name or cams[1].image_name))
                    return
                rect = data.rectified[CamPair]
            self.Q = np.asarray(rect["Q"], np.float64)
            self.R_final = np.asarray(rect["R_final"], np.float64)
            self.T_final = np.asarray(rect["T_final"], np.float64)
            cfg = _PairCfg()
            cfg.width, cfg.height, cfg.pyr_levels = W, H, data.m_PyrmNum
            cfg.radius, cfg.ws, cfg.offset = self.MatchBlockRadius, self.m_ws, self.m_offset
            cfg.origin_width = data.m_OriginSize[0] or W
            cfg.image = [cams[0].image, cams[1].image]
            cfg.mask = [cams[0].mask, cams[1].mask]
            cfg.Q, cfg.R_final, cfg.T_final = self.Q, self.R_final, self.T_final
            cfg.verbose = self.Verbose
            res = self._ctx.match_pair(cfg)
            self.margin = [res.margin[0], res.margin[1]]
            cams[0].bound = res.margin[0]     # .cpp:27-28
            cams[1].bound = res.margin[1]
            self.disparity = res.disparity
            if self.Verbose >= 1:
                print("\tconverting disparity to cloud %d..." % CamPair)
            if getattr(data, "isoutput", 0):   # .cpp:707-730,753-757: cloud%d.ply in the working directory
                write_ply("cloud%d.ply" % CamPair, res.xyz, res.bgr)
            sink = self.m_CloudOptimization
            if sink is not None:
                if hasattr(sink, "InsertPoints"):
                    sink.InsertPoints(res.xyz, res.bgr)
                else:
                    for p in res.xyz:
                        sink.InsertPoint(p)
                if hasattr(sink, "filter"):
                    sink.filter(CamPair)      # .cpp:31
            self.last_result = res


class CloudOptimization:
    """The part of CCloudOptimization (CloudOptimization/CCloudOptimization.cpp) that sits on the stereo path:
    Init (:40-57: only the per-pair filter's parameters are used), InsertPoint (:59-62), filter(idx) (:64-147: the
    StatisticalOutlierRemoval + NormalEstimation + normal flip of :82-121 on the GPU; the mesh / texture tooling
    after :123 is Windows executables and out of scope).  `cloud_normals` accumulates what the reference's global
    `*cloud_normals += *cloud_normal` (:123) does: per pair (xyz float32 [m,3], normals float32 [m,4])."""

    def __init__(self, ctx: Context | None = None, device: int = 0):
        self._ctx = ctx or Context(device)
        self._pts = []
        self._bgr = None
        self.cloud_normals = []
        self.cloud_bgr = []      # colours of the surviving points, per pair (imagePyrm[top][0], .cpp:756)
        self.stats = []

    def Init(self, sor_meank, sor_stdThres, outrem_neighbor, outrem_radius, mls_radius, ImageData, isdelete_=False):
        self.m_sor_meank, self.m_sor_stdThres, self.m_mls_radius = sor_meank, sor_stdThres, mls_radius
        self.m_ImageData = ImageData
        self.CamCenter = [np.asarray(c[0].CamCenter if c[0].CamCenter is not None else np.zeros(3), np.float32).ravel()
                          for c in ImageData.cam]

    def InsertPoint(self, p):
        self._pts.append(np.asarray(p, np.float64).ravel()[:3])

    def InsertPoints(self, xyz, bgr=None):
        self._pts = [np.asarray(xyz, np.float64).reshape(-1, 3)]
        self._bgr = None if bgr is None else np.asarray(bgr, np.uint8).reshape(-1, 3)

    def filter(self, idx):
        xyz = (np.concatenate([np.atleast_2d(p) for p in self._pts]) if self._pts else np.zeros((0, 3))).astype(np.float32)
        kept, nrm, st = self._ctx.filter_cloud(xyz, self.m_sor_meank, self.m_sor_stdThres, self.m_mls_radius, self.CamCenter[idx])
        self.cloud_normals.append((xyz[kept], nrm))
        self.cloud_bgr.append(self._bgr[kept] if self._bgr is not None and len(self._bgr) == len(xyz)
                              else np.zeros((len(kept), 3), np.uint8))
        self.stats.append(st)
        self._pts = []   # cloud_in->clear(), :145
        self._bgr = None
