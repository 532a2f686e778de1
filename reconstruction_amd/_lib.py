"""ctypes binding of librsm_mi355.so (include/rsm.h).

There is no CPU fallback: if the HIP library is missing or no MI355X is visible the product path
raises -- it never routes through the CPU oracle.
"""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "librsm_mi355.so")

RSM_OK = 0
RSM_E_INVALID = -1
RSM_E_DEGENERATE_MARGIN = -2
RSM_E_HIP = -3
RSM_E_NOMEM = -4
RSM_E_STATE = -5
RSM_E_COMM = -6
NOMATCH = -10000


class RsmError(RuntimeError):
    def __init__(self, code, msg=""):
        super().__init__("rsm error %d: %s" % (code, msg))
        self.code = code


class Boundary(C.Structure):
    """struct Boundary (CManageData.h:10-14)."""
    _fields_ = [("YL", C.c_int), ("YR", C.c_int), ("XL", C.c_int), ("XR", C.c_int),
                ("width", C.c_int), ("height", C.c_int)]

    def astuple(self):
        return (self.YL, self.YR, self.XL, self.XR, self.width, self.height)

    def __repr__(self):
        return "Boundary(YL=%d,YR=%d,XL=%d,XR=%d,w=%d,h=%d)" % self.astuple()


class PairIn(C.Structure):
    _fields_ = [("image", C.c_void_p * 2), ("mask", C.c_void_p * 2),
                ("width", C.c_int), ("height", C.c_int), ("pyr_levels", C.c_int),
                ("radius", C.c_int), ("ws", C.c_double), ("offset", C.c_int),
                ("origin_width", C.c_int), ("Q", C.c_double * 16), ("R_final", C.c_double * 9),
                ("T_final", C.c_double * 3), ("verbose", C.c_int)]


class PairOut(C.Structure):
    _fields_ = [("disparity", C.c_void_p * 2), ("margin", Boundary * 2),
                ("n_points", C.c_int64), ("max_points", C.c_int64),
                ("xyz", C.c_void_p), ("bgr", C.c_void_p), ("v_top", C.c_int64), ("points16", C.c_void_p)]


class FilterParams(C.Structure):
    _fields_ = [("sor_mean_k", C.c_int), ("sor_std_mul", C.c_double), ("normal_radius", C.c_double), ("cam_center", C.c_float * 3)]


class RectifyIn(C.Structure):
    _fields_ = [("K", (C.c_double * 9) * 2), ("E", (C.c_double * 12) * 2),
                ("origin_width", C.c_int), ("origin_height", C.c_int), ("lowest_width", C.c_int),
                ("lowest_height", C.c_int), ("pyr_levels", C.c_int),
                ("image", C.c_void_p * 2), ("mask", C.c_void_p * 2)]


class RectifyOut(C.Structure):
    _fields_ = [("Q", C.c_double * 16), ("R_final", C.c_double * 9), ("T_final", C.c_double * 3),
                ("P", (C.c_double * 12) * 2), ("width", C.c_int), ("height", C.c_int),
                ("image", C.c_void_p * 2), ("mask", C.c_void_p * 2)]


# every symbol include/rsm.h declares (tests/test_abi.py checks the header against this list)
EXPORTS = [
    "rsm_create", "rsm_destroy", "rsm_last_error", "rsm_version", "rsm_abi_version", "rsm_device_count", "rsm_filter_last_cloud_host", "rsm_filter_last_info", "rsm_filter_last_normals_info", "rsm_match_pair", "rsm_upload_pair",
    "rsm_upload_pair_device", "rsm_run_pair", "rsm_download_pair", "rsm_result_device", "rsm_export_cloud_device",
    "rsm_set_option", "rsm_profile_enable", "rsm_profile_stage_count", "rsm_profile_stage_name", "rsm_profile_get",
    "rsm_stage_find_margin", "rsm_stage_pyr_down", "rsm_stage_erode_ellipse", "rsm_stage_initial_match",
    "rsm_stage_smooth", "rsm_stage_order", "rsm_stage_uniqueness_pass_s16", "rsm_stage_uniqueness_pass_f64",
    "rsm_stage_set_boundary", "rsm_stage_rematch", "rsm_stage_median", "rsm_stage_refine", "rsm_stage_exp_neg", "rsm_stage_exp_neg_small", "rsm_stage_div_unscaled", "rsm_stage_sqrt_check", "rsm_stage_refine_xi", "rsm_stage_cloud",
    "rsm_bench_ncc", "rsm_write_ply", "rsm_write_ply16", "rsm_rectify_pair", "rsm_stereo_rectify", "rsm_stage_rect_map",
    "rsm_stage_remap", "rsm_stage_erode_gray", "rsm_run_pairs", "rsm_run_pairs_repeat", "rsm_match_pairs", "rsm_match_pairs_multi_gpu",
    "rsm_pack_cloud16", "rsm_comm_unique_id", "rsm_comm_create", "rsm_comm_destroy", "rsm_comm_last_error",
    "rsm_gather_clouds", "rsm_gather_counts", "rsm_gather_meta_fill", "rsm_gather_plan", "rsm_comm_create_transport",
    "rsm_filter_cloud", "rsm_filter_last_cloud", "rsm_host_alloc", "rsm_host_free", "rsm_host_register", "rsm_host_unregister",
]

_lib = None


def load():
    """Load the HIP library; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RsmError(RSM_E_STATE, "librsm_mi355.so not built: run `python -c 'import __graft_entry__ as g; "
                                    "g.build()'` (hipcc, gfx950); there is no CPU fallback")
    lib = C.CDLL(LIB_PATH)
    lib.rsm_last_error.restype = C.c_char_p
    lib.rsm_last_error.argtypes = [C.c_void_p]
    lib.rsm_version.restype = C.c_char_p
    abi = lib.rsm_abi_version() if hasattr(lib, "rsm_abi_version") else 1   # include/rsm.h: RSM_ABI_VERSION (PairOut below carries points16)
    if abi != 2 and os.environ.get("RSM_AB_OLD_LIBRARY") != "1":          # (the A/B scripts load round 5's library: same layouts, no version symbol)
        raise RsmError(RSM_E_STATE, "librsm_mi355.so has ABI %d, this binding ABI 2: rebuild the library" % abi)
    lib.rsm_profile_stage_name.restype = C.c_char_p
    lib.rsm_destroy.restype = None
    lib.rsm_host_alloc.restype = C.c_void_p
    lib.rsm_host_alloc.argtypes = [C.c_size_t]
    lib.rsm_host_free.restype = None
    lib.rsm_host_free.argtypes = [C.c_void_p]
    lib.rsm_host_register.argtypes = [C.c_void_p, C.c_size_t]
    lib.rsm_host_unregister.argtypes = [C.c_void_p]
    lib.rsm_destroy.argtypes = [C.c_void_p]
    lib.rsm_comm_last_error.restype = C.c_char_p
    lib.rsm_comm_last_error.argtypes = [C.c_void_p]
    lib.rsm_comm_destroy.restype = None
    lib.rsm_comm_destroy.argtypes = [C.c_void_p]
    _lib = lib
    return lib
