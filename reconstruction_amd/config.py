"""Config + calibration loader: what CReconstrction::Init / CManageData::Init read from the
OpenCV-YAML config (reconstruction/CReconstruction.cpp:8-14, CManageData.cpp:24-79; schema witnessed by
BatchProcess/main.cpp:53-72) and the image / mask files, without OpenCV.

  filepath, outfilename, isoutput, camera_calib_name, LowestLevelWidth, LowestLevelHeight, PyrmNum,
  imagelist, masklist (sequences indexed by camera id), camID (uint8 matrix, one row per pair)
  calibration file: intrinsic-<id> (3x3 fp64), extrinsic-<id> (3x4 fp64)

OpenCV's FileStorage YAML is YAML 1.0 with a `%YAML:1.0` directive and `!!opencv-matrix` mappings
(rows, cols, dt, data); both are handled here with PyYAML.  Images are read with Pillow and returned as BGR
(cv::imread's channel order).
"""
from __future__ import annotations

import os
import re

import numpy as np
import yaml

from .api import Camera, ManageData

_DT = {"u": np.uint8, "c": np.int8, "w": np.uint16, "s": np.int16, "i": np.int32, "f": np.float32, "d": np.float64}


class _Loader(yaml.SafeLoader):
    pass


def _opencv_matrix(loader, node):
    m = loader.construct_mapping(node, deep=True)
    dt = str(m["dt"])
    ch = int(dt[:-1]) if len(dt) > 1 else 1
    a = np.array(m["data"], dtype=_DT[dt[-1]])
    shape = (int(m["rows"]), int(m["cols"])) + ((ch,) if ch > 1 else ())
    return a.reshape(shape)


_Loader.add_constructor("tag:yaml.org,2002:opencv-matrix", _opencv_matrix)


def load_opencv_yaml(path: str) -> dict:
    """cv::FileStorage(path, READ) for the subset of YAML OpenCV 2.4 writes."""
    txt = open(path, "r").read()
    txt = re.sub(r"^%YAML[: ]1\.0\s*\n", "", txt)          # '%YAML:1.0' is not a valid YAML 1.1 directive
    txt = re.sub(r"^---\s*\n", "", txt)
    return yaml.load(txt, Loader=_Loader) or {}


def dump_opencv_yaml(path: str, data: dict) -> None:
    """Writer for tests and tools: the inverse of load_opencv_yaml (strings, ints, floats, string lists, matrices)."""
    rev = {np.dtype(v).str[1:]: k for k, v in _DT.items()}
    with open(path, "w") as f:
        f.write("%YAML:1.0\n")
        for k, v in data.items():
            if isinstance(v, np.ndarray):
                dt = rev[v.dtype.str[1:]]
                f.write("%s: !!opencv-matrix\n   rows: %d\n   cols: %d\n   dt: %s\n   data: [ %s ]\n" %
                        (k, v.shape[0], v.shape[1], dt, ", ".join(repr(x) for x in v.ravel().tolist())))
            elif isinstance(v, (list, tuple)):
                f.write("%s:\n" % k)
                for s in v:
                    f.write('   - "%s"\n' % str(s).replace("\\", "\\\\"))
            elif isinstance(v, str):
                f.write('%s: "%s"\n' % (k, v.replace("\\", "\\\\")))
            else:
                f.write("%s: %s\n" % (k, v))


def imread_bgr(path: str):
    """cv::imread(path) -> HxWx3 uint8 BGR, or None if unreadable (the reference then returns silently)."""
    try:
        from PIL import Image
        with Image.open(path) as im:
            rgb = np.asarray(im.convert("RGB"))
        return np.ascontiguousarray(rgb[:, :, ::-1])
    except Exception:
        return None


def imwrite_bgr(path: str, img) -> bool:
    """cv::imwrite(path, bgr): the format follows the extension (the reference dumps "%d_%d.jpg", .cpp:163-164)."""
    try:
        from PIL import Image
        Image.fromarray(np.ascontiguousarray(np.asarray(img)[:, :, ::-1])).save(path)
        return True
    except Exception:
        return False


def imread_gray(path: str):
    """cv::imread(path, CV_LOAD_IMAGE_GRAYSCALE) -> HxW uint8 or None."""
    try:
        from PIL import Image
        with Image.open(path) as im:
            if im.mode in ("L", "1", "P", "I;16"):
                return np.ascontiguousarray(np.asarray(im.convert("L")))
            rgb = np.asarray(im.convert("RGB")).astype(np.float64)
        # cvtColor BGR2GRAY: (R*4899 + G*9617 + B*1868 + 8192) >> 14
        g = (rgb[..., 0] * 4899 + rgb[..., 1] * 9617 + rgb[..., 2] * 1868 + 8192).astype(np.int64) >> 14
        return np.ascontiguousarray(g.astype(np.uint8))
    except Exception:
        return None


def load_config(configfile: str):
    """CReconstrction::Init + CManageData::Init. Returns (ManageData, info) or raises FileNotFoundError with the
    reference's message ('cannot open file ...')."""
    if not os.path.isfile(configfile):
        raise FileNotFoundError("cannot open file %s" % configfile)
    fs = load_opencv_yaml(configfile)
    filepath = str(fs.get("filepath", ""))
    calib_name = str(fs["camera_calib_name"])
    calib_path = filepath + calib_name
    if not os.path.isfile(calib_path):
        raise FileNotFoundError("cannot open file %s" % calib_name)          # CManageData.cpp:46-49
    calib = load_opencv_yaml(calib_path)
    camID = np.asarray(fs["camID"])
    imagelist = [str(s) for s in fs["imagelist"]]
    masklist = [str(s) for s in fs["masklist"]]
    data = ManageData(m_PyrmNum=int(fs["PyrmNum"]),
                      m_LowestLevelSize=(int(fs["LowestLevelWidth"]), int(fs["LowestLevelHeight"])),
                      isoutput=int(fs.get("isoutput", 0)))
    data.m_FilePath = filepath
    data.outfilename = str(fs.get("outfilename", ""))
    data.m_CameraNum = len(imagelist)
    for i in range(camID.shape[0]):
        pair = []
        for k in range(2):
            cid = int(camID[i, k])
            cam = Camera(camID=cid, image_name=filepath + imagelist[cid].replace("\\", os.sep),
                         mask_name=filepath + masklist[cid].replace("\\", os.sep))
            cam.MatIntrinsics = np.asarray(calib["intrinsic-%d" % cid], np.float64)   # CManageData.cpp:59
            cam.MatExtrinsics = np.asarray(calib["extrinsic-%d" % cid], np.float64)   # :60
            R, t = cam.MatExtrinsics[:, :3], cam.MatExtrinsics[:, 3]
            cam.CamCenter = (-R.T @ t).astype(np.float32)                              # :61-62
            pair.append(cam)
        data.cam.append(pair)
    m0 = imread_gray(filepath + masklist[0].replace("\\", os.sep))                     # :68-69
    if m0 is None:   # the reference would go on with an empty m_OriginSize and fail inside Rectify; say what is wrong
        raise ValueError("read image %s error (the first mask gives m_OriginSize, CManageData.cpp:68-69)"
                         % (filepath + masklist[0]))
    data.m_OriginSize = (m0.shape[1], m0.shape[0])
    return data, {"filepath": filepath, "config": fs}
