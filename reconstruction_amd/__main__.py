"""python -m reconstruction_amd <config.yml> [--device N] [--out cloud.ply]

The command-line shape of the reference's main() (reconstruction/main.cpp:5-23) for the part this package covers:
CReconstrction::Init (configuration + calibration, CReconstruction.cpp:5-19) -> CStereoMatching::MatchAllLayer
(Rectify, pyramid, matching, refinement, cloud; on the MI355X) -> per pair CCloudOptimization::filter's outlier removal
and normals (--filter; CCloudOptimization.cpp:82-121, on the GPU) -> the merged point cloud as a PLY file.
CCloudOptimization::run (MLS, Poisson meshing, texture; main.cpp:19) is outside this package: feed the PLY to it.
Needs an MI355X; there is no CPU path.
"""
from __future__ import annotations

import argparse
import sys
import time

import numpy as np


class CloudSink:
    """Stands where CCloudOptimization stands in CStereoMatching::Init: collects what InsertPoint would receive."""

    def __init__(self):
        self.xyz, self.bgr = [], []

    def InsertPoints(self, xyz, bgr=None):   # one call per pair (the per-point InsertPoint order is preserved)
        self.xyz.append(np.asarray(xyz, np.float64))
        if bgr is not None:
            self.bgr.append(np.asarray(bgr, np.uint8))

    def filter(self, CamPair):               # no per-pair filter: the cloud as DisparityToCloud emits it
        pass


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(prog="python -m reconstruction_amd", description=__doc__.split("\n\n")[1])
    ap.add_argument("config", help="OpenCV-YAML configuration file (the reference's config_*.yml)")
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--out", default=None, help="PLY path (default: <outfilename of the configuration>.ply)")
    ap.add_argument("--radius", type=int, default=2, help="MatchBlockRadius (CReconstruction.cpp:17: 2)")
    ap.add_argument("--ws", type=float, default=0.03, help="smoothness weight (CReconstruction.cpp:17: 0.03)")
    ap.add_argument("--filter", action="store_true",
                    help="per-pair StatisticalOutlierRemoval (k=100, 1 sigma) as CCloudOptimization::filter (CReconstruction.cpp:18)")
    ap.add_argument("--mls-radius", type=float, default=2.5,
                    help="search radius of the per-pair normals, in scene units (m_mls_radius; CReconstruction.cpp:18 passes 2.5)")
    args = ap.parse_args(argv)

    from .config import load_config
    try:
        data, info = load_config(args.config)
    except (FileNotFoundError, ValueError) as e:
        print(e)                              # "cannot open file ..." (CReconstruction.cpp:9-13, CManageData.cpp:46-49)
        return 1
    from . import StereoMatching, write_ply   # loads the HIP library: fails loudly without it / without a GPU
    t0 = time.perf_counter()
    sm = StereoMatching(args.device)
    if args.filter:
        from . import CloudOptimization
        sink = CloudOptimization(sm._ctx)
        sink.Init(100, 1, 50, 2, args.mls_radius, data, False)    # CReconstruction.cpp:18
    else:
        sink = CloudSink()
    sm.Init(data, sink, args.radius, args.ws)
    sm.MatchAllLayer()
    print("Matching time: %.3f s" % (time.perf_counter() - t0))   # main.cpp:18
    if args.filter:
        if not sink.cloud_normals:
            print("no points")
            return 2
        xyz = np.concatenate([c[0] for c in sink.cloud_normals]).astype(np.float64)
        bgr = np.concatenate(sink.cloud_bgr)
        normals = np.concatenate([c[1] for c in sink.cloud_normals])
    else:
        if not sink.xyz:
            print("no points")
            return 2
        xyz = np.concatenate(sink.xyz)
        bgr = np.concatenate(sink.bgr) if sink.bgr else np.zeros((len(xyz), 3), np.uint8)
    # the configuration's outfilename already carries its extension ("%s%d.ply", BatchProcess/main.cpp:56)
    name = data.outfilename or "cloud"
    out = args.out or (name if name.lower().endswith(".ply") else name + ".ply")
    write_ply(out, xyz, bgr, normals if args.filter else None)
    print("%d points -> %s" % (len(xyz), out))
    return 0


if __name__ == "__main__":
    sys.exit(main())
