"""reconstruction_amd -- MI355X-native CStereoMatching hot path of seed93/reconstruction.

Only what the path needs: csrc/ (HIP kernels + the C ABI of include/rsm.h), a ctypes binding, the
host-side mirror of the reference's CStereoMatching call surface, synthetic inputs, and the
pair-sharding / RCCL cloud gather for multi-GPU runs.
"""
from .api import (Context, StereoMatching, ManageData, Camera, PairResult, Boundary, NOMATCH, RsmError,  # noqa: F401
                  write_ply, stereo_rectify, run_pairs, match_pairs, match_pairs_multi_gpu, CloudOptimization, host_empty)
from . import synth  # noqa: F401
