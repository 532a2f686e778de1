"""Builds librsm_mi355.so (HIP kernels + C ABI) in-tree with hipcc for gfx950."""
from __future__ import annotations

import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "librsm_mi355.so")


def build(force: bool = False, jobs: int = 8) -> str:
    csrc = os.path.join(HERE, "csrc")
    if force:
        subprocess.check_call(["make", "-s", "-C", csrc, "clean"])
    subprocess.check_call(["make", "-s", "-j%d" % jobs, "-C", csrc, "all"])
    if not os.path.exists(LIB):
        raise RuntimeError("librsm_mi355.so was not produced")
    return LIB


if __name__ == "__main__":
    print(build())
