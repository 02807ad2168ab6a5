"""Builds csrc/libpseg.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
import os
import subprocess

_CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "csrc")


def build_library(force=False, jobs=4):
    if force:
        subprocess.check_call(["make", "-s", "-C", _CSRC, "clean"])
    subprocess.check_call(["make", "-s", "-C", _CSRC, "-j%d" % jobs, "libpseg.so"])
    return os.path.join(_CSRC, "libpseg.so")
