"""Builds csrc/libpseg.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
import os
import subprocess

_CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "csrc")


def build_library(force=False, jobs=4):
    if force:
        subprocess.check_call(["make", "-s", "-C", _CSRC, "clean"])
    subprocess.check_call(["make", "-s", "-C", _CSRC, "-j%d" % jobs, "libpseg.so"])
    return os.path.join(_CSRC, "libpseg.so")


def build_diag_library(jobs=4):
    """csrc/libpseg_diag.so: the same sources with -DPSEG_DIAG=1 (trace stamps, timing ablations, forced failures).  Never loaded by
    the product; tests/test_bf16_gpu.py uses it to force conv_sp_kernel's give-up path, tools/ for in-kernel traces."""
    subprocess.check_call(["make", "-s", "-C", _CSRC, "-j%d" % jobs, "libpseg_diag.so"])
    return os.path.join(_CSRC, "libpseg_diag.so")
