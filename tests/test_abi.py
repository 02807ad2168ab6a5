"""The C-ABI library loads without a GPU, exports every symbol include/pseg.h declares, and fails
loudly (no CPU fallback) when no HIP device is visible.  No compute calls here."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "pseg.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(pseg_[a-z0-9_]+)\s*\(", hdr)))


def test_library_exports_every_declared_symbol():
    import pseg_amd
    L = pseg_amd.lib()
    declared = _declared_symbols()
    assert len(declared) >= 20
    for sym in declared:
        assert hasattr(L, sym), "libpseg.so does not export %s" % sym
    assert sorted(pseg_amd.EXPORTED_SYMBOLS) == declared
    assert L.pseg_abi_version() == 1


def test_rccl_entry_points_are_pinned_against_the_rccl_header():
    """pseg_dist.hip binds RCCL by dlopen and restates its ABI by hand (128-byte id, ncclFloat32 = 7, ncclSum = 0, five prototypes);
    where <rccl/rccl.h> exists at build time -- the ROCm image -- static_asserts compare every one of them with the header, and the
    library says whether they ran.  No GPU and no second rank needed: a mismatch is a build failure."""
    import pseg_amd
    L = pseg_amd.lib()
    pinned = L.pseg_rccl_abi_pinned()
    assert pinned == (1 if os.path.exists("/opt/rocm/include/rccl/rccl.h") else 0)


def test_oracle_library_is_separate_from_the_product():
    """Nothing of the oracle is linked into or imported by the product package."""
    import subprocess
    import pseg_amd
    out = subprocess.run(["nm", "-D", pseg_amd.lib_path()], capture_output=True, text=True).stdout
    assert "orc_" not in out
    pkg = os.path.join(ROOT, "page-segmentation_amd")
    for dp, _, fns in os.walk(pkg):
        for fn in fns:
            if fn.endswith((".py", ".hip", ".cpp", ".h")):
                src = open(os.path.join(dp, fn)).read()
                assert "import oracle" not in src and "from oracle" not in src and "liboracle" not in src, fn


def test_fails_loudly_without_a_gpu():
    import pseg_amd
    if pseg_amd.device_count() > 0:
        pytest.skip("a HIP device is present")
    with pytest.raises(pseg_amd.PsegError) as ei:
        pseg_amd.Engine("fcn_skip", 3)
    assert "no CPU fallback" in str(ei.value) or "HIP" in str(ei.value)
    import numpy as np
    with pytest.raises(pseg_amd.PsegError):
        pseg_amd.cc_vote(np.zeros((4, 4), np.int64), np.ones((4, 4), np.uint8), 3)
    with pytest.raises(pseg_amd.PsegError):
        pseg_amd.masks(np.zeros((4, 4), np.int64), np.ones((4, 4), np.uint8), np.zeros((3, 3), np.uint8))
    L = pseg_amd.lib()
    h = ctypes.c_void_p()
    assert L.pseg_create(0, 3, 1, 0, 1, ctypes.byref(h)) != 0
    assert L.pseg_create(99, 3, 1, 0, 1, ctypes.byref(h)) != 0
    assert len(L.pseg_last_error()) > 0


def test_library_links_no_vendor_gemm():
    """Every GEMM on the hot path is a hand-written MFMA kernel: libpseg.so must not pull in hipBLASLt / rocBLAS / MIOpen."""
    import subprocess
    import pseg_amd
    out = subprocess.run(["ldd", pseg_amd.lib_path()], capture_output=True, text=True).stdout.lower()
    assert not any(k in out for k in ("hipblas", "rocblas", "miopen")), out


def test_release_library_has_no_wrong_result_switches():
    """Timing ablations that produce wrong results (PSEG_DBG, PSEG_XM_DBG, PSEG_PP_NODMA, PSEG_PP_NOEPI, the in-kernel
    trace stamps) are compiled into the diagnostic build only (PSEG_DIAG_KNOB is a constant nullptr in the release
    library): the release .so does not even contain their names, so no environment can switch them on."""
    import pseg_amd
    blob = open(pseg_amd.lib_path(), "rb").read()
    for name in (b"PSEG_PP_NODMA", b"PSEG_PP_NOEPI", b"PSEG_XM_DBG", b"PSEG_DBG\0", b"PSEG_WS_TRACE", b"PSEG_TRACE\0"):
        assert name not in blob, name
    src = open(os.path.join(ROOT, "page-segmentation_amd", "csrc", "pseg_common.h")).read()
    assert "#define PSEG_DIAG_KNOB(name) ((const char*)nullptr)" in src


def test_reciprocal_pixel_index_is_exact_for_every_channel_count():
    """conv1x1_exact_kernel splits a flat element index e < 256*C into (pixel, channel) with a reciprocal multiply,
    floor(e * (2^32 // C + 1) / 2^32); exact for every per-source channel count the launch gate admits (<= 127).  (A
    20-bit reciprocal was wrong for C in 73..127 -- ADVICE round 2.)  conv_exact_mfma_kernel's staging keeps the 20-bit
    form under its own gate: C < 64 and e < 67 * 64."""
    import numpy as np
    for C in range(1, 128):
        e = np.arange(256 * C, dtype=np.uint64)
        inv = np.uint64((1 << 32) // C + 1)
        assert np.array_equal((e * inv) >> np.uint64(32), e // np.uint64(C)), C
    for C in range(1, 64):
        e = np.arange(67 * 64, dtype=np.uint64)
        inv = np.uint64((1 << 20) // C + 1)
        assert np.array_equal((e * inv) >> np.uint64(20), e // np.uint64(C)), C
