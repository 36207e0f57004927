"""The C-ABI library loads (no GPU needed) and exports every entry point include/svx.h declares."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared():
    text = open(os.path.join(ROOT, "include", "svx.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(svx_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_header():
    from svx import _lib
    lib = _lib.load()
    names = declared()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), "libsvx.so does not export %s" % n
    assert sorted(_lib.EXPORTS) == names
    assert lib.svx_version().decode().startswith("svx")


def test_host_side_helpers_without_gpu():
    from svx import _lib
    lib = _lib.load()
    assert lib.svx_num_levels(4096, 4096, 300) == 4
    assert lib.svx_num_levels(237, 217, 300) == 0
    assert lib.svx_num_levels(32768, 32768, 300) == 7
    assert lib.svx_knob_count(90, 80, 20000) == 7200
    assert lib.svx_knob_count(4096, 4096, 20000) == 20000
    # struct layout agreed between ctypes and the header (sizes only; offsets are natural alignment)
    assert ctypes.sizeof(_lib.AlignParams) == 4 * 3 + 4 * 256 + 4 * 4 + 4 + 8 + 2 * 4  # (+ search_mode, reserved0)
    assert ctypes.sizeof(_lib.Pair) == 8 * 2 + 4 * 4 + 8 * 8


def test_no_cpu_fallback():
    """Without a GPU the compute entry points raise instead of silently running elsewhere."""
    import numpy as np
    import pytest
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from svx.vecalign import dp_core
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        dp_core.dense_dp(np.zeros((2, 2), np.float32), 0.1)
