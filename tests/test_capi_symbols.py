"""CPU-side checks of the C-ABI boundary: the shared library builds/loads here (hipcc cross-compiles
for gfx950 without a GPU) and exports every function include/stylex_hip.h declares, and the ctypes
binding table mirrors the header.  No compute calls are made (there is no GPU in this container)."""
import ctypes
import os
import re

import pytest

import hip_backend

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "stylex_hip.h")


def declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(stylex_[a-z0-9_]+)\s*\(", src)))


def test_header_declares_the_kernel_set():
    names = declared_functions()
    for must in ("stylex_conv2d_fwd", "stylex_conv2d_bwd_data", "stylex_conv2d_bwd_weight", "stylex_pack_weight",
                 "stylex_upsample2x_bilinear_fwd", "stylex_upsample2x_bilinear_bwd", "stylex_blur3x3_reflect_fwd",
                 "stylex_blur3x3_reflect_bwd", "stylex_bias_act_fwd", "stylex_bias_act_bwd", "stylex_rowwise_sumsq"):
        assert must in names


def test_library_exports_every_declared_symbol():
    if not os.path.isfile(hip_backend.LIB_PATH):
        pytest.skip("libstylex_hip.so not built yet (python __graft_entry__.py build)")
    lib = ctypes.CDLL(hip_backend.LIB_PATH)
    for name in declared_functions():
        assert hasattr(lib, name), "missing export: " + name
    assert lib.stylex_version is not None
    lib.stylex_version.restype = ctypes.c_char_p
    assert b"gfx950" in lib.stylex_version()


def test_binding_table_matches_header():
    assert sorted(hip_backend.SIGNATURES.keys()) == declared_functions()
    hip_backend.load_library()  # binds all of them; raises on mismatch


def test_argument_validation_without_gpu():
    """Entry points reject bad arguments before touching the device."""
    lib = hip_backend.load_library()
    sh = (ctypes.c_int64 * 11)(1, 4, 4, 4, 4, 7, 7, 1, 3, 4, 4)  # 7x7 kernel: unsupported
    assert lib.stylex_conv2d_bwd_weight_workspace_bytes(sh) == -1
    assert lib.stylex_conv2d_fwd(None, None, None, sh, 0, None, 0, None, 0, None) == -1
    assert lib.stylex_timing_report(7, None, None, None, None) == -1
