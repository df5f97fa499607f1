"""CPU-only: the C-ABI library loads and exports every symbol include/strotss_hip.h declares, and
the ctypes table in nn/_hip.py types exactly that set (no compute calls: there is no GPU here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "strotss_hip.h")


def declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(strotss_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_hot_path():
    syms = declared_symbols()
    for must in ("strotss_selfsim_fwd_bwd", "strotss_remd_cos_fwd_bwd", "strotss_moment_fwd_bwd",
                 "strotss_conv3x3_relu_fwd", "strotss_conv3x3_dgrad", "strotss_hypercol_gather",
                 "strotss_hypercol_scatter", "strotss_rmsprop_step", "strotss_resize_bilinear"):
        assert must in syms


def test_header_is_self_contained_c99(tmp_path):
    """The boundary is a C ABI: the header must compile on its own as plain C (what a cgo / JNI / ctypes-free
    binding would include), with no C++ and no HIP types in the signatures."""
    import shutil
    import subprocess
    cc = shutil.which("gcc") or shutil.which("cc")
    if cc is None:
        pytest.skip("no C compiler")
    src = tmp_path / "use_header.c"
    src.write_text('#include "strotss_hip.h"\nint (*probe)(void) = strotss_abi_version;\nint main(void) { return probe == 0 ? 1 : 0; }\n')
    out = subprocess.run([cc, "-std=c99", "-Wall", "-Werror", "-fsyntax-only", "-I", os.path.dirname(HEADER), str(src)],
                         capture_output=True, text=True)
    assert out.returncode == 0, out.stderr


def test_library_exports_every_declared_symbol():
    from nn import _hip
    if not os.path.exists(_hip.LIB_PATH):
        import __graft_entry__ as ge
        ge.build()
    lib = ctypes.CDLL(_hip.LIB_PATH)
    for s in declared_symbols():
        assert hasattr(lib, s), f"{s} declared in strotss_hip.h but not exported"
    assert sorted(_hip.SIGNATURES) == declared_symbols()
    typed = _hip.load_library()
    assert typed.strotss_abi_version() == _hip.ABI_VERSION == 8
    assert b"gfx950" in typed.strotss_build_info()


def test_product_path_fails_loudly_without_gpu():
    import torch
    from nn import _hip
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(_hip.StrotssHipError):
        _hip.lib()
