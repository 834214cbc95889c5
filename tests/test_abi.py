"""The shipped C-ABI library: builds for gfx950, loads, and exports every symbol include/sbk.h declares.
No compute calls (no GPU needed)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib_path():
    from speechbrain_amd.csrc.build import build

    return build(verbose=False)


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "sbk.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(sbk_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_exported(lib_path):
    lib = ctypes.CDLL(lib_path)
    names = declared_symbols()
    assert len(names) >= 15
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/sbk.h but not exported"
    lib.sbk_abi_version.restype = ctypes.c_int
    assert lib.sbk_abi_version() == 11


def test_binding_covers_header(lib_path):
    from speechbrain_amd import native

    import emu_utils

    emu_utils.detach()
    native.load()
    assert set(declared_symbols()) == set(native.EXPORTS)


def test_bad_arguments_are_reported_not_crashed(lib_path):
    lib = ctypes.CDLL(lib_path)
    lib.sbk_last_error.restype = ctypes.c_char_p
    rc = lib.sbk_gemm_nt_f32(None, 0, None, 0, None, None, 0, None, 0, 1, 1, 1, 0, ctypes.c_float(1.0), None, 0, None)
    assert rc == -22 and b"gemm" in lib.sbk_last_error()


def test_product_refuses_cpu_tensors(lib_path):
    """No CPU fallback: a CPU tensor on the product binding raises instead of computing."""
    import torch

    from speechbrain_amd import native

    import emu_utils

    emu_utils.detach()
    native.load()
    with pytest.raises(native.SbkError):
        native.gemm_nt(torch.zeros(4, 4), torch.zeros(4, 4))
