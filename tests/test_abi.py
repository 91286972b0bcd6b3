"""CPU-side checks of the drop-in boundary: the library loads and exports every symbol that
include/multiclust_hip.h declares; without a GPU the product path fails loudly (no CPU fallback)."""
import ctypes as C
import os
import re

import pytest

import multiclust_amd as mc
from multiclust_amd import hip

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    src = open(os.path.join(ROOT, "include", "multiclust_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(mchip_[a-z_0-9]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    lib = C.CDLL(mc.lib_path())
    syms = header_symbols()
    assert len(syms) >= 25
    for s in syms:
        assert hasattr(lib, s), s
    assert sorted(hip.ABI_SYMBOLS) == syms          # the ctypes view binds exactly the header


def test_abi_version_and_device_count():
    lib = mc.load()
    assert lib.mchip_abi_version() == hip.ABI_VERSION == 2
    n = C.c_int(-1)
    assert lib.mchip_device_count(C.byref(n)) == 0
    assert n.value >= 0


def test_no_gpu_fails_loudly():
    lib = mc.load()
    n = C.c_int(0)
    lib.mchip_device_count(C.byref(n))
    if n.value > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(mc.HipError):
        mc.Context(0)


def test_header_bindings_and_build_entry_agree_on_the_abi_version():
    """MCHIP_ABI_VERSION of the header = the version the Python bindings were written against = what __graft_entry__.build()
    checks after building (a bump of one without the others would fail the driver's build step)."""
    src = open(os.path.join(ROOT, "include", "multiclust_hip.h")).read()
    m = re.search(r"#define\s+MCHIP_ABI_VERSION\s+(\d+)", src)
    assert m and int(m.group(1)) == hip.ABI_VERSION
    entry = open(os.path.join(ROOT, "__graft_entry__.py")).read()
    assert "hip.ABI_VERSION" in entry
