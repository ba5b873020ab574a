"""CPU tier: the C-ABI library builds/loads, exports every symbol of include/avm.h, struct layouts
match the ctypes mirror, and the product refuses to run without a GPU (no CPU fallback)."""
import ctypes as C
import os
import re

import pytest

from conftest import PKG, ROOT, mod


def test_library_exports_every_declared_symbol():
    lib_m = mod("lib")
    L = lib_m.lib()
    hdr = open(os.path.join(ROOT, "include", "avm.h")).read()
    declared = set(re.findall(r"\b(avm_[a-z_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    assert declared == set(lib_m.EXPORTS), declared ^ set(lib_m.EXPORTS)
    for name in declared:
        assert hasattr(L, name), name


def test_struct_sizes_match_ctypes_mirror(abi):
    L = mod("lib").lib()
    out = (C.c_int * 8)()
    n = L.avm_debug_struct_sizes(out)
    assert n == 7
    exp = [abi.Options, abi.WindowBatch, abi.PriorOut, abi.SolveSummary, abi.FselBatch, abi.FselOut, abi.Config]
    assert [out[i] for i in range(7)] == [C.sizeof(t) for t in exp]
    assert abi.SUMMARY_DTYPE.itemsize == C.sizeof(abi.SolveSummary)


def test_default_options_agree(abi, oracle):
    L = mod("lib").lib()
    a, b, c = abi.Options(), abi.Options(), abi.default_options()
    assert L.avm_default_options(C.byref(a)) == 0
    assert oracle.lib().avmo_default_options(C.byref(b)) == 0
    assert bytes(a) == bytes(b) == bytes(c)


def test_no_cpu_fallback_without_device():
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    lib_m = mod("lib")
    with pytest.raises(lib_m.AvmError):
        lib_m.Context(0)


def test_product_package_never_touches_the_oracle():
    """No file of the shipped package may import, include, link or call anything under oracle/."""
    banned = re.compile(r"oracle_py|libavm_oracle|avmo_|#include\s+\"[^\"]*oracle|import\s+oracle|from\s+oracle")
    # the package, the public headers (C ABI + the C++ host side) and the measurement / profiling scripts
    for top in (os.path.join(ROOT, PKG), os.path.join(ROOT, "include"), os.path.join(ROOT, "scripts")):
        for dirpath, _, files in os.walk(top):
            for f in files:
                if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h", ".sh", "Makefile")):
                    txt = open(os.path.join(dirpath, f)).read()
                    assert not banned.search(txt), (dirpath, f, banned.search(txt).group(0))


def test_avm_create_refuses_a_caller_built_against_another_header(abi):
    """avm_config::abi_version: the structs carry no size fields, so a host compiled against an older avm.h - avm_fsel_out had three
    members before min_gap - is stopped at avm_create(), before any call reads its structs.  Checked first: no device needed."""
    import ctypes as C

    L = __import__("importlib").import_module("anticipated-vins-mono_amd.lib").lib()
    assert L.avm_abi_version() == abi.AVM_ABI_VERSION
    for stale in (0, abi.AVM_ABI_VERSION - 1, abi.AVM_ABI_VERSION + 1):
        cfg = abi.Config()
        cfg.abi_version = stale
        h = C.c_void_p()
        assert L.avm_create(C.byref(cfg), C.byref(h)) == abi.AVM_ERR_INVALID and not h.value
