"""The C-ABI library loads and exports every symbol include/dust_hip.h declares (no compute without a GPU)."""
import os
import re

import pytest

from dust_amd import _lib as L

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "dust_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(dust_(?:hip|vdb|vox|png|sky)_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree():
    assert declared_symbols() == sorted(L.SYMBOLS)


def test_library_exports_every_declared_symbol():
    lib = L.load()
    for name in declared_symbols():
        assert hasattr(lib, name), name


def test_device_entry_points_fail_loudly_without_gpu():
    lib = L.load()
    if lib.dust_hip_device_count() > 0:
        pytest.skip("GPU present")
    import ctypes as C
    h = C.c_void_p()
    st = lib.dust_hip_context_create(None, C.byref(h))
    assert st == L.ERR_NO_DEVICE
    assert b"no CPU fallback" in lib.dust_hip_last_error()
    from dust_amd import api
    with pytest.raises(L.DustError):
        api.Context()


def test_product_does_not_reference_the_oracle():
    """The oracle is test infrastructure: nothing under dust_amd/, include/ or bench.py's timed path links it."""
    for base in ("dust_amd", "include"):
        for dp, _, files in os.walk(os.path.join(ROOT, base)):
            for f in files:
                if f.endswith((".py", ".cpp", ".hpp", ".h", ".hip")):
                    text = open(os.path.join(dp, f)).read()
                    assert "oracle_lib" not in text and "liboracle" not in text and "orc_" not in text, os.path.join(dp, f)


def test_bench_touches_the_oracle_only_in_its_cpu_baseline_leg():
    text = open(os.path.join(ROOT, "bench.py")).read()
    leg = text.index("if not args.no_cpu_baseline")
    head = text[:leg]
    assert "import oracle_lib" not in head and "parity_util" not in head
    assert "import oracle_lib" in text[leg:]
    for tool in ("tools/kernel_sections.py", "tools/ray_stats.py", "tools/gi_timing.py", "tools/deep_tree_timing.py", "examples/render_castle.py"):
        t = open(os.path.join(ROOT, tool)).read()
        assert "oracle_lib" not in t and "parity_util" not in t, tool
