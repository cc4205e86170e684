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


_NULL_BATTERY = r'''
import sys, ctypes as C
sys.path.insert(0, ROOT)
from dust_amd import _lib as L
lib = L.load()
handles = {}
if MODE == "handles":   # valid handles first, everything else zero
    import numpy as np
    from dust_amd import api, synth
    ctx = api.Context(device=0)
    pal = synth.make_palette(1)
    model = api.Model(ctx, *api.flatten_model(np.array([[1, 1, 1, 3], [2, 1, 1, 4]], np.uint8), (8, 8, 8), pal), pal)
    scene = api.Scene(ctx)
    scene.add_instance(model, np.eye(3, 4, dtype=np.float32).reshape(12))
    scene.commit()
    pipe = api.StandardPipeline(ctx, 32, 16)
    handles = {"context": ctx._h, "model": model._h, "scene": scene._h, "pipeline": pipe._h, "gi": pipe._h}
for name, (res, args) in L.SYMBOLS.items():
    if name.endswith("_destroy") or name in ("dust_vox_free", "dust_vdb_pool_free"):
        if MODE == "handles":
            continue
    vals = []
    for i, a in enumerate(args):
        if a in (C.c_uint32, C.c_int32, C.c_int, C.c_uint64, C.c_size_t):
            vals.append(0)
        elif a is C.c_float:
            vals.append(0.0)
        elif i == 0 and a is C.c_void_p and MODE == "handles" and name.startswith("dust_hip_") and name.split("_")[2] in handles:
            vals.append(handles[name.split("_")[2]])
        else:
            vals.append(None)
    print("CALL", name, flush=True)
    r = getattr(lib, name)(*vals)
    print("RET", name, r if res is C.c_int else "-", flush=True)
print("BATTERY_DONE", flush=True)
'''


def _battery(mode):
    import subprocess
    import sys
    r = subprocess.run([sys.executable, "-c", f"ROOT={ROOT!r}\nMODE={mode!r}\n" + _NULL_BATTERY], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       text=True, timeout=600)
    calls = [l.split()[1] for l in r.stdout.splitlines() if l.startswith("CALL ")]
    assert r.returncode == 0 and "BATTERY_DONE" in r.stdout, f"died in {calls[-1] if calls else '?'} (exit {r.returncode}): {r.stderr[-1500:]}"
    return {l.split()[1]: l.split()[2] for l in r.stdout.splitlines() if l.startswith("RET ")}


def test_null_arguments_never_cross_the_abi_as_a_crash():
    """every function returns a status, nothing crosses the boundary as a fault. Every exported function called with
    nothing but null pointers and zeroes (its own process: a fault is a failed test that names the function): each comes back, and
    each status-returning one that takes arguments refuses."""
    rets = _battery("nulls")
    from dust_amd import _lib as L
    import ctypes as C
    assert len(rets) == len(L.SYMBOLS)
    for name, (res, args) in L.SYMBOLS.items():
        if res is C.c_int and args:
            assert rets[name] != "0", f"{name} accepted null arguments"


@pytest.mark.gpu
def test_valid_handles_with_null_arguments_do_not_crash_either():
    """the same with live context / model / scene / pipeline handles in the first argument and null or zero everywhere else: the calls
    that cannot work with that refuse, none faults (dust_hip_sync, commit, clear and the like simply succeed)"""
    rets = _battery("handles")
    for name in ("dust_hip_render_frame", "dust_hip_pipeline_set_noise", "dust_hip_pipeline_read_plane",   # (an edit batch of 0 voxels is fine)
                 "dust_hip_scene_add_instance", "dust_hip_pipeline_configure_gi", "dust_hip_pipeline_gi_exchange", "dust_hip_tone_map",
                 "dust_hip_pipeline_set_frames_in_flight", "dust_hip_pipeline_pass_stats"):
        assert rets[name] != "0", f"{name} accepted null arguments"


def test_integration_doc_declares_every_entry_point():
    """INTEGRATION.md shows the reference-side binding (the Rust extern block a maintainer would add): it names exactly the functions
    include/dust_hip.h declares."""
    header = set(re.findall(r"\b(dust_[a-z0-9_]+)\s*\(", open(os.path.join(ROOT, "include", "dust_hip.h")).read()))
    doc = set(re.findall(r"pub fn (dust_[a-z0-9_]+)", open(os.path.join(ROOT, "INTEGRATION.md")).read()))
    assert header == set(L.SYMBOLS)
    assert doc == header, (sorted(header - doc), sorted(doc - header))
