"""include/dust_hip.hpp (the C++ mirror of the reference's Rust surface) compiled and driven as a host would."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "cpp", "_build", "host_mirror_test")


@pytest.fixture(scope="module")
def exe():
    os.makedirs(os.path.dirname(EXE), exist_ok=True)
    src = os.path.join(ROOT, "tests", "cpp", "host_mirror_test.cpp")
    if not os.path.exists(EXE) or os.path.getmtime(EXE) < max(os.path.getmtime(src), os.path.getmtime(os.path.join(ROOT, "include", "dust_hip.hpp"))):
        subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-I", os.path.join(ROOT, "include"), src,
                               "-L", os.path.join(ROOT, "dust_amd"), "-ldust_hip", "-Wl,-rpath," + os.path.join(ROOT, "dust_amd"),
                               "-Wl,-rpath,/opt/rocm/lib", "-o", EXE])
    return EXE


def test_cpp_mirror_cpu(exe):
    out = subprocess.run([exe, "cpu"], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    assert "cpu ok" in out.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["gpu", "bands", "frames"])
def test_cpp_mirror_gpu_frame_matches_python_path(exe, tmp_path, mode):
    """mode "bands": the frame as 8 emulated ranks render and gather it through dust::DeviceComm (a loopback group on this device,
    include/dust_hip.hpp) -- bit-identical to the single-device frame. mode "frames": three frames in one call and one launch
    (dust::StandardPipeline::render_frames -> dust_hip_render_frames); the first one's planes are the single frame's."""
    import parity_util as P
    from dust_amd import _lib as L, api, synth
    data, _ = synth.castle_scene(scale=0.15)
    (tmp_path / "castle.vox").write_bytes(data)
    noise5 = synth.stbn_unitvec3_cosine(layers=2)
    (tmp_path / "noise5.bin").write_bytes(noise5.tobytes())
    sky = P.sky_state()
    (tmp_path / "sky.bin").write_bytes(sky.astype(np.float32).tobytes())
    w, h = 160, 90
    out = subprocess.run([exe, mode, str(tmp_path / "castle.vox"), str(w), str(h), str(tmp_path / "noise5.bin"),
                          str(tmp_path / "sky.bin"), str(tmp_path / "out")], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    ctx = api.Context(device=0)
    desc = P.SceneDesc.from_vox(data)
    scene = P.hip_scene(ctx, desc)
    pipe = api.StandardPipeline(ctx, w, h)
    pipe.set_noise(5, noise5)
    s = 0.15
    eye = (122.0 * s, 300.61 * s, 54.45 * s)
    pipe.render(scene, P.camera_for(eye), sky, L.PASS_PRIMARY | L.PASS_AMBIENT_OCCLUSION, frame_index=1, rand=4242)
    depth = np.fromfile(tmp_path / "out.depth", np.float32).reshape(h, w)
    vid = np.fromfile(tmp_path / "out.vid", np.uint32).reshape(h, w)
    ill = np.fromfile(tmp_path / "out.ill", np.uint16).reshape(h, w, 4)
    assert np.array_equal(depth.view(np.uint32), pipe.read_plane(L.PLANE_DEPTH).view(np.uint32))
    assert np.array_equal(vid, pipe.read_plane(L.PLANE_VOXEL_ID))
    assert np.array_equal(ill, pipe.read_plane(L.PLANE_ILLUMINANCE))


@pytest.mark.gpu
def test_moving_instance_commits_do_not_stall_the_host(exe, tmp_path):
    """castle.rs:287-291 moves the teapot every frame and tlas.rs:37-65 rebuilds the TLAS in the frame's command stream. Here:
    dust_hip_scene_set_transform + dust_hip_scene_commit from C++, 300 frames of the full-size castle with a frame in flight --
    the pair must cost microseconds of host time (it used to be a flush, a stream wait and five hipFree/hipMalloc/hipMemcpy)."""
    import parity_util as P
    from dust_amd import synth
    data, _ = synth.castle_scene()
    (tmp_path / "castle.vox").write_bytes(data)
    (tmp_path / "noise5.bin").write_bytes(synth.stbn_unitvec3_cosine(layers=2).tobytes())
    (tmp_path / "sky.bin").write_bytes(P.sky_state().astype(np.float32).tobytes())
    out = subprocess.run([exe, "commit", str(tmp_path / "castle.vox"), "1920", "1080", str(tmp_path / "noise5.bin"), str(tmp_path / "sky.bin"), "300"],
                         capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    print(out.stdout.strip())
    us = float(out.stdout.split("set_transform + commit")[1].split("us")[0])
    assert us < 60.0, out.stdout   # measured ~10 us; the old path was ~300 us + a stream wait


def test_shared_reciprocal_division_is_ieee_division():
    """kernels.hip div_by (Markstein's sequence on an exactly rounded reciprocal) == IEEE a / b, bit for bit."""
    src = os.path.join(ROOT, "tests", "cpp", "division_identity_test.c")
    exe = os.path.join(ROOT, "tests", "cpp", "_build", "division_identity_test")
    os.makedirs(os.path.dirname(exe), exist_ok=True)
    if not os.path.exists(exe) or os.path.getmtime(exe) < os.path.getmtime(src):
        subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", src, "-o", exe, "-lm"])
    out = subprocess.run([exe, "20000000"], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout
    assert "bad 0" in out.stdout
