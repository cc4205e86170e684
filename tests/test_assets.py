"""dust_amd/assets.py: the reference's LFS assets replace their stand-ins only when the file's sha256 is the pointer's oid."""
import hashlib
import os

import numpy as np

from dust_amd import assets, synth


def test_oids_are_the_reference_pointers():
    """the table is data copied from /root/reference/assets/* (LFS pointer files); where the checkout is present, compare"""
    ref = "/root/reference/assets"
    for name, (oid, size) in assets.LFS_OIDS.items():
        assert len(oid) == 64 and size > 0
        path = os.path.join(ref, name)
        if os.path.exists(path) and os.path.getsize(path) < 1024:
            text = open(path).read()
            assert f"oid sha256:{oid}" in text and f"size {size}" in text, name


def test_stand_ins_unless_the_hash_matches(tmp_path):
    a = assets.Assets(None)
    vox, info, real = a.castle(scale=0.1)
    assert not real and info["n_instances"] > 10 and vox[:4] == b"VOX "
    assert "stand-in" in a.sources["castle.vox"]
    # a directory with files of the right NAMES but other contents (an LFS pointer, a truncated download): refused, with the reason
    (tmp_path / "castle.vox").write_text("version https://git-lfs.github.com/spec/v1\noid sha256:00\nsize 1\n")
    (tmp_path / "teapot.vox").write_bytes(b"x" * assets.LFS_OIDS["teapot.vox"][1])
    b = assets.Assets(str(tmp_path))
    _, _, real = b.castle()
    tea, tea_real = b.teapot()
    assert not real and not tea_real and tea[:4] == b"VOX "
    assert "LFS pointer" in b.sources["castle.vox"] and "sha256" in b.sources["teapot.vox"]
    assert b.summary()["used"] == []


def test_a_file_with_the_listed_hash_is_used(tmp_path):
    """with a table that lists the stand-ins' own hashes, the files ARE taken (and parsed by the product loaders)"""
    vox = synth.teapot_scene(32)
    n0 = synth.stbn_scalar(layers=2)
    png = synth.write_apng([n0[i] for i in range(2)])
    os.makedirs(tmp_path / "stbn")
    (tmp_path / "teapot.vox").write_bytes(vox)
    (tmp_path / "stbn" / "scalar_2Dx1Dx1D_128x128x64x1.png").write_bytes(png)
    table = dict(assets.LFS_OIDS)
    table["teapot.vox"] = (hashlib.sha256(vox).hexdigest(), len(vox))
    table["stbn/scalar_2Dx1Dx1D_128x128x64x1.png"] = (hashlib.sha256(png).hexdigest(), len(png))
    a = assets.Assets(str(tmp_path), table)
    data, real = a.teapot()
    assert real and data == vox
    tex0, tex5 = a.noise()
    assert tex0.shape == (2, 128, 128) and np.array_equal(tex0, n0)          # the APNG, through dust_png_load_array
    assert tex5.shape == (64, 128, 128, 4)                                    # still the stand-in
    s = a.summary()
    assert sorted(s["used"]) == ["stbn/scalar_2Dx1Dx1D_128x128x64x1.png", "teapot.vox"]
