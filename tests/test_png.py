"""PNG / APNG ingest (the reference's PngLoader, rhyolite_bevy/src/loaders/png.rs:70-200), which turns the
stbn/*.png animations into sliced image arrays. The repository's real textures are Git-LFS pointers, so the files here
are written by synth.write_apng (zlib + all five scanline filters) and decoded by the library's own parser."""
import numpy as np
import pytest

from dust_amd import _lib as L, api, synth


def test_grey_apng_like_stbn_scalar():
    rng = np.random.default_rng(1)
    frames = rng.integers(0, 256, (6, 32, 48), dtype=np.uint8)
    got = api.load_png_array(synth.write_apng(frames))
    assert got.shape == (6, 32, 48, 1) and got.dtype == np.uint8
    assert np.array_equal(got[..., 0], frames)


def test_rgb_is_widened_to_rgba_with_zero_alpha():
    rng = np.random.default_rng(2)
    frames = rng.integers(0, 256, (4, 16, 16, 3), dtype=np.uint8)
    got = api.load_png_array(synth.write_apng(frames))
    assert got.shape == (4, 16, 16, 4)
    assert np.array_equal(got[..., :3], frames) and not got[..., 3].any()   # png.rs:150-162


@pytest.mark.parametrize("channels", [2, 4])
def test_alpha_formats_and_single_images(channels):
    rng = np.random.default_rng(3)
    frames = rng.integers(0, 256, (1, 9, 7, channels), dtype=np.uint8)
    for filt in ((0,), (1,), (2,), (3,), (4,), (4, 3, 2, 1, 0)):
        got = api.load_png_array(synth.write_apng(frames, filters=filt))
        assert got.shape == (1, 9, 7, channels) and np.array_equal(got, frames)


def test_sixteen_bit_samples_stay_big_endian():
    rng = np.random.default_rng(4)
    frames = rng.integers(0, 65536, (2, 8, 8), dtype=np.uint16)
    got = api.load_png_array(synth.write_apng(frames))
    assert got.dtype == np.dtype(">u2") and np.array_equal(got[..., 0].astype(np.uint16), frames)


def test_what_the_reference_rejects():
    frames = np.zeros((3, 8, 8), np.uint8)
    with pytest.raises(L.DustError) as e:
        api.load_png_array(synth.write_apng(frames, interlace=1))
    assert e.value.status == L.ERR_UNSUPPORTED
    with pytest.raises(L.DustError) as e:
        api.load_png_array(synth.write_apng(frames, frame_rect=(4, 4, 2, 2)))
    assert e.value.status == L.ERR_UNSUPPORTED
    good = synth.write_apng(frames)
    for broken in (good[:40], b"JUNK" + good[4:], good[:60] + bytes(20) + good[80:]):
        with pytest.raises(L.DustError) as e:
            api.load_png_array(broken)
        assert e.value.status == L.ERR_PARSE


def test_loaded_noise_has_the_layout_set_noise_takes():
    scalar = synth.stbn_scalar(layers=3)                  # (3, 128, 128) R8
    cosine = synth.stbn_unitvec3_cosine(layers=3)         # (3, 128, 128, 4) RGBA8
    a = api.load_png_array(synth.write_apng(scalar))
    b = api.load_png_array(synth.write_apng(cosine[..., :3]))
    assert np.array_equal(a[..., 0], scalar)
    assert np.array_equal(b[..., :3], cosine[..., :3]) and b.shape == (3, 128, 128, 4)
