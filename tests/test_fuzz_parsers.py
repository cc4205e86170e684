"""Mutated .vox and .png files must come back as DUST_ERR_PARSE / UNSUPPORTED (or parse), never crash the process.
(The same corpus generator was run 4000x under ASan + UBSan against vox.cpp / png.cpp / vdb.cpp: clean.)"""
import random

import numpy as np

from dust_amd import _lib as L, api, synth


def _mutate(rnd, b):
    b = bytearray(b)
    for _ in range(rnd.choice([1, 1, 2, 4, 16])):
        op, i = rnd.random(), rnd.randrange(len(b))
        if op < 0.5:
            b[i] = rnd.randrange(256)
        elif op < 0.7:
            b[i:i + 4] = rnd.choice([b"\xff\xff\xff\xff", b"\x00\x00\x00\x00", b"\xff\xff\xff\x7f", b"\x00\x00\x00\x80"])
        elif op < 0.85:
            del b[i:i + rnd.randrange(1, 64)]
        else:
            b[i:i] = bytes(rnd.randrange(256) for _ in range(rnd.randrange(1, 32)))
    if rnd.random() < 0.1:
        b = b[:rnd.randrange(len(b))]
    return bytes(b)


def test_mutated_files_are_rejected_cleanly():
    rnd = random.Random(7)
    rng = np.random.default_rng(7)
    vox = synth.teapot_scene(24)
    castle, _ = synth.castle_scene(scale=0.08)
    png = synth.write_apng(rng.integers(0, 256, (3, 16, 16, 3), dtype=np.uint8))
    parsed = rejected = 0
    for _ in range(700):
        src, kind = rnd.choice([(vox, "vox"), (castle, "vox"), (png, "png")])
        data = _mutate(rnd, src)
        try:
            if kind == "vox":
                s = api.VoxScene(data)
                for m in sorted({m for m, _ in s.instances})[:2]:
                    s.model_data(m)
            else:
                api.load_png_array(data)
            parsed += 1
        except L.DustError as e:
            assert e.status in (L.ERR_PARSE, L.ERR_UNSUPPORTED, L.ERR_INVALID_ARGUMENT, L.ERR_OUT_OF_MEMORY), e
            rejected += 1
    assert rejected > 100 and parsed > 10
