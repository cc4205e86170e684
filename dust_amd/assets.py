"""Real-asset drop-in: the reference keeps castle.vox, teapot.vox and the spatiotemporal blue-noise textures in Git
LFS, so a checkout holds only pointer files (`oid sha256:...`, `size ...`). Everything here runs on the synthetic stand-ins of
dust_amd.synth unless a directory is given that holds a file whose sha256 IS the oid of the reference's pointer -- then that file
is parsed by the product loaders (dust_vox_load, dust_png_load_array) and used instead, and the caller is told which was which.

The table below is data copied from the pointer files under /root/reference/assets (the oids are also in SURVEY.md 8d)."""
import hashlib
import os

import numpy as np

# name -> (sha256 oid, size in bytes) of /root/reference/assets/<name>
LFS_OIDS = {
    "castle.vox": ("cbc09a7c84fc44d5f669c0c0801715fb281c2c2c09ae78853c9ee1ac7a4161a2", 88233039),
    "teapot.vox": ("68cdab795e00401866ae46584d31608da2a3137065bfdfc21ab8fdd9d67b52ff", 143083),
    "stbn/scalar_2Dx1Dx1D_128x128x64x1.png": ("448a1d7afedbc4ae584ea07c6b4094af5903140c44736bc13c3688528173d47c", 1060989),
    "stbn/unitvec3_cosine_2Dx1D_128x128x64.png": ("11f209d7962aa9f690cf00fe9a2bdf6d6728415ff646a28b33a9754fc73c7be2", 3124383),
}


def find(assets_dir, name, table=None):
    """-> (bytes, None) if <assets_dir>/<name> (or its basename directly in the directory) has the listed sha256, else
    (None, why not). A Git LFS pointer file, a truncated download or any other file of that name is refused."""
    table = LFS_OIDS if table is None else table
    if not assets_dir:
        return None, "no --assets directory"
    oid, size = table[name]
    for cand in (os.path.join(assets_dir, name), os.path.join(assets_dir, os.path.basename(name))):
        if not os.path.isfile(cand):
            continue
        if os.path.getsize(cand) != size:
            return None, f"{cand}: {os.path.getsize(cand)} bytes, the reference's is {size}" + (
                " (a Git LFS pointer file)" if os.path.getsize(cand) < 1024 else "")
        with open(cand, "rb") as f:
            data = f.read()
        got = hashlib.sha256(data).hexdigest()
        if got != oid:
            return None, f"{cand}: sha256 {got[:12]}... is not the reference's {oid[:12]}..."
        return data, None
    return None, f"{name} not in {assets_dir}"


class Assets:
    """What a run uses for the scene file and the two noise textures the shaders sample, and where each came from."""

    def __init__(self, assets_dir=None, table=None):
        self.dir, self.table = assets_dir, table
        self.sources = {}   # name -> "reference asset (sha256 ok)" | "stand-in (<why>)"

    def _get(self, name):
        data, why = find(self.dir, name, self.table)
        self.sources[name] = "reference asset, sha256 verified" if data is not None else f"stand-in ({why})"
        return data

    def castle(self, scale=1.0):
        """-> (.vox bytes, info or None, is_real)"""
        from . import synth
        data = self._get("castle.vox") if scale == 1.0 else None
        if data is not None:
            return data, None, True
        if scale != 1.0:
            self.sources["castle.vox"] = f"stand-in (scale {scale})"
        vox, info = synth.castle_scene(scale=scale)
        return vox, info, False

    def teapot(self):
        from . import synth
        data = self._get("teapot.vox")
        return (data, True) if data is not None else (synth.teapot_scene(96), False)

    def noise(self):
        """-> (texture 0: (layers,128,128) u8, texture 5: (layers,128,128,4) u8)"""
        from . import api, synth
        out = []
        for name, standin, channels in (("stbn/scalar_2Dx1Dx1D_128x128x64x1.png", synth.stbn_scalar, 1),
                                        ("stbn/unitvec3_cosine_2Dx1D_128x128x64.png", synth.stbn_unitvec3_cosine, 4)):
            data = self._get(name)
            tex = None
            if data is not None:
                arr = api.load_png_array(data)   # (layers, h, w, channels): PngLoader's layout (png.rs:70-200)
                if arr.dtype == np.uint8 and arr.shape[1:3] == (128, 128) and arr.shape[3] == channels:
                    tex = np.ascontiguousarray(arr[..., 0] if channels == 1 else arr)
                else:
                    self.sources[name] = f"stand-in (the file decodes to {arr.shape} {arr.dtype}, not 128x128x{channels} u8)"
            out.append(tex if tex is not None else standin())
        return out[0], out[1]

    def summary(self):
        real = [n for n, s in self.sources.items() if s.startswith("reference")]
        return {"dir": self.dir, "used": real, "sources": dict(self.sources)}
