"""Generates Sunlight::bake() outputs (56 floats each):
  * tests/golden/sky_states.json -- four named suns, the vectors the product's dust_sky_bake is checked against;
  * dust_amd/data/sky_sweep.json -- the default sun plus a dense sweep (turbidity 1..10 x solar elevation 2..90 degrees,
    ground albedo 0.2) for hosts that do not have the model's tables: the baked state depends on the sun's direction only
    through its elevation, so a swept state is exact for every azimuth at its elevation (set state[48:51] = direction).

Runs ONLY in the build container: it reads the Hosek-Wilkie tables the reference embeds
(/root/reference/crates/render/src/pipeline/{dataset,datasetSolar}.bin) and restates
crates/render/src/pipeline/sky.rs:90-268 in float32 numpy. The JSON files are data (inputs and expected outputs);
nothing from the reference travels with the repo. bake_with(tables, ...) is also what tests/test_sky_bake.py runs
against dust_sky_bake on synthetic tables, where no reference file is needed.
"""
import json
import os
import sys

import numpy as np

REF = "/root/reference/crates/render/src/pipeline"
f32 = np.float32


def load():
    return split_tables(np.fromfile(os.path.join(REF, "dataset.bin"), dtype="<f4"), np.fromfile(os.path.join(REF, "datasetSolar.bin"), dtype="<f4"))


def split_tables(dataset_f32, solar_f32):
    raw = np.asarray(dataset_f32, "<f4").reshape(-1, 3)
    cfg = raw[:1080]
    rad = raw[1080:1200]
    cfg_low = cfg[:540].reshape(10, 9, 6, 3)    # sky.rs:40-41
    cfg_high = cfg[540:].reshape(10, 9, 6, 3)   # sky.rs:42-47
    rad_low = rad[:60].reshape(10, 6, 3)        # sky.rs:51-52
    rad_high = rad[60:].reshape(10, 6, 3)       # sky.rs:53-54
    sol = np.asarray(solar_f32, "<f4").reshape(-1, 3)
    return cfg_low, cfg_high, rad_low, rad_high, sol[:1800], sol[1800:1806]


def powi(x, n):
    """f32::powi with a constant exponent as LLVM expands it (and compiler-rt's __powisf2 evaluates it): square and multiply
    from the low bit -- x^4 = (x^2)^2, x^5 = x * (x^2)^2, which rounds differently from x*x*x*x*x."""
    r, a = f32(1.0), f32(x)
    while True:
        if n & 1:
            r = f32(r * a)
        n //= 2
        if n == 0:
            return r
        a = f32(a * a)


def powf(x, y):
    """libm powf on float32 arguments: evaluated in double and rounded once (a correctly rounded powf; glibc's is within
    0.52 ulp of that, so the two agree except on rare near-ties)."""
    return f32(np.power(np.float64(x), np.float64(y)))


def coefficient(m, e):  # sky.rs:135-143
    rev = f32(f32(1.0) - e)
    terms = [powi(rev, 5) * m[0],
             f32(f32(5.0) * powi(rev, 4)) * e * m[1],
             f32(f32(10.0) * powi(rev, 3)) * powi(e, 2) * m[2],
             f32(f32(10.0) * powi(rev, 2)) * powi(e, 3) * m[3],
             f32(f32(5.0) * rev) * powi(e, 4) * m[4],
             powi(e, 5) * m[5]]
    acc = terms[0].astype(f32)
    for t in terms[1:]:
        acc = (acc + t.astype(f32)).astype(f32)
    return acc


def blend(low, high, turbidity, albedo, elev):  # sky.rs:145-227
    it = int(turbidity)
    rem = f32(turbidity - f32(it))
    e = powf(f32(elev / f32(np.pi / 2)), f32(1.0 / 3.0))
    res = ((f32(1.0) - albedo) * f32(f32(1.0) - rem) * coefficient(low[it - 1], e)).astype(f32)
    res = (res + (albedo * f32(f32(1.0) - rem) * coefficient(high[it - 1], e)).astype(f32)).astype(f32)
    if it < 10:
        res = (res + ((f32(1.0) - albedo) * rem * coefficient(low[it], e)).astype(f32)).astype(f32)
        res = (res + (albedo * rem * coefficient(high[it], e)).astype(f32)).astype(f32)
    return res


def sr_internal(sol, turb, elev):  # sky.rs:229-254
    pieces, order = 45, 4
    pos = int(f32(powf(f32(f32(f32(2.0) * elev) / f32(np.pi)), f32(1.0 / 3.0)) * f32(pieces)))
    pos = min(pos, pieces - 1)
    break_x = f32(powi(f32(f32(pos) / f32(pieces)), 3) * f32(np.pi / 2))
    x = f32(elev - break_x)
    x_exp = f32(1.0)
    res = np.zeros(3, f32)
    base = order * pieces * turb + order * pos
    for coef in sol[base:base + order][::-1]:
        res = (res + coef * x_exp).astype(f32)
        x_exp = f32(x_exp * x)
    return res


def bake(turbidity, albedo, direction):
    return bake_with(load(), turbidity, albedo, direction)


def bake_with(tables, turbidity, albedo, direction):  # sky.rs:90-132
    cfg_low, cfg_high, rad_low, rad_high, sol, ld = tables
    turbidity = f32(turbidity)
    albedo = np.asarray(albedo, f32)
    direction = np.asarray(direction, f32)
    elev = f32(np.arcsin(np.float64(direction[1])))  # asinf: evaluated in double, rounded once
    it = int(turbidity)
    configs = np.zeros((3, 9), f32)
    for i in range(9):
        configs[:, i] = blend(cfg_low[:, i], cfg_high[:, i], turbidity, albedo, elev)
    radiances = blend(rad_low, rad_high, turbidity, albedo, elev)
    turb_low = int(turbidity) - 1
    turb_frac = f32(turbidity - f32(turb_low + 1))
    if turb_low == 9:
        turb_low, turb_frac = 8, f32(1.0)
    solar = ((f32(1.0) - turb_frac) * sr_internal(sol, turb_low, elev) + turb_frac * sr_internal(sol, turb_low + 1, elev)).astype(f32)
    out = np.zeros(56, f32)
    for c in range(3):
        o = c * 16
        out[o:o + 9] = configs[c]
        out[o + 9] = radiances[c]
        out[o + 10] = ld[0][c]
        out[o + 11] = ld[1][c]
        out[o + 12:o + 16] = [ld[2][c], ld[3][c], ld[4][c], ld[5][c]]
    out[48:51] = direction
    out[51] = 0.0
    out[52:55] = solar
    out[55] = f32(f32(0.51) * f32(np.pi / 180.0)) / f32(2.0)
    return out


SUNS = {
    "default": (1.0, (0.2, 0.2, 0.2), (0.0, 0.80114365, -0.5984721)),   # Sunlight::default, sky.rs:15-23
    "low_sun": (3.0, (0.3, 0.3, 0.3), (0.70710678, 0.17364818, -0.68548)),
    "hazy_noon": (6.5, (0.1, 0.2, 0.4), (0.0, 0.99, -0.14106736)),
    "turbid10": (10.0, (0.5, 0.5, 0.5), (-0.5, 0.5, 0.70710678)),
}

if __name__ == "__main__":
    if not os.path.isdir(REF):
        sys.exit("reference checkout not present; fixtures can only be regenerated in the build container")
    tables = load()
    out = {}
    for name, (t, a, d) in SUNS.items():
        d = np.asarray(d, np.float64)
        d = (d / np.linalg.norm(d)).astype(f32) if name != "default" else np.asarray(d, f32)
        out[name] = {"turbidity": t, "albedo": list(a), "direction": [float(v) for v in d],
                     "state": [float(v) for v in bake_with(tables, t, a, d)]}
    here = os.path.dirname(os.path.abspath(__file__))
    path = os.path.join(here, "sky_states.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", path)
    s = out["default"]["state"]
    print("default radiance XYZ", s[9], s[25], s[41], "solar", s[52:55], "cfg0[0..2]", s[0:3])
    # the packaged sweep: sun in the y-z plane (azimuth as the default sun's), elevation 2..90 degrees, turbidity 1..10
    sweep = {"default": out["default"], "albedo": [0.2, 0.2, 0.2], "turbidity": list(range(1, 11)),
             "elevation_deg": list(range(2, 91, 2)), "states": []}
    for t in sweep["turbidity"]:
        row = []
        for e in sweep["elevation_deg"]:
            er = np.deg2rad(np.float64(e))
            d = np.array([0.0, min(1.0, np.sin(er)), -np.cos(er)], f32)
            row.append([float(v) for v in bake_with(tables, float(t), sweep["albedo"], d)])
        sweep["states"].append(row)
    data_dir = os.path.join(os.path.dirname(os.path.dirname(here)), "dust_amd", "data")
    os.makedirs(data_dir, exist_ok=True)
    with open(os.path.join(data_dir, "sky_sweep.json"), "w") as f:
        json.dump(sweep, f, separators=(",", ":"))
    print("wrote", os.path.join(data_dir, "sky_sweep.json"), len(sweep["turbidity"]) * len(sweep["elevation_deg"]), "states")
