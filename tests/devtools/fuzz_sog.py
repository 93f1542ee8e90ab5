"""Randomised sweep of the SOG writer's device-resident core (formats/sog_device.py) on the GPU box against the restated
reference statements (oracle/sog.py: write_core_without_kmeans, formats/sog.py:264-503) -- random sizes, SH degrees, zeroed
coefficient tails (band downgrade), duplicate / tied / signed-zero coordinates, coordinate ranges over many orders of
magnitude, tables widened by u1 fields, every compression level.  Bar: the five images byte for byte, min / max bits.
usage: python tests/devtools/fuzz_sog.py [cases] [seed]"""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import datasets, sog as osog, kmeans as okm
w = importlib.import_module("3dgsconverter_amd.formats.sog_writer")
sd = importlib.import_module("3dgsconverter_amd.formats.sog_device")


def make(rng):
    n = int(np.exp(rng.uniform(np.log(1100), np.log(400000))))
    deg = int(rng.choice([0, 1, 2, 3], p=[0.1, 0.15, 0.15, 0.6]))
    t = datasets.sog_scene(n, int(rng.integers(1 << 30)), sh_degree=deg)
    kind = str(rng.choice(["scene", "wide", "tiny", "grid", "dups", "onesided", "offset"]))
    for a in "xyz":
        if kind == "wide":
            t[a] = (rng.standard_normal(n) * np.exp(rng.uniform(-10, 10, n))).astype(np.float32)
        elif kind == "tiny":
            t[a] = (rng.standard_normal(n) * 1e-5).astype(np.float32)
        elif kind == "grid":
            c = np.round(rng.standard_normal(n) * rng.uniform(1, 6)).astype(np.float32) * np.float32(0.5)
            c[rng.random(n) < 0.05] = np.float32(-0.0)
            t[a] = c
        elif kind == "onesided":
            t[a] = (np.abs(rng.standard_normal(n)) + rng.uniform(0, 5)).astype(np.float32)
        elif kind == "offset":      # a capture far from the origin: most position texels sit next to a rounding boundary of numpy's log
            t[a] = (rng.standard_normal(n) * rng.uniform(0.2, 3) + rng.choice([-1, 1]) * np.exp(rng.uniform(np.log(5), np.log(3000)))).astype(np.float32)
    if kind == "dups":
        src = rng.integers(0, n, n // 3)
        dst = rng.integers(0, n, n // 3)
        for a in "xyz":
            t[a][dst] = t[a][src]
    ncoef = 3 * ((deg + 1) ** 2 - 1)
    if ncoef and rng.random() < 0.4:      # zero a tail of the coefficients: band downgrade
        for i in range(int(rng.integers(0, ncoef)), ncoef):
            t["f_rest_%d" % i] = np.float32(-0.0) if rng.random() < 0.3 else 0.0
    if rng.random() < 0.4:                # rows off the 4-byte grid: u1 fields behind (the converter's colours) and / or in front of the floats
        front, back = int(rng.integers(0, 4)), int(rng.integers(0, 6))
        if front + back == 0:
            back = 3
        wide = np.zeros(n, dtype=np.dtype([("pre%d" % i, "u1") for i in range(front)] + t.dtype.descr + [("post%d" % i, "u1") for i in range(back)]))
        for nm in wide.dtype.names:
            wide[nm] = t[nm] if nm in t.dtype.names else rng.integers(0, 256, n, dtype=np.uint8)
        t = wide
        kind += "+u1(%d,%d)" % (front, back)
    return kind, t, int(rng.integers(0, 10))


def main(cases=40, seed=0):
    rng = np.random.default_rng(seed)
    bad, t0 = 0, time.time()
    for c in range(cases):
        kind, tab, level = make(rng)
        n = len(tab)
        what = []
        try:
            with np.errstate(all="ignore"):
                core = w.encode(tab, level, device_resident=True)
                ref = osog.write_core_without_kmeans(tab, core["scale_codebook"], core["color_codebook"])
        except sd.NotEligible as e:
            print("%3d %-18s n=%7d level %d declined (%s)" % (c, kind, n, level, e), flush=True)
            continue
        for name in ("means_l", "means_u", "quats", "scales", "sh0"):
            if not np.array_equal(core["textures"][name], ref[name]):
                what.append("%s (%d texels)" % (name, int((core["textures"][name] != ref[name]).any(axis=1).sum())))
        for a in range(3):
            if np.float32(core["mins"][a]).tobytes() != np.float32(ref["mins"][a]).tobytes() or np.float32(core["maxs"][a]).tobytes() != np.float32(ref["maxs"][a]).tobytes():
                what.append("min/max axis %d" % a)
        names = tab.dtype.names
        want_bands = sd.bands_from_mask(sd.sh_coeffs_present(names), sum(1 << i for i in range(45) if "f_rest_%d" % i in names and np.any(tab["f_rest_%d" % i] != 0)))
        if core["bands"] != want_bands:
            what.append("bands %d != %d" % (core["bands"], want_bands))
        if core["bands"]:
            plan = okm.sog_sh_plan(n, level)
            lab = core["textures"]["shN_labels"][:n, 0].astype(np.int64) + 256 * core["textures"]["shN_labels"][:n, 1].astype(np.int64)
            if not np.all(lab // plan["k_per_chunk"] == np.arange(n) // plan["chunk_size"]):
                what.append("labels outside their chunk's slice")
        bad += bool(what)
        print("%3d %-18s n=%7d level %d bands %d uncertain %d+%d %s" % (c, kind, n, level, core["bands"], core["stats"]["uncertain_positions"],
                                                                     core["stats"]["uncertain_alpha"], "MISMATCH " + "; ".join(what) if what else "ok"), flush=True)
    print("fuzz_sog: %d cases, %d mismatches, %.1f s" % (cases, bad, time.time() - t0))
    return bad


if __name__ == "__main__":
    sys.exit(1 if main(*(int(a) for a in sys.argv[1:3])) else 0)
