"""Golden HASHES of bundles the un-patched reference writes (formats/sog.py:249-639, ``SogFormat.write``) for tables the
committed texture fixtures of make_golden_kmeans.py do not cover: a larger scene, trailing SH coefficients that are all zero
(band downgrade, :476-491), a degree-1 table, coordinates on a coarse lattice with ties and signed zeros (the lexsort's stable
order decides which quaternion lands where).  TEST INFRASTRUCTURE ONLY; needs /root/reference (build container).

    python oracle/make_golden_sog.py      ->  tests/golden/sog_ref_hashes.json

Stored per case: sha256 of the init-independent textures (means_l, means_u, quats, the alpha channel of sh0; first n texels and the
padding), `meta.means`, bands, palette count, texture size.  tests/test_sog_gpu.py replays the cases through the device-resident
core."""
from __future__ import annotations

import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import make_golden_kmeans as mk, refload          # noqa: E402
from oracle.datasets import sog_scene                          # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "sog_ref_hashes.json")

CASES = {
    "sog_120k_l5": {"n": 120000, "seed": 61, "level": 5, "np_seed": 71, "variant": "plain"},
    "sog_60k_l0_bands2": {"n": 60000, "seed": 62, "level": 0, "np_seed": 72, "variant": "zero_tail_24"},
    "sog_40k_l9_lattice": {"n": 40000, "seed": 63, "level": 9, "np_seed": 73, "variant": "lattice"},
    "sog_30k_l3_bands1": {"n": 30000, "seed": 64, "level": 3, "np_seed": 74, "variant": "zero_tail_9"},
}


def build(case):
    """the table of a case (shared with the test)"""
    t = sog_scene(case["n"], case["seed"])
    v = case["variant"]
    if v.startswith("zero_tail_"):
        for i in range(int(v.rsplit("_", 1)[1]), 45):
            t["f_rest_%d" % i] = 0.0
    elif v == "lattice":
        rng = np.random.default_rng(case["seed"] + 5)
        for a, s in zip("xyz", (3, 2, 1.5)):
            c = np.round(rng.standard_normal(len(t)) * s).astype(np.float32) * np.float32(0.5)
            c[rng.random(len(t)) < 0.05] = np.float32(-0.0)
            t[a] = c
    return t


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()[:16]


def main():
    refload.load()
    import gsconverter.formats.sog as sogmod
    import io
    import tempfile
    import zipfile
    out = {"_meta": {"generator": "oracle/make_golden_sog.py run against /root/reference (v0.8, sklearn fallback for the codebooks)",
                     "numpy": np.__version__}}
    saved = sogmod.status_print
    sogmod.status_print = lambda *a, **k: None
    try:
        for name, case in CASES.items():
            data = build(case)
            n = len(data)
            with tempfile.TemporaryDirectory() as td:
                path = os.path.join(td, "out.sog")
                np.random.seed(case["np_seed"])
                sogmod.SogFormat().write(data, path, compression_level=case["level"])
                with zipfile.ZipFile(path) as zf:
                    meta = json.loads(zf.read("meta.json"))
                    tex = {nm[:-5]: mk.decode_webp(zf, nm) for nm in zf.namelist() if nm.endswith(".webp")}
            rec = dict(case)
            rec.update({"means": meta["means"], "bands": meta.get("shN", {}).get("bands", 0), "palette": meta.get("shN", {}).get("count", 0),
                        "texels": int(len(tex["means_l"])),
                        "sha": {"means_l": sha(tex["means_l"]), "means_u": sha(tex["means_u"]), "quats": sha(tex["quats"]),
                                "sh0_alpha": sha(tex["sh0"][:, 3]), "scales_alpha": sha(tex["scales"][:, 3]),
                                # (padding texels of the label image are (0, 0, 0, 0): under alpha 0 the lossless WebP encoder is free to rewrite
                                # the colour bytes -- PIL's `exact` is off, as in the reference -- so only their alpha survives the round trip)
                                "labels_pad_alpha": sha(tex["shN_labels"][n:, 3]) if "shN_labels" in tex else None}})
            out[name] = rec
            print(name, rec["bands"], rec["palette"], rec["sha"])
    finally:
        sogmod.status_print = saved
    with open(OUT, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("wrote", OUT)


if __name__ == "__main__":
    main()
