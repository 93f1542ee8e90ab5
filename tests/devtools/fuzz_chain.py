"""GPU box: random sequences of the drop-in DataProcessor's methods (crop_by_bbox, apply_alpha_filter, apply_density_filter, remove_flyers,
cap_sh_degree, add_rgb_from_sh, apply_auto_bbox -- any order, repeats allowed) on random tables, LAZY class (device chain, deferred
column fills and colours, one fused compaction) against the EAGER class (a host table after every call, the reference's order of
operations): the final tables must be the same bytes.  The eager class's single steps are pinned to the reference elsewhere
(tests/test_density_gpu.py, test_sor_gpu.py, test_host_rows.py, the e2e drop-in test).
usage: python tests/devtools/fuzz_chain.py [cases] [seed]"""
import importlib, io, os, sys, time, contextlib
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import datasets   # noqa: E402
gsx = importlib.import_module("3dgsconverter_amd")


def make(rng):
    n = int(np.exp(rng.uniform(np.log(3000), np.log(250000))))
    t = datasets.sog_scene(n, int(rng.integers(1 << 30)))
    edge = float(np.cbrt(n / rng.uniform(200, 3000)))
    for a in "xyz":
        t[a] = (rng.random(n) * edge).astype(np.float32)
    if rng.random() < 0.3:          # a few far flyers
        m = max(1, n // 200)
        for a in "xyz":
            t[a][rng.integers(0, n, m)] = (rng.random(m) * edge * 8 - edge * 3).astype(np.float32)
    steps = []
    for _ in range(int(rng.integers(1, 7))):
        kind = str(rng.choice(["bbox", "alpha", "density", "sor", "cap", "rgb", "auto"]))
        if kind == "bbox":
            lo, hi = sorted(rng.uniform(-0.1 * edge, 1.1 * edge, 2))
            # bounds as the caller may hand them: Python floats (weak scalars: the comparison stays float32), numpy scalars of either
            # width (float64 promotes the comparison), ints
            cast = [float, np.float32, np.float64, lambda v: int(round(v))][int(rng.integers(0, 4))]
            steps.append(("crop_by_bbox", tuple(cast(v) for v in (lo, lo, lo, hi, hi, hi))))
        elif kind == "alpha":
            steps.append(("apply_alpha_filter", (int(rng.integers(0, 256)),)))
        elif kind == "density":
            steps.append(("apply_density_filter", (float(rng.uniform(0.5, 2.0) * edge / 6), float(rng.uniform(0.01, 1.5)), None, bool(rng.random() < 0.5))))
        elif kind == "sor":
            steps.append(("remove_flyers", (int(rng.integers(2, 40)), float(rng.uniform(0.3, 3.0)))))
        elif kind == "cap":
            steps.append(("cap_sh_degree", (int(rng.integers(0, 4)),)))
        elif kind == "rgb":
            steps.append(("add_rgb_from_sh", ()))
        else:
            steps.append(("apply_auto_bbox", ()))
    return t, steps


def main(cases=100, seed=0):
    rng = np.random.default_rng(seed)
    bad, t0 = 0, time.time()
    for c in range(cases):
        t, steps = make(rng)
        out = {}
        for lazy in (True, False):
            p = gsx.DataProcessor(t.copy(), lazy=lazy)
            with contextlib.redirect_stdout(io.StringIO()):
                for name, args in steps:
                    getattr(p, name)(*args)
                out[lazy] = p.data
        a, b = out[True], out[False]
        same = a.dtype == b.dtype and len(a) == len(b) and a.tobytes() == b.tobytes()
        bad += not same
        print("%3d n=%7d -> %7d rows  %s  %s" % (c, len(t), len(b), " ".join({"crop_by_bbox": "bbox", "apply_alpha_filter": "alpha", "apply_density_filter": "density", "remove_flyers": "sor",
                                                         "cap_sh_degree": "cap", "add_rgb_from_sh": "rgb", "apply_auto_bbox": "auto"}[s[0]] for s in steps),
                                               "ok" if same else "MISMATCH (lazy %d rows %s)" % (len(a), a.dtype.itemsize)), flush=True)
    print("fuzz_chain: %d cases, %d mismatches, %.1f s" % (cases, bad, time.time() - t0))
    return bad


if __name__ == "__main__":
    sys.exit(1 if main(*(int(a) for a in sys.argv[1:3])) else 0)
