"""GPU box: every public method of the drop-in DataProcessor on a 10M-row, 248-byte-row table, eager and lazy (what install() binds),
host table in -> host table out; the previous result is freed OUTSIDE the clock (the munmap of a 2.5 GB array takes ~100 ms).
    python tools/probe_dataprocessor.py        # PROBE_N=10000000"""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.probe_sog import table   # noqa: E402
dp = importlib.import_module("3dgsconverter_amd.processing.data_processor")
n = int(os.environ.get("PROBE_N", 10_000_000))
base = table(n, 7)
rng = np.random.default_rng(3)
for a in "xyz":                       # a cloud the density filter keeps: uniform in a 5-unit box (bench.py's headline cloud)
    base[a] = rng.random(n, dtype=np.float32) * np.float32(5.0)


def run(label, lazy, fn):
    ts, rows = [], 0
    for _ in range(3):
        d = base.copy()
        p = dp.DataProcessor(d, lazy=lazy)
        t = time.perf_counter()
        fn(p)
        out = p.data
        ts.append((time.perf_counter() - t) * 1e3)
        rows = len(out)
        del out, p, d
    print("%-46s lazy=%d  %s ms  -> %d rows" % (label, lazy, [round(x, 1) for x in ts], rows), flush=True)


for lazy in (False, True):
    run("apply_density_filter(s=0.5)", lazy, lambda p: p.apply_density_filter(sensitivity=0.5))
    run("remove_flyers(k=16, sigma=1)", lazy, lambda p: p.remove_flyers(k=16, threshold_factor=1.0))
    run("apply_alpha_filter(20)", lazy, lambda p: p.apply_alpha_filter(20))
    run("crop_by_bbox(0.5..4.5)", lazy, lambda p: p.crop_by_bbox(0.5, 0.5, 0.5, 4.5, 4.5, 4.5))
    run("apply_auto_bbox", lazy, lambda p: p.apply_auto_bbox())
    run("cap_sh_degree(1)", lazy, lambda p: p.cap_sh_degree(1))
    run("add_rgb_from_sh", lazy, lambda p: p.add_rgb_from_sh())
    run("bbox+alpha+density+sor (converter.py:196-236)", lazy, lambda p: (p.crop_by_bbox(0.1, 0.1, 0.1, 4.9, 4.9, 4.9), p.apply_alpha_filter(10),
                                                                          p.apply_density_filter(sensitivity=0.5), p.remove_flyers(k=16, threshold_factor=1.0)))
    run("... + add_rgb_from_sh (:243-252)", lazy, lambda p: (p.crop_by_bbox(0.1, 0.1, 0.1, 4.9, 4.9, 4.9), p.apply_alpha_filter(10),
                                                             p.apply_density_filter(sensitivity=0.5), p.remove_flyers(k=16, threshold_factor=1.0),
                                                             p.add_rgb_from_sh()))
