import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
from tools.probe_sog import table
dp = importlib.import_module("3dgsconverter_amd.processing.data_processor")
n = 10_000_000
base = table(n, 7)
def run(label, lazy, fn):
    ts = []
    for _ in range(3):
        d = base.copy()
        p = dp.DataProcessor(d, lazy=lazy)
        t = time.perf_counter(); fn(p); out = p.data; ts.append((time.perf_counter() - t) * 1e3)
        rows = len(out)
    print("%-34s lazy=%d  %s ms  -> %d rows" % (label, lazy, [round(x, 1) for x in ts], rows), flush=True)
for lazy in (False, True):
    run("apply_density_filter(s=0.5)", lazy, lambda p: p.apply_density_filter(sensitivity=0.5))
    run("remove_flyers(intensity=5)", lazy, lambda p: p.remove_flyers(intensity=5))
    run("apply_alpha_filter(20)", lazy, lambda p: p.apply_alpha_filter(20))
    run("crop_by_bbox(-3..3)", lazy, lambda p: p.crop_by_bbox(-3, -3, -3, 3, 3, 3))
    run("apply_auto_bbox", lazy, lambda p: p.apply_auto_bbox())
    run("cap_sh_degree(1)", lazy, lambda p: p.cap_sh_degree(1))
    run("add_rgb_from_sh", lazy, lambda p: p.add_rgb_from_sh())
    run("bbox+alpha+density+sor chain", lazy, lambda p: (p.crop_by_bbox(-6, -6, -6, 6, 6, 6), p.apply_alpha_filter(10), p.apply_density_filter(sensitivity=0.5), p.remove_flyers(intensity=5)))
