"""GPU box: the converter's filter block (converter.py:196-252: bbox -> alpha -> density -> SOR -> add_rgb_from_sh -> .data) on a table
beyond the suite's sizes (default 50M rows = 12.4 GB), lazy class (device chain, one fused compaction) against the eager class
(a host table after every call) -- two code paths, same bytes -- and the colours against the reference's numpy expression.
usage: python tests/devtools/check_chain_large.py [n]"""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tools.probe_sog import table   # noqa: E402
gsx = importlib.import_module("3dgsconverter_amd")


def main(n=50_000_000):
    t = table(n, 11)
    rng = np.random.default_rng(12)
    edge = float(np.cbrt(n / 80000.0))          # ~80 000 points per unit volume, like the 10M / L = 5 headline cloud
    for a in "xyz":
        t[a] = rng.random(n, dtype=np.float32) * np.float32(edge)
    print("table: %d rows, %.1f GB, box edge %.2f" % (n, t.nbytes / 1e9, edge), flush=True)

    def block(p):
        p.crop_by_bbox(0.05, 0.05, 0.05, edge - 0.05, edge - 0.05, edge - 0.05)
        p.apply_alpha_filter(10)
        p.apply_density_filter(voxel_size=1.0, threshold_percentage=0.1)      # 0.1 % of the rows per unit voxel: the thin edge voxels go
        p.remove_flyers(k=16, threshold_factor=1.0)
        p.apply_auto_bbox()
        p.add_rgb_from_sh()
    out = {}
    for lazy in (True, False):
        p = gsx.DataProcessor(t, lazy=lazy)
        t0 = time.perf_counter()
        block(p)
        res = p.data
        print("lazy=%d: %.0f ms -> %d rows of %d bytes" % (lazy, (time.perf_counter() - t0) * 1e3, len(res), res.dtype.itemsize), flush=True)
        out[lazy] = res
        del p
    a, b = out[True], out[False]
    bad = int(a.dtype != b.dtype or len(a) != len(b))
    if not bad:
        step = 2_000_000
        for i in range(0, len(a), step):
            bad += int(a[i:i + step].tobytes() != b[i:i + step].tobytes())
        for c, nm in enumerate(("red", "green", "blue")):
            f = a["f_dc_%d" % c][:5_000_000]
            lin = np.clip(0.5 + f * 0.28209479177387814, 0.0, 1.0)
            bad += int(np.count_nonzero((np.power(lin, 1.0 / 2.2) * 255).astype(np.uint8) != a[nm][:5_000_000]))
    print("check_chain_large: n=%d, %d survivors, %d mismatches (lazy vs eager tables, colours vs numpy)" % (n, len(a), bad), flush=True)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(*[int(v) for v in sys.argv[1:2]]))
