"""GPU box: the SOG writer's device-resident core at a size beyond the bench's (default 50M splats = 12.4 GB of rows), every
position / quaternion / alpha / codebook-index texel against the restated reference statements (oracle/sog.py,
formats/sog.py:264-459) on the lexsorted table.  Minutes of numpy on the host; not part of the pytest suite.
usage: python tests/devtools/check_sog_large.py [n] [level]"""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import sog as osog, kmeans as okm   # noqa: E402
from tools.probe_sog import table               # noqa: E402
w = importlib.import_module("3dgsconverter_amd.formats.sog_writer")


def main(n=50_000_000, level=2):
    t0 = time.time()
    data = table(n, 5)
    print("table: %d rows, %.1f GB, %.0f s" % (n, data.nbytes / 1e9, time.time() - t0), flush=True)
    np.random.seed(1)
    times = []
    for _ in range(2):
        t = time.perf_counter()
        core = w.encode(data, level, device_resident=True)
        times.append((time.perf_counter() - t) * 1e3)
    print("encode: %s ms, stats %s" % ([round(x, 1) for x in times], core["stats"]), flush=True)
    tex = core["textures"]
    t0 = time.time()
    order = osog.order(data)
    bad = 0
    with np.errstate(all="ignore"):
        for lo_i in range(0, n, 5_000_000):      # the sorted table in pieces: no second 12 GB copy
            sl = slice(lo_i, min(n, lo_i + 5_000_000))
            ds = data[order[sl]]
            q, a = osog.quats(ds), osog.opacity_u8(ds)
            bad += int(np.count_nonzero(tex["quats"][sl] != q)) + int(np.count_nonzero(tex["sh0"][sl, 3] != a))
            for name, cols, cbk in (("scales", ["scale_0", "scale_1", "scale_2"], "scale_codebook"), ("sh0", ["f_dc_0", "f_dc_1", "f_dc_2"], "color_codebook")):
                cb = np.asarray(core[cbk], dtype=np.float32)
                for ch, col in enumerate(cols):
                    bad += int(np.count_nonzero(okm.quantize_to_codebook(ds[col], cb) != tex[name][sl, ch]))
            # positions: the reference's expression needs the global min / max of the transformed axes -- the core's own are checked
            # against numpy's below, then used here
            for c, ax in enumerate("xyz"):
                l = np.sign(ds[ax]) * np.log(np.abs(ds[ax]) + 1.0)
                u = np.clip((l - core["mins"][c]) / (core["maxs"][c] - core["mins"][c]) * 65535, 0, 65535).astype(np.uint16)
                bad += int(np.count_nonzero((u & 0xff).astype(np.uint8) != tex["means_l"][sl, c]))
                bad += int(np.count_nonzero((u >> 8).astype(np.uint8) != tex["means_u"][sl, c]))
        for c, ax in enumerate("xyz"):
            l = np.sign(data[ax]) * np.log(np.abs(data[ax]) + 1.0)
            bad += int(np.float32(np.min(l)).tobytes() != np.float32(core["mins"][c]).tobytes())
            bad += int(np.float32(np.max(l)).tobytes() != np.float32(core["maxs"][c]).tobytes())
    lab = tex["shN_labels"][:n, 0].astype(np.int64) + 256 * tex["shN_labels"][:n, 1].astype(np.int64)
    plan = okm.sog_sh_plan(n, level)
    bad += int(np.count_nonzero(lab // plan["k_per_chunk"] != np.arange(n) // plan["chunk_size"]))
    for name, fill in (("means_l", 255), ("means_u", 255), ("quats", 255), ("scales", 0), ("sh0", 0), ("shN_labels", 0)):
        bad += int(np.count_nonzero(tex[name][n:] != fill))
    print("check_sog_large: n=%d level=%d, %d mismatching bytes, numpy check %.0f s" % (n, level, bad, time.time() - t0), flush=True)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(*[int(v) for v in sys.argv[1:3]]))
