"""GPU box: the compressed-PLY writer's numeric core (device-resident rows: Morton order, chunk bounds, packed words, SH bytes) at a
size beyond the pytest suite's 2M, against the restated reference (oracle/cply.py: formats/compressed_ply.py:126-297) -- the
order, every chunk row, every packed word, every SH byte.  The oracle's chunk loop takes ~1 min per 10M splats.
usage: python tests/devtools/check_cply_large.py [n] [kind: scene|clustered]"""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import cply as ocply                 # noqa: E402
from tools.probe_sog import table                # noqa: E402
writer = importlib.import_module("3dgsconverter_amd.formats.compressed_ply_writer")


def main(n=10_000_000, kind="clustered"):
    data = table(n, 7)
    if kind == "clustered":      # a clump that needs the second and third Morton level, coincident points, a flat sheet
        rng = np.random.default_rng(8)
        m = n // 5
        for ax in "xyz":
            data[ax][:m] = np.float32(1.25) + rng.standard_normal(m).astype(np.float32) * np.float32(2e-4)
            data[ax][:m // 2] = np.float32(1.25) + rng.standard_normal(m // 2).astype(np.float32) * np.float32(1.2e-7)
        data["x"][m:m + 4000], data["y"][m:m + 4000], data["z"][m:m + 4000] = np.float32(-2.0), np.float32(0.5), np.float32(3.0)
        data["z"][m + 4000:m + 9000] = np.float32(0.75)
        data = data[rng.permutation(n)]
    times = []
    for _ in range(2):
        t = time.perf_counter()
        chunk, vertex, sh, order = writer.encode(data)
        times.append((time.perf_counter() - t) * 1e3)
    print("encode of %d splats: %s ms" % (n, [round(x, 1) for x in times]), flush=True)
    t0 = time.time()
    want_order, depth = ocply.morton_order(data["x"], data["y"], data["z"])
    bad = int(np.count_nonzero(order != want_order))
    print("order: %d differences, %d level(s), %.0f s" % (bad, depth, time.time() - t0), flush=True)
    sh_names = writer.active_sh_names(data)
    t0 = time.time()
    wc, wv, wsh = ocply.encode(data, want_order, sh_names)
    got_c = chunk.view(np.float32).reshape(len(chunk), 18)
    got_v = vertex.view(np.uint32).reshape(n, 4)
    bad_c = int(np.count_nonzero(got_c.view(np.uint32) != wc.view(np.uint32)))
    bad_v = int(np.count_nonzero(got_v != wv))
    bad_s = int(np.count_nonzero(sh.view(np.uint8).reshape(n, -1) != wsh)) if sh_names else 0
    print("check_cply_large: n=%d %s: chunk words %d, vertex words %d, SH bytes %d differ; oracle %.0f s" % (n, kind, bad_c, bad_v, bad_s, time.time() - t0), flush=True)
    return 1 if bad + bad_c + bad_v + bad_s else 0


if __name__ == "__main__":
    a = sys.argv[1:]
    sys.exit(main(int(a[0]) if a else 10_000_000, a[1] if len(a) > 1 else "clustered"))
