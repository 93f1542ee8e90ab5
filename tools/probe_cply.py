"""GPU box: the compressed-PLY writer's numeric core on device-resident rows -- per-stage clock (PROBE_N splats, degree-3 table).
    python tools/probe_cply.py            # PROBE_N=10000000 PROBE_REPS=4 PROBE_KIND=scene|clustered"""
import importlib, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools.probe_sog import table   # noqa: E402


def main():
    n = int(os.environ.get("PROBE_N", 10_000_000))
    reps = int(os.environ.get("PROBE_REPS", 4))
    L = importlib.import_module("3dgsconverter_amd._lib")
    w = importlib.import_module("3dgsconverter_amd.formats.compressed_ply_writer")
    data = table(n, 7)
    if os.environ.get("PROBE_KIND", "scene") == "clustered":
        rng = np.random.default_rng(8)
        m = n // 5
        for ax in "xyz":
            data[ax][:m] = np.float32(1.25) + rng.standard_normal(m).astype(np.float32) * np.float32(2e-4)
        data = data[rng.permutation(n)]
    names = w.active_sh_names(data)
    runs, stages = [], None
    for r in range(reps):
        st = {}
        t = time.perf_counter()
        out = L.cply_pack_table(data, names, None, None, st)
        runs.append(round((time.perf_counter() - t) * 1e3, 2))
        stages = {k: round(v, 3) for k, v in st.items()}
        del out
    print(json.dumps({"n": n, "runs_ms": runs, "stage_ms_last": stages, "levels": None}))


if __name__ == "__main__":
    main()
