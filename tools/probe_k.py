"""Step time of the SOR filter for a list of k on two clouds (uniform 10M: grid path; six blobs 10M: Morton-tree path).
usage: GSX_LIB_PATH=... python tools/probe_k.py 25 27 [--n 10000000]   (A/B of list capacities; prints one line per k and cloud)"""
import argparse, importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("ks", type=int, nargs="+")
ap.add_argument("--n", type=int, default=10_000_000)
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--clouds", default="uniform,blobs")
ap.add_argument("--param", action="append", default=[])
a = ap.parse_args()
L = importlib.import_module("3dgsconverter_amd._lib")
ctx = L.Context(0)
sweep = [None]
for kv in a.param:
    nm, v = kv.split("=")
    if "," in v:   # one swept knob: name=v1,v2,...
        sweep = [(nm, float(x)) for x in v.split(",")]
    else:
        ctx.set_param(nm, float(v))
clouds = [c for c in [("uniform", bench.synth_uniform(a.n, 5.0, 0), False), ("blobs", bench.synth_clustered(a.n, 0), True),
                        ("floaters", bench.synth_scene_with_floaters(a.n, 0), True)] if c[0] in a.clouds.split(",")]
for name, xyz, adaptive in clouds:
    for k, sw in [(k, sw) for k in a.ks for sw in sweep]:
        if sw:
            ctx.set_param(*sw)
            name = "%s %s=%g" % (name.split(" ")[0], sw[0], sw[1])
        r = bench.run_sor(L, ctx, xyz, k, 10.5, a.steps, 3, adaptive=adaptive, groups=True)
        g = r["kernel_ms_per_step"]
        print("%s k=%d: %.3f ms/step, knn %.3f, bin %.3f, fallback %.3f, stats %.3f ms; fallback queries %s, survivors %d" % (name, k, r["ms_per_step"], r["knn_kernel_ms"], g["bin"], g["fallback"], g["stats"], r["grid"].get("n_fallback"), r["survivors"]), flush=True)
