"""Where the DataProcessor-level time goes at 10M splats: host AoS->SoA
gather, PCIe, kernels, boolean-index compaction of the 248 B/splat structured array."""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
gsx = importlib.import_module("3dgsconverter_amd")
L = gsx._lib


def t(label, fn, reps=3):
    fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        r = fn()
    dt = (time.perf_counter() - t0) / reps
    print("%-58s %8.2f ms" % (label, dt * 1e3), flush=True)
    return r


def main(n=10_000_000):
    names = ["x", "y", "z", "nx", "ny", "nz"] + ["f_dc_%d" % i for i in range(3)] + ["f_rest_%d" % i for i in range(45)] \
        + ["opacity"] + ["scale_%d" % i for i in range(3)] + ["rot_%d" % i for i in range(4)]
    dt = np.dtype([(nm, "f4") for nm in names])
    print("row bytes", dt.itemsize, "cores", os.cpu_count())
    rng = np.random.default_rng(0)
    data = np.zeros(n, dt)
    xyz = rng.random((n, 3), dtype=np.float32) * np.float32(5.0)
    for a, nm in enumerate("xyz"):
        data[nm] = xyz[:, a]
    cols = t("gather: 3 x np.ascontiguousarray(v[name])", lambda: [np.ascontiguousarray(data[nm]) for nm in "xyz"])
    t("gather: np.column_stack", lambda: np.column_stack((data["x"], data["y"], data["z"])))
    res = t("gsx_sor_filter host->host (3 columns in, mask out)", lambda: L.sor_filter(tuple(cols), 16, 1.0, want_mean=False))
    mask = res["mask"]
    print("survivors", int(mask.sum()))
    t("compaction: data[mask] (numpy)", lambda: data[mask])
    idx = np.flatnonzero(mask)
    t("compaction: np.take(data, idx)", lambda: np.take(data, idx))
    if hasattr(L, "host_gather_xyz"):
        t("native gather (threads)", lambda: L.host_gather_xyz(data))
        t("native compaction (threads)", lambda: L.host_compact_rows(data, mask))
    import importlib as il
    dp = il.import_module("3dgsconverter_amd.processing")
    rows = L.host_gather_xyz(data)
    occ = t("gsx_density_voxels host->host ((N,3) rows in)", lambda: L.density_voxels(rows, 1.1, 55000))
    kk = occ["dense_keys"]
    t("gsx_density_mask host->host", lambda: L.density_mask(rows, 1.1, kk))
    t("gsx_sor_filter host->host ((N,3) rows in)", lambda: L.sor_filter(rows, 16, 1.0, want_mean=False))
    p0 = dp.DataProcessor(data)
    t("DataProcessor.apply_density_filter(s=0.5)", lambda: dp.DataProcessor(data).apply_density_filter(sensitivity=0.5), reps=2)
    t("DataProcessor.remove_flyers(16, 1.0)", lambda: dp.DataProcessor(data).remove_flyers(16, 1.0), reps=2)
    def chain():
        p = dp.DataProcessor(data)
        p.apply_density_filter(sensitivity=0.5)
        p.remove_flyers(16, 1.0)
        return p.data
    gsx.utils.set_quiet(True) if hasattr(gsx.utils, "set_quiet") else None
    out = t("DataProcessor: density(s=0.5) + SOR(k=16) chain", chain, reps=2)
    print("chain survivors", len(out))


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000)
