"""Import the ACTUAL reference package from /root/reference (build container only).

TEST INFRASTRUCTURE ONLY.  /root/reference does not exist on the GPU box, so
nothing that runs there may call this; it is used by oracle/make_golden.py (to
produce tests/golden/) and by the ``-m "not gpu"`` tests that re-validate the
restatements when the reference happens to be mounted.

``plyfile`` and ``taichi`` are not installed here: a two-line
``plyfile`` stub lets ``gsconverter`` import, and the package then runs on its
CPU fallbacks (``HAS_TAICHI`` is False).
"""
from __future__ import annotations

import os
import sys
import types

REFERENCE_ROOT = "/root/reference"


def available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "gsconverter"))


def load():
    """-> (DataProcessor class, gpu_ops module, data_processor module)."""
    if not available():
        raise RuntimeError("reference not mounted at %s" % REFERENCE_ROOT)
    if "plyfile" not in sys.modules:
        stub = types.ModuleType("plyfile")
        stub.PlyData = object
        stub.PlyElement = object
        sys.modules["plyfile"] = stub
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    from gsconverter.processing import DataProcessor, gpu_ops  # noqa: E402
    import gsconverter.processing.data_processor as dpmod  # noqa: E402
    return DataProcessor, gpu_ops, dpmod


def xyz_to_struct(xyz, extra_index=True):
    import numpy as np
    dt = [("x", "f4"), ("y", "f4"), ("z", "f4")]
    if extra_index:
        dt.append(("orig_index", "i8"))
    arr = np.zeros(len(xyz), dtype=dt)
    arr["x"], arr["y"], arr["z"] = xyz[:, 0], xyz[:, 1], xyz[:, 2]
    if extra_index:
        arr["orig_index"] = np.arange(len(xyz))
    return arr


def reference_sor(xyz, k=25, threshold_factor=10.5, intensity=None):
    """Run the reference's own ``DataProcessor.remove_flyers`` CPU branch and capture the
    locals of its frame (the function computes ``mask`` at data_processor.py:180 but returns
    the unfiltered data)."""
    import numpy as np
    DataProcessor, gpu_ops, dpmod = load()
    assert not gpu_ops.HAS_TAICHI
    cap = {}
    saved = dpmod.status_print

    def spy(*a, **kw):
        f = sys._getframe(1)
        loc = f.f_locals
        if "mask" in loc and "all_mean_dists" in loc:
            cap["mask"] = np.array(loc["mask"], copy=True)
            cap["mean_dists"] = np.array(loc["all_mean_dists"], copy=True)
            cap["threshold"] = loc["threshold"]
            cap["mean"] = loc["global_mean"]
            cap["std"] = loc["global_std"]
            cap["k"] = loc["k"]
            cap["threshold_factor"] = loc["threshold_factor"]

    dpmod.status_print = spy
    try:
        with np.errstate(invalid="ignore"):
            DataProcessor(xyz_to_struct(xyz, False)).remove_flyers(k, threshold_factor, intensity=intensity)
    finally:
        dpmod.status_print = saved
    return cap


def reference_density(xyz, voxel_size=1.0, threshold_percentage=0.32, sensitivity=None, keep_multicluster=False):
    """Run the reference's own ``apply_density_filter``; the survivor mask is recovered from an
    extra ``orig_index`` field carried through the structured array."""
    import numpy as np
    DataProcessor, _, dpmod = load()
    msgs = []
    saved = dpmod.status_print
    dpmod.status_print = lambda *a, **kw: msgs.append(" ".join(map(str, a)))
    try:
        out = DataProcessor(xyz_to_struct(xyz, True)).apply_density_filter(
            voxel_size, threshold_percentage, sensitivity=sensitivity, keep_multicluster=keep_multicluster)
    finally:
        dpmod.status_print = saved
    mask = np.zeros(len(xyz), dtype=bool)
    mask[out["orig_index"]] = True
    return {"mask": mask, "messages": msgs}


def load_gpu_ops_with_taichi_shim():
    """A SECOND copy of the reference's gpu_ops.py imported with oracle/taichi_shim.py standing in for
    ``taichi`` -> ``HAS_TAICHI`` is True and ``_kmeans_taichi`` / ``k_means_assign`` / ``k_means_update``
    (gpu_ops.py:57-96,178-191) run as plain Python on numpy arrays.  The regular import (``load()``)
    is left untouched."""
    import importlib.util
    from . import taichi_shim
    load()  # puts the stubbed package in sys.modules (parent of the module created below)
    name = "gsconverter.processing._gpu_ops_taichi_shimmed"
    if name in sys.modules:
        return sys.modules[name]
    taichi_shim.install()
    try:
        spec = importlib.util.spec_from_file_location(
            name, os.path.join(REFERENCE_ROOT, "gsconverter", "processing", "gpu_ops.py"))
        mod = importlib.util.module_from_spec(spec)
        sys.modules[name] = mod
        spec.loader.exec_module(mod)
    finally:
        taichi_shim.uninstall()
    assert mod.HAS_TAICHI, "the shim did not take"
    return mod


def reference_cply(data, stable_ties=False):
    """What the reference's ``CompressedPlyFormat.write`` (formats/compressed_ply.py:128-243) hands to its PLY
    container -- the writer itself runs, only ``_write_ply_file`` (plyfile, absent here) is intercepted.
    stable_ties: run it with ``np.argsort`` made stable inside that module (the only change), which fixes the order of
    splats with equal Morton code.  -> dict(order, chunk, vertex, sh)"""
    import numpy as np
    load()
    import gsconverter.formats.compressed_ply as mod  # type: ignore

    class _StableNp:
        def __getattr__(self, name):
            return getattr(np, name)

        @staticmethod
        def argsort(a, *args, **kw):
            return np.argsort(a, kind="stable")

    got = {}
    fmt = mod.CompressedPlyFormat()
    orig_sort = fmt._sort_morton_order

    def sort_and_record(d, indices):
        orig_sort(d, indices)
        got["order"] = indices.copy()

    fmt._sort_morton_order = sort_and_record
    fmt._write_ply_file = lambda path, chunk, vertex, sh: got.update(chunk=chunk.copy(), vertex=vertex.copy(),
                                                                   sh=None if sh is None else sh.copy())
    saved = mod.np
    try:
        if stable_ties:
            mod.np = _StableNp()
        fmt.write(data, "/dev/null")
    finally:
        mod.np = saved
    return got
