"""oracle/ -- CPU restatement of the reference's algorithm for the hot path.

TEST INFRASTRUCTURE ONLY.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import anything from this package, and
only as the checker -- never as the thing measured or shipped.  The product
(`3dgsconverter_amd/`) must not import it; ``tests/test_boundary.py`` greps for
that.

Parity status (details in DESIGN.md):
  * SOR (exact KNN mean distance + threshold): PINNED -- golden vectors in
    tests/golden/ were produced by running the reference's own
    ``DataProcessor.remove_flyers`` CPU path in the build container
    (oracle/make_golden.py), and the restatements here reproduce them bit for bit.
  * voxel-density filter: PINNED the same way (``apply_density_filter``).
  * K-Means (Taichi Lloyd kernels): PINNED -- tests/golden/kmeans_ref.* were produced by the
    reference's own ``k_means_assign`` / ``k_means_update`` / ``_kmeans_taichi`` executed through
    oracle/taichi_shim.py (a test-only stand-in for the uninstallable ``taichi`` that runs the
    kernel bodies as Python with Taichi's default f32/i32 types; ``np.random.seed`` pins the
    unseeded init), by its sklearn front door under ``np.random.seed``, and by its
    ``SogFormat.write`` (bundle decoded with pillow) -- oracle/make_golden_kmeans.py.  The C
    restatement with binary32 index-order accumulation reproduces the Lloyd fixtures bit for
    bit; the GPU is compared within the floating-point tolerance of SURVEY.md 8(c) (the
    reference itself is order-nondeterministic: f32 atomics, backend-dependent FMA contraction).
"""
from __future__ import annotations

import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "libgsx_oracle.so")
_lib = None


def build(force: bool = False) -> str:
    """Compile oracle/gsx_oracle.c with gcc (idempotent)."""
    src = os.path.join(_HERE, "gsx_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B"])
    return _LIB_PATH


def clib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        build()
        lib = ctypes.CDLL(_LIB_PATH)
        c = ctypes
        lib.gsxo_pairwise_sum_f32.restype = c.c_float
        lib.gsxo_pairwise_sum_f32.argtypes = [c.c_void_p, c.c_int64]
        lib.gsxo_np_sum_f32.restype = c.c_float
        lib.gsxo_np_sum_f32.argtypes = [c.c_void_p, c.c_int64]
        lib.gsxo_pairwise_sum_f64.restype = c.c_double
        lib.gsxo_pairwise_sum_f64.argtypes = [c.c_void_p, c.c_int64]
        lib.gsxo_sor_stats_f32.restype = None
        lib.gsxo_sor_stats_f32.argtypes = [c.c_void_p, c.c_int64, c.c_double, c.c_void_p]
        lib.gsxo_sor_mean_dists_brute.restype = None
        lib.gsxo_sor_mean_dists_brute.argtypes = [c.c_void_p, c.c_int64, c.c_int, c.c_void_p]
        lib.gsxo_sor_mean_dists_brute_subset.restype = None
        lib.gsxo_sor_mean_dists_brute_subset.argtypes = [
            c.c_void_p, c.c_int64, c.c_int, c.c_void_p, c.c_int64, c.c_void_p]
        lib.gsxo_voxel_keys.restype = None
        lib.gsxo_voxel_keys.argtypes = [c.c_void_p, c.c_int64, c.c_double, c.c_void_p]
        lib.gsxo_kmeans_assign.restype = None
        lib.gsxo_kmeans_assign.argtypes = [c.c_void_p, c.c_int64, c.c_int, c.c_void_p, c.c_int, c.c_void_p]
        lib.gsxo_kmeans_update.restype = None
        lib.gsxo_kmeans_update.argtypes = [
            c.c_void_p, c.c_int64, c.c_int, c.c_void_p, c.c_int, c.c_void_p, c.c_void_p]
        lib.gsxo_kmeans_update_f32seq.restype = None
        lib.gsxo_kmeans_update_f32seq.argtypes = lib.gsxo_kmeans_update.argtypes
        _lib = lib
    return _lib
