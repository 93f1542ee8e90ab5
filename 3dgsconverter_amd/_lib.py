"""ctypes binding of libgsx_hip.so (C ABI declared in include/gsx_hip.h).

numpy + ctypes only -- no torch, no Taichi.  The library is built in-tree by
``3dgsconverter_amd/build.py`` (hipcc, gfx950).  There is deliberately NO CPU
fallback anywhere in this package: if the library or a GPU is missing the calls
raise ``GsxError`` (a RuntimeError).
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# GSX_LIB_PATH: an experiment build of build.py (GSX_EXTRA_FLAGS -> 3dgsconverter_amd/variants/), for A/B runs only
LIB_PATH = os.environ.get("GSX_LIB_PATH") or os.path.join(_HERE, "libgsx_hip.so")

KNN_AUTO, KNN_BRUTE, KNN_GRID, KNN_TREE = 0, 1, 2, 3
T_SOR_KNN, T_SOR_BIN, T_SOR_FALLBACK, T_SOR_STATS, T_DENSITY, T_KMEANS_ASSIGN, T_KMEANS_UPDATE, T_QUANTIZE = range(8)
T_SLAB_PREP, T_SLAB_ROWS, T_SLAB_MEANS, T_SLAB_COLL = range(8, 12)   # multi-GPU slab step (round 5)


class GsxError(RuntimeError):
    """Raised for every failure of the HIP path (missing library, no GPU, HIP error)."""


class SorInfo(C.Structure):
    _fields_ = [("algo", C.c_int32), ("grid_dim", C.c_int32 * 3), ("cell_size", C.c_float),
                ("n_cells", C.c_int64), ("n_bricks", C.c_int64), ("n_fallback", C.c_int64),
                ("n_exhaustive", C.c_int64), ("n_deferred_bricks", C.c_int64), ("n_refined", C.c_int64)]

    def as_dict(self):
        return {"algo": int(self.algo), "grid_dim": tuple(int(v) for v in self.grid_dim),
                "cell_size": float(self.cell_size), "n_cells": int(self.n_cells), "n_bricks": int(self.n_bricks),
                "n_fallback": int(self.n_fallback), "n_exhaustive": int(self.n_exhaustive),
                "n_deferred_bricks": int(self.n_deferred_bricks), "n_refined": int(self.n_refined)}


class SlabPlan(C.Structure):
    """gsx_slab_plan_t (include/gsx_hip.h)"""
    _fields_ = [("status", C.c_int32), ("world", C.c_int32), ("rank", C.c_int32), ("axis", C.c_int32), ("halo_bins", C.c_int32),
                ("lo", C.c_float), ("hi", C.c_float), ("plane_lo", C.c_float), ("plane_hi", C.c_float), ("cut", C.c_int32 * 17),
                ("n_local", C.c_int64), ("n_total", C.c_int64), ("n_own", C.c_int64), ("n_halo", C.c_int64), ("n_send", C.c_int64),
                ("halo_total", C.c_int64), ("sizes", C.c_int64 * 16),
                ("own_cnt", C.c_int64 * 16), ("halo_cnt", C.c_int64 * 16), ("own_off", C.c_int64 * 16), ("halo_off", C.c_int64 * 16),
                ("in_own", C.c_int64 * 16), ("in_halo", C.c_int64 * 16), ("r_own_off", C.c_int64 * 16), ("r_halo_off", C.c_int64 * 16)]

    def as_dict(self):
        g = int(self.world)
        out = {k: (int(getattr(self, k)) if k not in ("lo", "hi", "plane_lo", "plane_hi") else float(getattr(self, k)))
               for k in ("status", "world", "rank", "axis", "halo_bins", "lo", "hi", "plane_lo", "plane_hi", "n_local", "n_total", "n_own",
                         "n_halo", "n_send", "halo_total")}
        out["cut"] = [int(v) for v in self.cut[:g + 1]]
        for k in ("sizes", "own_cnt", "halo_cnt", "own_off", "halo_off", "in_own", "in_halo", "r_own_off", "r_halo_off"):
            out[k] = [int(v) for v in getattr(self, k)[:g]]
        return out


class SlabStep(C.Structure):
    """gsx_slab_step_t (include/gsx_hip.h)"""
    _fields_ = [("status", C.c_int32), ("plan", SlabPlan), ("mask_dev", C.c_void_p), ("mean_dists_dev", C.c_void_p),
                ("stats_dev", C.c_void_p), ("uncertain_dev", C.c_void_p)]


class DensityInfo(C.Structure):
    """gsx_density_info (include/gsx_hip.h)"""
    _fields_ = [("status", C.c_int32), ("n_unique", C.c_int64), ("n_dense", C.c_int64), ("n_kept_voxels", C.c_int64),
                ("kept_clusters", C.c_int64), ("largest", C.c_int64)]


SOG_FIELDS = 59   # GSX_SOG_FIELDS: x y z | rot_0..3 | scale_0..2 | f_dc_0..2 | opacity | f_rest_0..44
SOG_FIELD_NAMES = (["x", "y", "z", "rot_0", "rot_1", "rot_2", "rot_3", "scale_0", "scale_1", "scale_2", "f_dc_0", "f_dc_1", "f_dc_2", "opacity"]
                   + ["f_rest_%d" % i for i in range(45)])


class SogLayout(C.Structure):
    """gsx_sog_layout (include/gsx_hip.h): where the writer's float32 fields sit inside a row of the structured table"""
    _fields_ = [("row_bytes", C.c_int64), ("n_rest", C.c_int32), ("offset", C.c_int32 * SOG_FIELDS)]


class SogScan(C.Structure):
    """gsx_sog_scan (include/gsx_hip.h)"""
    _fields_ = [("vmin", C.c_float * 3), ("vmax", C.c_float * 3), ("nonfinite", C.c_uint32), ("reserved", C.c_uint32),
                ("rest_nonzero", C.c_uint64)]


DENSITY_OK, DENSITY_EMPTY, DENSITY_HOST = range(3)
SLAB_OK, SLAB_EMPTY, SLAB_NONFINITE, SLAB_SMALL_SHARD, SLAB_NO_STRUCTURE = range(5)
COMM_F32_MAX, COMM_F32_SUM, COMM_I64_SUM, COMM_F64_MAX, COMM_I64_MIN = range(5)

# name -> (restype, argtypes); every symbol include/gsx_hip.h declares
_P, _I64, _I, _D = C.c_void_p, C.c_int64, C.c_int, C.c_double
SIGNATURES = {
    "gsx_version": (C.c_char_p, []),
    "gsx_last_error": (C.c_char_p, []),
    "gsx_device_count": (_I, []),
    "gsx_device_uid": (_I, [_I, C.c_char_p, _I]),
    "gsx_ctx_create": (_I, [_I, C.POINTER(_P)]),
    "gsx_ctx_destroy": (None, [_P]),
    "gsx_ctx_set_stream": (_I, [_P, _P]),
    "gsx_ctx_own_stream": (_I, [_P]),
    "gsx_ctx_last_knn_algo": (_I, [_P]),
    "gsx_ctx_synchronize": (_I, [_P]),
    "gsx_ctx_check": (_I, [_P]),
    "gsx_ctx_set_timing": (_I, [_P, _I]),
    "gsx_ctx_reset_timing": (_I, [_P]),
    "gsx_ctx_get_timing": (_I, [_P, _I, C.POINTER(C.c_uint64), C.POINTER(_D)]),
    "gsx_ctx_set_param": (_I, [_P, C.c_char_p, _D]),
    "gsx_dev_malloc": (_I, [_P, C.c_size_t, C.POINTER(_P)]),
    "gsx_dev_free": (_I, [_P, _P]),
    "gsx_dev_upload": (_I, [_P, _P, _P, C.c_size_t]),
    "gsx_dev_download": (_I, [_P, _P, _P, C.c_size_t]),
    "gsx_dev_copy": (_I, [_P, _P, _P, C.c_size_t]),
    "gsx_dev_upload_async": (_I, [_P, _P, _P, C.c_size_t]),
    "gsx_dev_memset": (_I, [_P, _P, _I, C.c_size_t]),
    "gsx_fields_nonzero_dev": (_I, [_P, _P, _I64, _I64, _I, C.POINTER(C.c_uint64)]),
    "gsx_rows_repack_dev": (_I, [_P, _P, _I64, _I64, _P, _I64]),
    "gsx_host_pinned_alloc": (_I, [_P, C.c_size_t, C.POINTER(C.c_void_p)]),
    "gsx_host_pinned_free": (_I, [_P, _P]),
    "gsx_dev_upload_staged": (_I, [_P, _P, _P, C.c_size_t]),
    "gsx_dev_download_staged": (_I, [_P, _P, _P, C.c_size_t]),
    "gsx_host_gather_f32": (_I, [_P, _I64, _I64, C.POINTER(_I64), _I, _P]),
    "gsx_host_gather_columns_f32": (_I, [_P, _I64, _I64, C.POINTER(_I64), _I, _P]),
    "gsx_host_compact_rows": (_I, [_P, _I64, _I64, _P, _P, _I64, C.POINTER(_I64)]),
    "gsx_host_take_rows": (_I, [_P, _I64, _I64, _P, _I64, _P]),
    "gsx_host_zero_columns": (_I, [_P, _I64, _I64, C.POINTER(_I64), _I]),
    "gsx_host_append_columns": (_I, [_P, _I64, _I64, _P, _I64, _P, _I64]),
    "gsx_host_take_rows_append": (_I, [_P, _I64, _I64, _P, _I64, _P, _I64, _P, _I64]),
    "gsx_host_take_rows_shape": (_I, [_P, _I64, _I64, _P, _I64, _P, _I64, C.POINTER(_I64), _I, _P, _I64]),
    "gsx_rgb_from_sh": (_I, [_P, _I64, _P, _P]),
    "gsx_rgb_from_sh_dev": (_I, [_P, _P, _I64, _P, _P]),
    "gsx_rgb_from_sh_list": (_I, [_P, _I64, _P, _P, _I64, C.POINTER(_I64)]),
    "gsx_rgb_from_sh_list_dev": (_I, [_P, _P, _I64, _P, _P, _I64, _P]),
    "gsx_compact_rows_dev": (_I, [_P, _P, _P, _P, _I64, _P, _P, C.POINTER(_I64)]),
    "gsx_mask_bbox_dev": (_I, [_P, _P, _I64, _P, _P]),
    "gsx_mask_ge_dev": (_I, [_P, _P, _P, _I64, _D, _P]),
    "gsx_sor_knn_dev": (_I, [_P, _P, _P, _P, _I64, _I64, _I64, _I64, _I, _I, _P, C.POINTER(SorInfo)]),
    "gsx_sor_knn_share_dev": (_I, [_P, _P, _P, _P, _I64, _I64, _I, _I, _I, _I, _P, C.POINTER(SorInfo)]),
    "gsx_sor_stats_dev": (_I, [_P, _P, _I64, _D, _P]),
    "gsx_sor_mask_dev": (_I, [_P, _P, _I64, _P, _P]),
    "gsx_comm_unique_id": (_I, [_P]),
    "gsx_comm_init": (_I, [_P, _I, _I, _P]),
    "gsx_comm_destroy": (_I, [_P]),
    "gsx_comm_all_reduce": (_I, [_P, _P, _I64, _I]),
    "gsx_comm_all_gather": (_I, [_P, _P, _P, _I64]),
    "gsx_comm_all_to_all_v": (_I, [_P, _P, _P, _P, _P, _P, _P, _I]),
    "gsx_comm_all_to_all_segs": (_I, [_P, _P, _P, _I, _P, _P, _P, _P, _I]),
    "gsx_comm_abort": (_I, [_P]),
    "gsx_comm_transport": (_I, [_P]),
    "gsx_comm_rank": (_I, [_P, C.POINTER(_I), C.POINTER(_I)]),
    "gsx_comm_barrier": (_I, [_P]),
    "gsx_slab_plan": (_I, [_P, _I, _I, _I64, _I, _D, C.POINTER(SlabPlan)]),
    "gsx_sor_slab_step_dev": (_I, [_P, _P, _I64, _I, _D, _D, _P, C.POINTER(SlabStep)]),
    "gsx_slab_bbox_dev": (_I, [_P, _P, _P, _P, _I64, _I64, _P]),
    "gsx_slab_hist_dev": (_I, [_P, _P, _P, _P, _I64, _I64, _P, _P]),
    "gsx_slab_partition_dev": (_I, [_P, _P, _P, _P, _I64, _I64, _I, _I, C.c_float, C.c_float, _P, _I, _P, _P, _P, _P, _P]),
    "gsx_sor_knn_slab_dev": (_I, [_P, _P, _I64, _I64, _I, _P, _P]),
    "gsx_slab_certify_dev": (_I, [_P, _P, _I64, _I64, _P, C.c_float, C.c_float, _P]),
    "gsx_slab_unpermute_dev": (_I, [_P, _P, _P, _I64, _P]),
    "gsx_sor_piece_sums_dev": (_I, [_P, _P, _I64, _P, _P]),
    "gsx_sor_stats_from_pieces_dev": (_I, [_P, _P, _I64, _I64, _I, _D, _P]),
    "gsx_sor_filter": (_I, [_P, _P, _P, _I64, _I64, _I, _D, _I, _P, _P, _P, C.POINTER(SorInfo)]),
    "gsx_density_voxels": (_I, [_P, _P, _P, _I64, _I64, _D, _I64, _I64, C.POINTER(_I64), C.POINTER(_I64), _P, _P]),
    "gsx_density_mask": (_I, [_P, _P, _P, _I64, _I64, _D, _P, _I64, _P]),
    "gsx_kmeans_lloyd": (_I, [_P, _I64, _I, _I, _I, _P, _P, _P]),
    "gsx_quantize_sorted_codebook": (_I, [_P, _I64, _P, _I, _P]),
    "gsx_kmeans_pp": (_I, [_P, _I64, _I, _I, _P, _I, _P]),
    "gsx_kmeans_pp_dev": (_I, [_P, _P, _I64, _I, _I, _P, _I, _P]),
    "gsx_kmeans1d": (_I, [_P, _I64, _I, _I, _P, _P, _P]),
    "gsx_kmeans1d_dev": (_I, [_P, _P, _I64, _I, _I, _I, _P, _P, _P]),
    "gsx_lexsort3": (_I, [_P, _P, _P, _I64, _P]),
    "gsx_lexsort3_dev": (_I, [_P, _P, _P, _P, _I64, _I64, _P]),
    "gsx_sog_quats": (_I, [_P, _I64, _P]),
    "gsx_sog_positions": (_I, [_P, _I64, C.c_float, C.c_float, _P, _P]),
    "gsx_sog_positions_dev": (_I, [_P, _P, _I64, C.c_float, C.c_float, _P, _P]),
    "gsx_sog_alpha": (_I, [_P, _I64, _P, _P]),
    "gsx_sog_alpha_dev": (_I, [_P, _P, _I64, _P, _P]),
    "gsx_sog_quats_dev": (_I, [_P, _P, _I64, _P]),
    "gsx_sog_scan_dev": (_I, [_P, _P, C.POINTER(SogLayout), _I64, _P, C.POINTER(SogScan)]),
    "gsx_sog_extremes_dev": (_I, [_P, _P, _I64, _P, _P, _I, _P, _P]),
    "gsx_sog_order_dev": (_I, [_P, _P, _I64, _P]),
    "gsx_sog_gather_dev": (_I, [_P, _P, C.POINTER(SogLayout), _P, _I64, _I, _P, _P, _P, _P, _P, _P]),
    "gsx_sog_means_texels_dev": (_I, [_P, _P, _I64, _I64, _P, _P, _P, _P, _P, _P, _P, _I64, _P]),
    "gsx_sog_quats_texels_dev": (_I, [_P, _P, _I64, _I64, _P]),
    "gsx_sog_codes_texels_dev": (_I, [_P, _P, _I64, _I64, _P, _I, _P, _P, _P, _I64, _P]),
    "gsx_sog_labels_texels_dev": (_I, [_P, _P, _I64, _I64, _I64, _I, _P]),
    "gsx_gather_rows_dev": (_I, [_P, _P, _I, _P, _I64, _P]),
    "gsx_morton_order_dev": (_I, [_P, _P, _P, _P, _I64, _I64, _P, C.POINTER(_I)]),
    "gsx_cply_pack_dev": (_I, [_P, _P, _P, _I64, _P, _P]),
    "gsx_cply_sh_dev": (_I, [_P, _P, _I, _I64, _P, _I64, _P]),
    "gsx_cply_pack_strided_dev": (_I, [_P, _P, _P, _P, _I64, _P, _P]),
    "gsx_cply_pack_opacity_dev": (_I, [_P, _P, _P, _P, _I64, _P, _P, _P, _I64, _P]),
    "gsx_cply_sh_strided_dev": (_I, [_P, _P, _I, _I64, _I64, _P, _I64, _P]),
    "gsx_density_voxels_dev": (_I, [_P, _P, _P, _P, _I64, _I64, _D, _I64, _I64, C.POINTER(_I64), C.POINTER(_I64), _P, _P]),
    "gsx_density_mask_dev": (_I, [_P, _P, _P, _P, _I64, _I64, _D, _P, _I64, _P]),
    "gsx_density_filter_dev": (_I, [_P, _P, _P, _P, _I64, _I64, _D, _I64, _I, _P, _P, C.POINTER(DensityInfo)]),
    "gsx_density_hist_dev": (_I, [_P, _P, _P, _P, _I64, _I64, _D, _I64, C.POINTER(_I64), _P, _P]),
    "gsx_density_merge_dev": (_I, [_P, _P, _P, _I64, _I64, _I64, C.POINTER(_I64), C.POINTER(_I64), _P, _P]),
    "gsx_kmeans_lloyd_dev": (_I, [_P, _P, _I64, _I, _I, _I, _P, _P]),
    "gsx_kmeans_lloyd_batch_dev": (_I, [_P, _P, _P, _I, _I, _I, _I, _P, _P]),
    "gsx_quantize_sorted_codebook_dev": (_I, [_P, _P, _I64, _P, _I, _P]),
}

_lib = None


def load() -> C.CDLL:
    """dlopen the in-tree library and bind every declared symbol (raises GsxError if absent)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise GsxError("libgsx_hip.so is not built: run `python 3dgsconverter_amd/build.py` "
                       "(hipcc --offload-arch=gfx950); there is no CPU fallback")
    try:
        lib = C.CDLL(LIB_PATH)
    except OSError as e:  # e.g. libamdhip64 missing
        raise GsxError("cannot load %s: %s" % (LIB_PATH, e)) from e
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise GsxError("libgsx_hip.so does not export %s (stale build?)" % name) from e
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def last_error() -> str:
    return (load().gsx_last_error() or b"").decode("utf-8", "replace")


def check(rc: int, what: str):
    if rc != 0:
        raise GsxError("%s failed: %s" % (what, last_error()))


def device_count() -> int:
    try:
        return int(load().gsx_device_count())
    except GsxError:
        return 0


def device_uid(device: int) -> str:
    """PCI bus id of a visible device: equal strings <=> the same physical GPU"""
    buf = C.create_string_buffer(64)
    check(load().gsx_device_uid(device, buf, 64), "gsx_device_uid")
    return buf.value.decode()


def has_hip() -> bool:
    return device_count() > 0


def require_hip():
    lib = load()
    if lib.gsx_device_count() <= 0:
        raise GsxError("no MI355X (gfx950) device visible to HIP; the HIP path has no CPU fallback")
    return lib


def _xyz_pointers(xyz_or_cols):
    """Accept an (N,3) float32 C-contiguous array (the reference's `coords`, data_processor.py:139)
    or a tuple of three contiguous float32 columns. -> (keepalive, px, py, pz, stride, n)"""
    if isinstance(xyz_or_cols, (tuple, list)):
        cols = [np.ascontiguousarray(c, dtype=np.float32) for c in xyz_or_cols]
        if len(cols) != 3 or len({len(c) for c in cols}) != 1:
            raise ValueError("expected three equally long columns")
        return cols, cols[0].ctypes.data, cols[1].ctypes.data, cols[2].ctypes.data, 1, len(cols[0])
    a = np.ascontiguousarray(xyz_or_cols, dtype=np.float32)
    if a.ndim != 2 or a.shape[1] != 3:
        raise ValueError("Requires 3D data")  # same message as gpu_ops.py:198
    base = a.ctypes.data
    return a, base, base + 4, base + 8, 3, a.shape[0]


def host_gather_xyz(vertices: np.ndarray, names=("x", "y", "z"), out: "np.ndarray | None" = None) -> np.ndarray:
    """coords = np.column_stack((v['x'], v['y'], v['z'])) (data_processor.py:38,139) as float32 (N,3),
    threaded (C ABI gsx_host_gather_f32).  Any layout the C routine cannot take goes through numpy.  out: a C-contiguous
    (N, len(names)) float32 array to fill (a pinned staging buffer)"""
    fields = vertices.dtype.fields or {}
    ok = (vertices.ndim == 1 and vertices.flags.c_contiguous and
          all(nm in fields and fields[nm][0] == np.dtype("<f4") for nm in names))
    if not ok:
        res = np.column_stack([np.asarray(vertices[nm], dtype=np.float32) for nm in names])
        if out is not None:
            out[...] = res
            return out
        return res
    n = len(vertices)
    if out is None:
        out = np.empty((n, len(names)), dtype=np.float32)
    elif out.shape != (n, len(names)) or out.dtype != np.float32 or not out.flags.c_contiguous:
        raise ValueError("out must be a C-contiguous (n, %d) float32 array" % len(names))
    if n == 0:
        return out
    offs = (_I64 * len(names))(*[int(fields[nm][1]) for nm in names])
    check(load().gsx_host_gather_f32(vertices.ctypes.data, vertices.dtype.itemsize, n, offs, len(names), out.ctypes.data),
          "gsx_host_gather_f32")
    return out


def host_gather_columns(vertices: np.ndarray, names) -> np.ndarray:
    """(len(names), n) float32, row c = np.ascontiguousarray(vertices[names[c]]): every column a writer uploads, in ONE threaded
    pass over the table (C ABI gsx_host_gather_columns_f32) instead of one strided numpy copy per column"""
    names = list(names)
    fields = vertices.dtype.fields or {}
    n = len(vertices)
    ok = (vertices.ndim == 1 and vertices.flags.c_contiguous and 1 <= len(names) <= 64 and
          all(nm in fields and fields[nm][0] == np.dtype("<f4") for nm in names))
    if not ok or n == 0:
        out = np.empty((len(names), n), dtype=np.float32)
        for i, nm in enumerate(names):
            out[i] = vertices[nm]
        return out
    out = np.empty((len(names), n), dtype=np.float32)
    offs = (_I64 * len(names))(*[int(fields[nm][1]) for nm in names])
    check(load().gsx_host_gather_columns_f32(vertices.ctypes.data, vertices.dtype.itemsize, n, offs, len(names), out.ctypes.data),
          "gsx_host_gather_columns_f32")
    return out


def host_compact_rows(rows: np.ndarray, mask: np.ndarray) -> np.ndarray:
    """rows[mask] (data_processor.py:114,149) for a 1-D array of any (structured) dtype, threaded
    (C ABI gsx_host_compact_rows).  Same result as numpy's boolean indexing: a new array, order kept."""
    mask = np.asarray(mask)
    if rows.ndim != 1 or mask.shape != rows.shape or mask.dtype != np.bool_ or rows.dtype.hasobject:
        return rows[mask]
    src = np.ascontiguousarray(rows)
    m8 = np.ascontiguousarray(mask).view(np.uint8)
    keep = int(np.count_nonzero(m8))
    out = np.empty(keep, dtype=rows.dtype)
    n_out = _I64()
    if len(src):
        check(load().gsx_host_compact_rows(src.ctypes.data, src.dtype.itemsize, len(src), m8.ctypes.data, out.ctypes.data,
                                           keep, C.byref(n_out)), "gsx_host_compact_rows")
        assert n_out.value == keep
    return out


def host_take_rows(rows: np.ndarray, idx: np.ndarray) -> np.ndarray:
    """rows[idx] for a strictly ascending uint32 index list (the device chain's survivor list) == rows[mask], threaded
    (C ABI gsx_host_take_rows): a new array, order kept, no boolean mask in between"""
    idx = np.ascontiguousarray(idx, dtype=np.uint32)
    if rows.ndim != 1 or rows.dtype.hasobject:
        return rows[idx]
    src = np.ascontiguousarray(rows)
    out = np.empty(len(idx), dtype=rows.dtype)
    if len(idx):
        check(load().gsx_host_take_rows(src.ctypes.data, src.dtype.itemsize, len(src), idx.ctypes.data, len(idx), out.ctypes.data),
              "gsx_host_take_rows")
    return out


def host_zero_columns(rows: np.ndarray, names) -> None:
    """rows[name] = 0.0 for every float32 field in `names`, in place (data_processor.py:310-313), one threaded pass"""
    fields = rows.dtype.fields or {}
    names = [nm for nm in names if nm in fields]
    if not names or len(rows) == 0:
        return
    ok = rows.ndim == 1 and rows.flags.c_contiguous and rows.flags.writeable and all(fields[nm][0].itemsize == 4 for nm in names)
    if not ok:
        for nm in names:
            rows[nm] = 0.0
        return
    offs = (_I64 * len(names))(*[int(fields[nm][1]) for nm in names])
    check(load().gsx_host_zero_columns(rows.ctypes.data, rows.dtype.itemsize, len(rows), offs, len(names)), "gsx_host_zero_columns")


def host_append_u8_columns(rows: np.ndarray, names, columns: np.ndarray) -> np.ndarray:
    """a new structured array = every field of `rows` + the u1 fields `names` filled from columns (n, len(names)) uint8
    (data_processor.py:262-274), one threaded pass over the rows"""
    new_dtype = np.dtype(rows.dtype.descr + [(nm, "u1") for nm in names])
    cols = np.ascontiguousarray(columns, dtype=np.uint8).reshape(len(rows), len(names))
    out = np.empty(len(rows), dtype=new_dtype)
    tail_ok = all(new_dtype.fields[nm][1] == rows.dtype.itemsize + i for i, nm in enumerate(names))
    if rows.ndim != 1 or not rows.flags.c_contiguous or rows.dtype.hasobject or not tail_ok or len(rows) == 0:
        for nm in rows.dtype.names:
            out[nm] = rows[nm]
        for i, nm in enumerate(names):
            out[nm] = cols[:, i]
        return out
    check(load().gsx_host_append_columns(rows.ctypes.data, rows.dtype.itemsize, len(rows), cols.ctypes.data, len(names),
                                         out.ctypes.data, new_dtype.itemsize), "gsx_host_append_columns")
    return out


def host_take_rows_shape(rows: np.ndarray, idx: np.ndarray, names=(), columns: "np.ndarray | None" = None, zero_names=()) -> np.ndarray:
    """rows[idx] with the float32 fields `zero_names` set to 0.0 and, when `names` / `columns` are given, the u1 fields `names`
    appended from columns[idx] -- compaction, cap_sh_degree and add_rgb_from_sh (data_processor.py:114,149 / :310-313 / :262-274) in
    ONE threaded pass (C ABI gsx_host_take_rows_shape)"""
    idx = np.ascontiguousarray(idx, dtype=np.uint32)
    names = tuple(names)
    fields = rows.dtype.fields or {}
    zero_names = [nm for nm in zero_names if nm in fields]
    new_dtype = np.dtype(rows.dtype.descr + [(nm, "u1") for nm in names]) if names else rows.dtype
    tail_ok = all(new_dtype.fields[nm][1] == rows.dtype.itemsize + i for i, nm in enumerate(names))
    plain = (rows.ndim == 1 and rows.flags.c_contiguous and not rows.dtype.hasobject and tail_ok
             and all(fields[nm][0].itemsize == 4 for nm in zero_names))
    if not plain:
        out = host_take_rows_append_u8(rows, idx, names, columns) if names else host_take_rows(rows, idx)
        host_zero_columns(out, zero_names)
        return out
    cols = np.ascontiguousarray(columns, dtype=np.uint8).reshape(len(rows), len(names)) if names else None
    out = np.empty(len(idx), dtype=new_dtype)
    if len(idx):
        offs = (_I64 * max(len(zero_names), 1))(*[int(fields[nm][1]) for nm in zero_names])
        check(load().gsx_host_take_rows_shape(rows.ctypes.data, rows.dtype.itemsize, len(rows), idx.ctypes.data, len(idx),
                                              cols.ctypes.data if names else None, len(names), offs, len(zero_names), out.ctypes.data,
                                              new_dtype.itemsize), "gsx_host_take_rows_shape")
    return out


def host_take_rows_append_u8(rows: np.ndarray, idx: np.ndarray, names, columns: np.ndarray) -> np.ndarray:
    """host_append_u8_columns(host_take_rows(rows, idx), names, columns[idx]) in ONE threaded pass (C ABI gsx_host_take_rows_append):
    the filters' compaction and add_rgb_from_sh's widened copy (data_processor.py:114,149 + :262-274) without the table in between.
    columns: (len(rows), len(names)) uint8, indexed like `rows`"""
    idx = np.ascontiguousarray(idx, dtype=np.uint32)
    new_dtype = np.dtype(rows.dtype.descr + [(nm, "u1") for nm in names])
    cols = np.ascontiguousarray(columns, dtype=np.uint8).reshape(len(rows), len(names))
    tail_ok = all(new_dtype.fields[nm][1] == rows.dtype.itemsize + i for i, nm in enumerate(names))
    if rows.ndim != 1 or not rows.flags.c_contiguous or rows.dtype.hasobject or not tail_ok:
        return host_append_u8_columns(host_take_rows(rows, idx), names, cols[idx])
    out = np.empty(len(idx), dtype=new_dtype)
    if len(idx):
        check(load().gsx_host_take_rows_append(rows.ctypes.data, rows.dtype.itemsize, len(rows), idx.ctypes.data, len(idx), cols.ctypes.data,
                                               len(names), out.ctypes.data, new_dtype.itemsize), "gsx_host_take_rows_append")
    return out


def nonzero_bytes(flags: np.ndarray) -> np.ndarray:
    """np.flatnonzero of a SPARSE uint8 flag array, eight flags per look (the uncertain-element flags of the certificate kernels:
    a few hundred set bytes in 30M; numpy's byte-wise scan takes 5x as long)"""
    flags = np.ascontiguousarray(flags).reshape(-1)
    n = len(flags)
    m = n // 8 if flags.ctypes.data % 8 == 0 else 0
    parts = []
    if m:
        words = np.flatnonzero(flags[:8 * m].view(np.uint64))
        if len(words):
            r, c = np.nonzero(flags[:8 * m].reshape(m, 8)[words])
            parts.append(words[r] * 8 + c)
    tail = np.flatnonzero(flags[8 * m:])
    if len(tail):
        parts.append(tail + 8 * m)
    return np.concatenate(parts).astype(np.int64) if parts else np.zeros(0, np.int64)


def rgb_from_sh(f_dc: np.ndarray, stats: dict | None = None) -> np.ndarray:
    """one channel of data_processor.py:316-343 -> u8[n], byte-identical to numpy (C ABI gsx_rgb_from_sh: float64 power on
    the GPU + rounding certificate; numpy's own expression for the flagged elements)"""
    lib = require_hip()
    v = np.ascontiguousarray(f_dc, dtype=np.float32)
    n = len(v)
    out = np.empty(n, dtype=np.uint8)
    if n == 0:
        return out
    # the uncertain elements as a compact index list (about 1e-4 of the values); a table with more of them than the list holds (NaNs by
    # the million) takes the flag-per-element form
    cap = n // 64 + 4096
    lst = np.empty(cap, dtype=np.uint32)
    cnt = _I64(0)
    check(lib.gsx_rgb_from_sh_list(v.ctypes.data, n, out.ctypes.data, lst.ctypes.data, cap, C.byref(cnt)), "gsx_rgb_from_sh_list")
    if cnt.value <= cap:
        idx = np.sort(lst[:cnt.value]).astype(np.int64)
    else:
        unc = np.empty(n, dtype=np.uint8)
        check(lib.gsx_rgb_from_sh(v.ctypes.data, n, out.ctypes.data, unc.ctypes.data), "gsx_rgb_from_sh")
        idx = nonzero_bytes(unc)
    if len(idx):
        with np.errstate(all="ignore"):
            lin = np.clip(0.5 + v[idx] * 0.28209479177387814, 0.0, 1.0)
            out[idx] = (np.power(lin, 1.0 / 2.2) * 255).astype(np.uint8)
    if stats is not None:
        stats["uncertain"] = stats.get("uncertain", 0) + int(len(idx))
        stats["n"] = stats.get("n", 0) + n
    return out


_numpy_reduction_checked = None


def numpy_reduction_selfcheck(ctx: "Context | None" = None) -> bool:
    """The SOR threshold is bit-exact only if the device reproduces THIS process's numpy float32 reductions
    (csrc/sor_stats.hip restates numpy 2.2's rule: 8192-element buffer pieces added sequentially, pairwise inside a piece).
    The reference does not pin numpy, so the rule is probed once per process: np.mean / np.std of a 20 001-element
    float32 array with a wide dynamic range against gsx_sor_stats_dev.  A mismatch is reported with a RuntimeWarning (masks
    can then differ from this numpy's for mean distances within one ulp of the threshold) -- or, under GSX_STRICT_NUMPY=1
    (what tests/conftest.py sets), raised as a GsxError; nothing falls back.  Supported: numpy 1.22 ... 2.2 reduce this way."""
    global _numpy_reduction_checked
    if _numpy_reduction_checked is not None:
        return _numpy_reduction_checked
    import warnings
    rng = np.random.default_rng(8192)
    a = (rng.standard_normal(20001) * np.exp(rng.uniform(-6, 6, 20001))).astype(np.float32)
    a = np.abs(a) + np.float32(1e-3)
    own = ctx is None
    ctx = ctx or Context(0)
    try:
        d, st = ctx.alloc(a.nbytes + 16).upload(a), ctx.alloc(16)
        ctx.sor_stats(d.ptr, len(a), 1.5, st.ptr)
        got = st.download(np.float32, 3)
        d.free()
        st.free()
    finally:
        if own:
            ctx.close()
    m, sd = np.mean(a), np.std(a)
    want = np.array([m, sd, m + 1.5 * sd], dtype=np.float32)
    _numpy_reduction_checked = bool(np.array_equal(got.view(np.uint32), want.view(np.uint32)))
    if not _numpy_reduction_checked:
        msg = ("numpy %s reduces float32 arrays in a different order than the one libgsx_hip reproduces (numpy 2.2: 8192-element "
               "pieces, pairwise inside): SOR thresholds would differ from this numpy's in the last bit, hence masks for mean "
               "distances within one ulp of the threshold (device %r, numpy %r).  GSX_STRICT_NUMPY=1 turns this warning into an error."
               % (np.__version__, got.tolist(), want.tolist()))
        # a drop-in keeps running where the reference runs: a warning by default; the bit-exact-mask contract
        # cannot be kept on this numpy, so tests / CI / anyone who needs the contract opt into the hard error
        if os.environ.get("GSX_STRICT_NUMPY") == "1":
            _numpy_reduction_checked = None
            raise GsxError(msg)
        warnings.warn(msg, RuntimeWarning, stacklevel=2)
    return _numpy_reduction_checked


def sor_filter(xyz, k: int, threshold_factor: float, algo: int = KNN_AUTO, want_mean: bool = True,
               want_info: bool = False):
    """Whole SOR on one GPU from host buffers (C ABI gsx_sor_filter).
    -> dict(mask bool[N], mean_dists f32[N] | None, mean, std, threshold (np.float32), info | None)"""
    lib = require_hip()
    numpy_reduction_selfcheck()
    keep, px, py, pz, stride, n = _xyz_pointers(xyz)
    mask = np.empty(n, dtype=np.uint8)
    mean = np.empty(n, dtype=np.float32) if want_mean else None
    stats = np.empty(3, dtype=np.float32)
    info = SorInfo() if want_info else None
    rc = lib.gsx_sor_filter(px, py, pz, stride, n, int(k), float(threshold_factor), int(algo),
                            mask.ctypes.data, mean.ctypes.data if want_mean else None, stats.ctypes.data,
                            C.byref(info) if want_info else None)
    check(rc, "gsx_sor_filter")
    del keep
    return {"mask": mask.view(np.bool_), "mean_dists": mean, "mean": stats[0], "std": stats[1],
            "threshold": stats[2], "info": info.as_dict() if want_info else None}


def density_voxels(xyz, voxel_size: float, min_points: int, dense_cap: int | None = None):
    """Voxel occupancy on the GPU (C ABI gsx_density_voxels).
    -> dict(n_unique, dense_keys int64[M,3] in np.unique(axis=0) order, dense_counts int64[M])"""
    lib = require_hip()
    keep, px, py, pz, stride, n = _xyz_pointers(xyz)
    if dense_cap is None:
        # count >= min_points  =>  at most n / max(min_points,1) dense voxels
        dense_cap = int(min(n, n // max(int(min_points), 1) + 1))
    keys = np.empty((max(dense_cap, 1), 3), dtype=np.int64)
    counts = np.empty(max(dense_cap, 1), dtype=np.int64)
    n_unique = C.c_int64()
    n_dense = C.c_int64()
    rc = lib.gsx_density_voxels(px, py, pz, stride, n, float(voxel_size), int(min_points), int(dense_cap),
                                C.byref(n_unique), C.byref(n_dense), keys.ctypes.data, counts.ctypes.data)
    check(rc, "gsx_density_voxels")
    del keep
    m = int(n_dense.value)
    return {"n_unique": int(n_unique.value), "dense_keys": keys[:m].copy(), "dense_counts": counts[:m].copy()}


def density_mask(xyz, voxel_size: float, kept_keys: np.ndarray) -> np.ndarray:
    """mask[i] = voxel(i) in kept_keys (C ABI gsx_density_mask)."""
    lib = require_hip()
    keep, px, py, pz, stride, n = _xyz_pointers(xyz)
    kk = np.ascontiguousarray(kept_keys, dtype=np.int64).reshape(-1, 3)
    mask = np.empty(n, dtype=np.uint8)
    check(lib.gsx_density_mask(px, py, pz, stride, n, float(voxel_size), kk.ctypes.data, len(kk), mask.ctypes.data),
          "gsx_density_mask")
    del keep
    return mask.view(np.bool_)


def kmeans_lloyd(data: np.ndarray, init_centroids: np.ndarray, max_iter: int):
    """max_iter x (assign, update) on the GPU (C ABI gsx_kmeans_lloyd) -> (centroids f32[K,D], labels i32[N])."""
    lib = require_hip()
    data = np.ascontiguousarray(data, dtype=np.float32)
    init = np.ascontiguousarray(init_centroids, dtype=np.float32)
    n, d = data.shape
    k = init.shape[0]
    if init.shape[1] != d:
        raise ValueError("init_centroids has the wrong dimensionality")
    cent = np.empty((k, d), dtype=np.float32)
    labels = np.empty(n, dtype=np.int32)
    check(lib.gsx_kmeans_lloyd(data.ctypes.data, n, d, k, int(max_iter), init.ctypes.data, cent.ctypes.data,
                               labels.ctypes.data), "gsx_kmeans_lloyd")
    return cent, labels


def kmeans_pp_trials(k: int) -> int:
    """scikit-learn's n_local_trials (_kmeans_plusplus)"""
    return 2 + int(np.log(k))


def kmeans_pp(data: np.ndarray, k: int, uniforms: np.ndarray, n_local_trials: int | None = None) -> np.ndarray:
    """greedy k-means++ seeding on the GPU (C ABI gsx_kmeans_pp); uniforms: 1 + (k-1) * n_local_trials numbers in [0,1)
    (np.random.random_sample) -> initial centroids f32[k, d]"""
    lib = require_hip()
    data = np.ascontiguousarray(data, dtype=np.float32)
    n, d = data.shape
    L = kmeans_pp_trials(k) if n_local_trials is None else int(n_local_trials)
    u = np.ascontiguousarray(uniforms, dtype=np.float64)
    if u.shape != (1 + (int(k) - 1) * L,):
        raise ValueError("expected 1 + (k-1) * n_local_trials uniform numbers")
    cent = np.empty((int(k), d), dtype=np.float32)
    check(lib.gsx_kmeans_pp(data.ctypes.data, n, d, int(k), u.ctypes.data, L, cent.ctypes.data), "gsx_kmeans_pp")
    return cent


def kmeans1d(vals: np.ndarray, k: int, iters: int = 50, want_labels: bool = False, want_inertia: bool = False):
    """Scalar K-Means codebook (C ABI gsx_kmeans1d: sort + prefix sums + Lloyd on run boundaries from two companded starts).
    -> centroids f32[k] ascending (, labels i32[n]) (, inertia f64[3]: chosen, start A, start B).  Deterministic."""
    lib = require_hip()
    v = np.ascontiguousarray(vals, dtype=np.float32).reshape(-1)
    if not 1 <= int(k) <= 1024:
        raise ValueError("kmeans1d: 1 <= k <= 1024")
    cent = np.empty(int(k), dtype=np.float32)
    labels = np.empty(len(v), dtype=np.int32) if want_labels else None
    inertia = np.zeros(3, dtype=np.float64)
    check(lib.gsx_kmeans1d(v.ctypes.data, len(v), int(k), int(iters), cent.ctypes.data,
                           labels.ctypes.data if want_labels else None, inertia.ctypes.data), "gsx_kmeans1d")
    out = (cent,)
    if want_labels:
        out += (labels,)
    if want_inertia:
        out += (inertia,)
    return out[0] if len(out) == 1 else out


PALETTE_LANES = 8   # concurrent contexts for independent K-Means problems (bench.py config4 at 10M splats: 1 lane 72.8 ms,
                    # 2: 57.1, 4: 54.9, 8: 50.7 -- profiles/r03_variants.txt)


def kmeans_lloyd_many(problems, max_iter: int, lanes: int = PALETTE_LANES, device: int = 0):
    """Independent Lloyd problems -- the SOG palette's chunks, formats/sog.py:536-552 -- on `lanes` contexts with streams of
    their own: while one chunk's ~100 kernels run, the next chunks' rows are uploaded and their kernels fill the tails.
    problems: list of (data f32[n,d], init f32[k,d]); -> list of (centroids f32[k,d], labels i32[n]).  Same kernels and
    launch order per problem as kmeans_lloyd, hence the same results."""
    require_hip()
    if not problems:
        return []
    shapes = {(p[0].shape[1], p[1].shape[0], p[1].shape[1]) for p in problems}
    batched = len(shapes) == 1 and next(iter(shapes))[0] in (9, 24, 45) and next(iter(shapes))[1] >= 64
    if batched and len(problems) > 1 and all(len(p[0]) > 0 for p in problems):
        # the palette's usual case: every chunk the same d and k -> one batched call, the problem is a grid dimension
        # (round 5: gsx_kmeans_lloyd_batch_dev; per problem the same kernels, launch order and arithmetic).  Only the shapes
        # of the matrix-core path batch on the device (d in 9/24/45, k >= 64: the C side's `batched` condition); for the
        # others the batch entry point would loop the problems on ONE stream, so they keep the concurrent lanes below.
        return kmeans_lloyd_batch(problems, max_iter, device)
    lanes = max(1, min(int(lanes), len(problems)))
    ctxs = [Context(device, own_stream=True) for _ in range(lanes)]
    out = [None] * len(problems)
    bufs = []
    try:
        cap_d = max(p[0].size for p in problems) * 4
        cap_c = max(p[1].size for p in problems) * 4
        cap_l = max(len(p[0]) for p in problems) * 4
        bufs = [(c.alloc(cap_d), c.alloc(cap_c), c.alloc(cap_l + 16)) for c in ctxs]
        for base in range(0, len(problems), lanes):
            live = []
            for lane in range(min(lanes, len(problems) - base)):
                data = np.ascontiguousarray(problems[base + lane][0], dtype=np.float32)
                init = np.ascontiguousarray(problems[base + lane][1], dtype=np.float32)
                n, d = data.shape
                k = init.shape[0]
                c = ctxs[lane]
                bd, bc, bl = bufs[lane]
                check(c.lib.gsx_dev_upload_async(c.handle, bd.ptr, data.ctypes.data, data.nbytes), "gsx_dev_upload_async")
                check(c.lib.gsx_dev_upload_async(c.handle, bc.ptr, init.ctypes.data, init.nbytes), "gsx_dev_upload_async")
                check(c.lib.gsx_dev_memset(c.handle, bl.ptr, 0, 4 * n), "gsx_dev_memset")   # max_iter == 0: labels stay 0
                check(c.lib.gsx_kmeans_lloyd_dev(c.handle, bd.ptr, n, d, k, int(max_iter), bc.ptr, bl.ptr), "gsx_kmeans_lloyd_dev")
                live.append((lane, data, init, n, d, k))   # (keeps the host arrays alive until the lane has been drained)
            for lane, _, _, n, d, k in live:
                bd, bc, bl = bufs[lane]
                out[base + lane] = (bc.download(np.float32, k * d).reshape(k, d), bl.download(np.int32, n))
    finally:
        for trio in bufs:
            for b in trio:
                b.free()
        for c in ctxs:
            c.close()
    return out


def kmeans_lloyd_batch(problems, max_iter: int, device: int = 0):
    """problems of ONE shape (same d, same k): rows concatenated in HBM, gsx_kmeans_lloyd_batch_dev runs every iteration of
    all of them as one set of launches -> list of (centroids f32[k,d], labels i32[n_p]) like kmeans_lloyd_many"""
    require_hip()
    d, k = problems[0][0].shape[1], problems[0][1].shape[0]
    rows = [len(p[0]) for p in problems]
    off = np.zeros(len(problems) + 1, dtype=np.int64)
    np.cumsum(rows, out=off[1:])
    n_total = int(off[-1])
    ctx = Context(device)
    bd = bc = bl = None
    try:
        bd, bc, bl = ctx.alloc(4 * n_total * d), ctx.alloc(4 * len(problems) * k * d), ctx.alloc(4 * n_total + 16)
        keep = []
        for i, (data, init) in enumerate(problems):   # chunk by chunk: no second host copy of the whole table
            a = np.ascontiguousarray(data, dtype=np.float32)
            b = np.ascontiguousarray(init, dtype=np.float32)
            keep.append((a, b))
            check(ctx.lib.gsx_dev_upload_async(ctx.handle, bd.ptr + 4 * d * int(off[i]), a.ctypes.data, a.nbytes), "gsx_dev_upload_async")
            check(ctx.lib.gsx_dev_upload_async(ctx.handle, bc.ptr + 4 * k * d * i, b.ctypes.data, b.nbytes), "gsx_dev_upload_async")
        check(ctx.lib.gsx_dev_memset(ctx.handle, bl.ptr, 0, 4 * n_total), "gsx_dev_memset")   # max_iter == 0: labels stay 0
        check(ctx.lib.gsx_kmeans_lloyd_batch_dev(ctx.handle, bd.ptr, off.ctypes.data, len(problems), d, k, int(max_iter), bc.ptr, bl.ptr),
              "gsx_kmeans_lloyd_batch_dev")
        cent = bc.download(np.float32, len(problems) * k * d).reshape(len(problems), k, d)
        lab = bl.download(np.int32, n_total)
        del keep
        return [(cent[i].copy(), lab[int(off[i]):int(off[i + 1])].copy()) for i in range(len(problems))]
    finally:
        for b in (bd, bc, bl):
            if b is not None:
                b.free()
        ctx.close()


def quantize_sorted_codebook(vals: np.ndarray, codebook: np.ndarray) -> np.ndarray:
    lib = require_hip()
    vals = np.ascontiguousarray(vals, dtype=np.float32)
    cb = np.ascontiguousarray(codebook, dtype=np.float32)
    if len(cb) > 256:
        raise ValueError("codebook indices are uint8: at most 256 entries")
    out = np.empty(vals.shape, dtype=np.uint8)
    check(lib.gsx_quantize_sorted_codebook(vals.ctypes.data, vals.size, cb.ctypes.data, len(cb), out.ctypes.data),
          "gsx_quantize_sorted_codebook")
    return out


def lexsort3(k0: np.ndarray, k1: np.ndarray, k2: np.ndarray) -> np.ndarray:
    """np.lexsort((k0, k1, k2)) for float32 keys on the GPU (C ABI gsx_lexsort3) -> int64 indices"""
    lib = require_hip()
    cols = [np.ascontiguousarray(c, dtype=np.float32) for c in (k0, k1, k2)]
    n = len(cols[0])
    out = np.empty(n, dtype=np.uint32)
    if n:
        check(lib.gsx_lexsort3(cols[0].ctypes.data, cols[1].ctypes.data, cols[2].ctypes.data, n, out.ctypes.data), "gsx_lexsort3")
    return out.astype(np.int64)


def sog_quats(rot_rows: np.ndarray) -> np.ndarray:
    """formats/sog.py:315-386 on the GPU (C ABI gsx_sog_quats): (n,4) float32 -> (n,4) uint8 (c0, c1, c2, 252 + argmax)"""
    lib = require_hip()
    q = np.ascontiguousarray(rot_rows, dtype=np.float32)
    if q.ndim != 2 or q.shape[1] != 4:
        raise ValueError("expected (n,4) quaternion rows")
    out = np.empty((len(q), 4), dtype=np.uint8)
    if len(q):
        check(lib.gsx_sog_quats(q.ctypes.data, len(q), out.ctypes.data), "gsx_sog_quats")
    return out


def _log_transform(v):
    return np.sign(v) * np.log(np.abs(v) + 1.0)          # formats/sog.py:280-281, numpy's own float32 arithmetic


def sog_positions(v: np.ndarray, stats: dict | None = None):
    """formats/sog.py:279-309 for one axis -> (u16[n], np.float32 min, np.float32 max of the log-transformed axis), byte-identical
    to the reference's numpy expression.  The transcendental runs on the GPU (C ABI gsx_sog_positions, rounding certificate);
    numpy evaluates (i) the few values next to the extremes of v -- the transform is monotone, a relative window of 1e-3
    around min / max of v holds every candidate for np.min / np.max of the transformed axis -- and (ii) the flagged texels."""
    lib = require_hip()
    v = np.ascontiguousarray(v, dtype=np.float32)
    n = len(v)
    if n == 0:
        return np.zeros(0, np.uint16), np.float32(0), np.float32(0)
    vmin, vmax = np.min(v), np.max(v)
    span = np.float32(1e-3) * np.maximum(np.abs(vmin), np.abs(vmax)) + np.float32(1e-30)
    lo_c, hi_c = v[v <= vmin + span], v[v >= vmax - span]
    if not (np.isfinite(vmin) and np.isfinite(vmax)) or len(lo_c) + len(hi_c) > max(4096, n // 8):
        t = _log_transform(v)                              # degenerate input: the reference's expression as is
        mn, mx = np.min(t), np.max(t)
    else:
        mn, mx = np.min(_log_transform(lo_c)), np.max(_log_transform(hi_c))
    out = np.empty(n, dtype=np.uint16)
    unc = np.empty(n, dtype=np.uint8)
    check(lib.gsx_sog_positions(v.ctypes.data, n, float(mn), float(mx), out.ctypes.data, unc.ctypes.data), "gsx_sog_positions")
    idx = np.flatnonzero(unc)
    if len(idx):
        with np.errstate(all="ignore"):
            t = (_log_transform(v[idx]) - mn) / (mx - mn)
            out[idx] = np.clip(t * 65535, 0, 65535).astype(np.uint16)
    if stats is not None:
        stats["uncertain"] = stats.get("uncertain", 0) + int(len(idx))
        stats["n"] = stats.get("n", 0) + n
    return out, mn, mx


def sog_alpha(opacity: np.ndarray, stats: dict | None = None) -> np.ndarray:
    """formats/sog.py:457-459 -> u8[n], byte-identical to numpy (C ABI gsx_sog_alpha + numpy for the flagged texels)"""
    lib = require_hip()
    o = np.ascontiguousarray(opacity, dtype=np.float32)
    n = len(o)
    out = np.empty(n, dtype=np.uint8)
    if n == 0:
        return out
    unc = np.empty(n, dtype=np.uint8)
    check(lib.gsx_sog_alpha(o.ctypes.data, n, out.ctypes.data, unc.ctypes.data), "gsx_sog_alpha")
    idx = np.flatnonzero(unc)
    if len(idx):
        with np.errstate(all="ignore"):
            out[idx] = np.clip(1.0 / (1.0 + np.exp(-o[idx])) * 255, 0, 255).astype(np.uint8)
    if stats is not None:
        stats["uncertain"] = stats.get("uncertain", 0) + int(len(idx))
        stats["n"] = stats.get("n", 0) + n
    return out


def morton_order(x: np.ndarray, y: np.ndarray, z: np.ndarray, ctx: "Context | None" = None, keep_device: bool = False):
    """formats/compressed_ply.py:245-291 on the GPU (C ABI gsx_morton_order_dev) -> (uint32 order, recursion levels).
    Stable inside runs of equal Morton code.  keep_device: also return the device copy of the order (caller frees)."""
    require_hip()
    cols = [np.ascontiguousarray(c, dtype=np.float32) for c in (x, y, z)]
    n = len(cols[0])
    own = ctx is None
    ctx = ctx or Context(0)
    bufs = []
    try:
        bufs = [ctx.alloc(max(c.nbytes, 16)).upload(c) for c in cols]
        order = ctx.alloc(max(4 * n, 16))
        levels = C.c_int()
        check(ctx.lib.gsx_morton_order_dev(ctx.handle, bufs[0].ptr, bufs[1].ptr, bufs[2].ptr, 1, n, order.ptr, C.byref(levels)),
              "gsx_morton_order_dev")
        host = order.download(np.uint32, n) if n else np.zeros(0, np.uint32)
        if keep_device:
            return host, int(levels.value), order
        order.free()
        return host, int(levels.value)
    finally:
        for b in bufs:
            b.free()
        if own:
            ctx.close()


CPLY_COLUMNS = ("x", "y", "z", "scale_0", "scale_1", "scale_2", "f_dc_0", "f_dc_1", "f_dc_2", "alpha", "rot_0", "rot_1", "rot_2", "rot_3")


def cply_pack(columns: dict, order: "np.ndarray | None", sh_columns=(), ctx: "Context | None" = None):
    """formats/compressed_ply.py:205-241 on the GPU (C ABI gsx_cply_pack_dev / gsx_cply_sh_dev).
    columns: the 14 float32 columns named in CPLY_COLUMNS (original table order; 'alpha' = numpy's sigmoid of the opacity);
    order: uint32 Morton order or None.  -> (chunks (ceil(n/256), 18) f32, vertices (n, 4) u32, sh (n, m) u8 or None)"""
    require_hip()
    cols = [np.ascontiguousarray(columns[name], dtype=np.float32) for name in CPLY_COLUMNS]
    n = len(cols[0])
    nchunks = (n + 255) // 256
    own = ctx is None
    ctx = ctx or Context(0)
    bufs = []
    try:
        bufs = [ctx.alloc(max(c.nbytes, 16)).upload(c) for c in cols]
        ptrs = (C.c_void_p * 14)(*[b.ptr for b in bufs])
        d_order = None
        if order is not None:
            d_order = ctx.alloc(max(4 * n, 16)).upload(np.ascontiguousarray(order, dtype=np.uint32))
            bufs.append(d_order)
        d_chunk, d_vert = ctx.alloc(max(72 * nchunks, 16)), ctx.alloc(max(16 * n, 16))
        bufs += [d_chunk, d_vert]
        check(ctx.lib.gsx_cply_pack_dev(ctx.handle, ptrs, d_order.ptr if d_order else None, n, d_chunk.ptr, d_vert.ptr), "gsx_cply_pack_dev")
        chunks = d_chunk.download(np.float32, 18 * nchunks).reshape(nchunks, 18)
        verts = d_vert.download(np.uint32, 4 * n).reshape(n, 4)
        sh = None
        m = len(sh_columns)
        if m:
            flat = np.empty((m, n), dtype=np.float32)
            for i, c in enumerate(sh_columns):
                flat[i] = c
            d_sh, d_out = ctx.alloc(max(flat.nbytes, 16)).upload(flat), ctx.alloc(max(n * m, 16))
            bufs += [d_sh, d_out]
            check(ctx.lib.gsx_cply_sh_dev(ctx.handle, d_sh.ptr, m, n, d_order.ptr if d_order else None, n, d_out.ptr), "gsx_cply_sh_dev")
            sh = d_out.download(np.uint8, n * m).reshape(n, m)
        return chunks, verts, sh
    finally:
        for b in bufs:
            b.free()
        if own:
            ctx.close()


def prefault(*arrays):
    """touch every page of freshly allocated result arrays on a helper thread WHILE the device works (numpy's strided fill releases
    the GIL; the caller sits in ctypes calls anyway): the download that follows then writes into mapped pages -- 610 MB of texels
    / vertices arrive at 28-34 GB/s through the staging lanes when every 8 MiB chunk first has to fault its 2048 pages in, at link
    rate when the pages are there.  -> join() handle"""
    import threading

    def run():
        for a in arrays:
            flat = a.reshape(-1).view(np.uint8)
            flat[::4096] = 0
    th = threading.Thread(target=run, name="gsx-prefault", daemon=True)
    th.start()
    return th


def file_backed(a) -> bool:
    """is this array a view of a memory-mapped FILE (np.memmap, np.frombuffer over an mmap)?  Pinning such pages in place would
    either fail (read-only mapping) or, for a copy-on-write mapping, make the kernel copy every page first"""
    import mmap
    seen = 0
    while a is not None and seen < 16:
        if isinstance(a, (np.memmap, mmap.mmap)):
            return True
        a = a.base if isinstance(a, np.ndarray) else (a.obj if isinstance(a, memoryview) else None)
        seen += 1
    return False


def upload_table(lib, ctx, dev_ptr: int, rows: np.ndarray):
    """the writers' one upload of a whole table: anonymous memory goes through gsx_dev_upload_staged (pages pinned in place ahead of
    the DMA), a view of a mapped file through the runtime's plain copy"""
    if file_backed(rows):
        check(lib.gsx_dev_upload(ctx.handle, dev_ptr, rows.ctypes.data, rows.nbytes), "gsx_dev_upload")
    else:
        check(lib.gsx_dev_upload_staged(ctx.handle, dev_ptr, rows.ctypes.data, rows.nbytes), "gsx_dev_upload_staged")


def cply_pack_table(data: np.ndarray, sh_names, order: "np.ndarray | None" = None, ctx: "Context | None" = None, stage_ms: "dict | None" = None):
    """The compressed-PLY writer's numeric core on a whole splat table (formats/compressed_ply.py:200-297).  Round 6: the raw rows
    are uploaded ONCE (gsx_dev_upload_staged) and the Morton sort, the chunk packers and the SH packer read their fields straight
    out of them (element stride = row_bytes / 4: gsx_cply_pack_strided_dev / gsx_cply_sh_strided_dev) -- round 5 gathered the 59
    columns on the host (one threaded pass, 472 MB of freshly faulted pages per 2M splats) and uploaded that.  The alpha byte comes
    from the opacity field on the device (float64 exp + the rounding certificate of the SOG textures; numpy's own sigmoid only for
    the ~1e-4 listed splats: gsx_cply_pack_opacity_dev); results come back through the staging lanes.  Tables the row path does not take (fields that are
    not float32, rows that are not a multiple of 4 bytes, f_rest fields that are not consecutive) and calls with a caller's
    context gather their columns on the host as before.
    -> (chunks (ceil(n/256), 18) f32, vertices (n, 4) u32, sh (n, m) u8 or None, order u32[n], recursion levels or None)"""
    lib = require_hip()
    n = len(data)
    fields = data.dtype.fields or {}
    # sh_names: the list itself, or a function (highest f_rest index holding a non-zero | None) -> list: the writer's degree
    # detection (compressed_ply.py:139-171), which on resident rows takes the index from ONE device pass (gsx_fields_nonzero_dev)
    # instead of numpy's `np.any(data[f] != 0)` per strided column (~0.1 s each at 10M splats: more than this whole function)
    resolve = sh_names if callable(sh_names) else None
    present = [i for i in range(45) if "f_rest_%d" % i in fields]
    sh_names = ["f_rest_%d" % i for i in present] if resolve else list(sh_names)
    m = len(sh_names)
    base_names = ["opacity" if c == "alpha" else c for c in CPLY_COLUMNS]
    names = base_names + sh_names
    # (rows that are not a multiple of 4 bytes -- three u1 colour fields behind the floats: 251 -- are copied to 252-byte rows ON the
    #  device after the upload: gsx_rows_repack_dev, ~2 ms per 10M rows)
    resident = (ctx is None and n >= 1024 and data.ndim == 1 and data.flags.c_contiguous
                and all(nm in fields and fields[nm][0] == np.dtype("<f4") and fields[nm][1] % 4 == 0 for nm in names)
                and all(fields[sh_names[i]][1] == fields[sh_names[0]][1] + 4 * i for i in range(m)))
    if resolve and not resident:
        sh_names = list(resolve(None))          # the host's own scan
        m = len(sh_names)
        names = base_names + sh_names
    nchunks = (n + 255) // 256
    own = ctx is None and not resident
    ar = arena(0) if resident else None
    ctx = ar.ctx if resident else (ctx or Context(0))
    bufs = []
    import time as _time
    _t = [_time.perf_counter()]

    def mark(name):   # stage clock (a synchronisation per stage) when the caller asks for it
        if stage_ms is not None:
            ctx.synchronize()
            now = _time.perf_counter()
            stage_ms[name] = round(stage_ms.get(name, 0.0) + (now - _t[0]) * 1e3, 3)
            _t[0] = now

    def alloc(nbytes, name):
        if ar is not None:
            return ar.buf("cply_" + name, nbytes)
        b = ctx.alloc(max(int(nbytes), 16))
        bufs.append(b)
        return b
    try:
        verts = np.empty((n, 4), np.uint32)
        sh = np.empty((n, m), np.uint8) if m and not (resolve and resident) else None
        toucher = prefault(*(a for a in (verts, sh) if a is not None)) if resident and n >= (1 << 18) else None
        toucher2 = None
        if resident:
            pitch = (data.dtype.itemsize + 3) & ~3
            rd = pitch // 4
            d_rows = alloc(data.nbytes + 16, "rows")
            upload_table(lib, ctx, d_rows.ptr, data)
            mark("upload")
            if pitch != data.dtype.itemsize:
                d_raw, d_rows = d_rows, alloc(n * pitch, "rows_aligned")
                check(lib.gsx_rows_repack_dev(ctx.handle, d_raw.ptr, data.dtype.itemsize, n, d_rows.ptr, pitch), "gsx_rows_repack_dev")
                mark("repack")
            if resolve:
                word = C.c_uint64(0)
                if m:
                    check(lib.gsx_fields_nonzero_dev(ctx.handle, d_rows.ptr + int(fields[sh_names[0]][1]), rd, n, m, C.byref(word)),
                          "gsx_fields_nonzero_dev")
                active = [present[j] for j in range(m) if (word.value >> j) & 1]
                sh_names = list(resolve(max(active) if active else -1))
                m = len(sh_names)
                names = base_names + sh_names
                sh = np.empty((n, m), np.uint8) if m else None
                if sh is not None and toucher is not None:
                    toucher2 = prefault(sh)
                mark("sh_detect")
            # (round 6: the sigmoid's byte is decided on the device from the opacity field, with a rounding certificate; numpy only
            #  evaluates the listed ~1e-4 of the splats, below)
            col = lambda i: d_rows.ptr + int(fields[names[i]][1])
            strides = (_I64 * 14)(*([rd] * 14))
            xyz_stride, sh_ptr, sh_col_stride, sh_elem_stride = rd, (d_rows.ptr + int(fields[sh_names[0]][1])) if m else None, 1, rd
        else:
            mat = host_gather_columns(data, names)                     # (14 + m, n), row 9 = opacity
            with np.errstate(over="ignore"):
                mat[9] = 1.0 / (1.0 + np.exp(-mat[9]))                 # :200-203 (float32 in, float32 out, as in the reference)
            d_mat = alloc(max(mat.nbytes, 16), "mat")
            d_mat.upload(mat)
            col = lambda i: d_mat.ptr + 4 * n * i
            strides = (_I64 * 14)(*([1] * 14))
            xyz_stride, sh_ptr, sh_col_stride, sh_elem_stride = 1, col(14), n, 1
        levels = None
        d_order = alloc(4 * n, "order")
        if order is None:
            lv = C.c_int()
            check(lib.gsx_morton_order_dev(ctx.handle, col(0), col(1), col(2), xyz_stride, n, d_order.ptr, C.byref(lv)), "gsx_morton_order_dev")
            levels = int(lv.value)
            mark("morton")
            order = d_order.download(np.uint32, n) if n else np.zeros(0, np.uint32)
        else:
            order = np.ascontiguousarray(order, dtype=np.uint32)
            d_order.upload(order)
        ptrs = (C.c_void_p * 14)(*[col(i) for i in range(14)])
        d_chunk, d_vert = alloc(72 * nchunks, "chunk"), alloc(16 * n, "vert")
        unc_cap = n // 64 + 4096
        d_list, d_cnt = (alloc(8 * unc_cap, "unc"), alloc(16, "unc_count")) if resident else (None, None)
        check(lib.gsx_cply_pack_opacity_dev(ctx.handle, ptrs, strides, d_order.ptr, n, d_chunk.ptr, d_vert.ptr, d_list.ptr if resident else None,
                                            unc_cap if resident else 0, d_cnt.ptr if resident else None), "gsx_cply_pack_opacity_dev")
        d_out = None
        if m:
            d_out = alloc(n * m, "sh")
            check(lib.gsx_cply_sh_strided_dev(ctx.handle, sh_ptr, m, sh_col_stride, sh_elem_stride, d_order.ptr, n, d_out.ptr), "gsx_cply_sh_strided_dev")
        mark("pack")
        chunks = d_chunk.download(np.float32, 18 * nchunks).reshape(nchunks, 18)
        for th in (toucher, toucher2):
            if th is not None:
                th.join()
        dl = lib.gsx_dev_download_staged if resident else lib.gsx_dev_download
        if n:
            check(dl(ctx.handle, verts.ctypes.data, d_vert.ptr, verts.nbytes), "gsx_dev_download")
            if m:
                check(dl(ctx.handle, sh.ctypes.data, d_out.ptr, sh.nbytes), "gsx_dev_download")
        mark("download")
        if resident:
            m_unc = int(d_cnt.download(np.uint32, 1)[0])
            if m_unc > unc_cap:      # (thousands of opacities beyond +-80, or NaNs: numpy's expression for the whole column)
                pos = np.arange(n, dtype=np.int64)
                x = np.ascontiguousarray(data["opacity"])[order]
            else:
                lst = d_list.download(np.uint32, 2 * m_unc).reshape(m_unc, 2) if m_unc else np.zeros((0, 2), np.uint32)
                pos, x = lst[:, 0].astype(np.int64), lst[:, 1].copy().view(np.float32)
            if len(pos):
                with np.errstate(all="ignore"):
                    a = 1.0 / (1.0 + np.exp(-x))                                               # :200-203
                    byte = np.clip(np.floor(a * 255 + 0.5), 0, 255).astype(np.uint32)       # :312
                verts[pos, 3] = (verts[pos, 3] & np.uint32(0xffffff00)) | byte
            mark("host_patch")
        return chunks, verts, sh, order, levels
    except GsxError:
        if ar is not None:
            release_arenas()
        raise
    finally:
        for b in bufs:
            b.free()
        if own:
            ctx.close()


class DeviceArray:
    """A raw HBM allocation owned by a Context (only used where no other allocator is around)."""

    def __init__(self, ctx: "Context", nbytes: int):
        self.ctx = ctx
        self.nbytes = int(nbytes)
        p = C.c_void_p()
        check(ctx.lib.gsx_dev_malloc(ctx.handle, self.nbytes, C.byref(p)), "gsx_dev_malloc")
        self.ptr = p.value

    def upload(self, arr: np.ndarray):
        a = np.ascontiguousarray(arr)
        assert a.nbytes <= self.nbytes
        check(self.ctx.lib.gsx_dev_upload(self.ctx.handle, self.ptr, a.ctypes.data, a.nbytes), "gsx_dev_upload")
        return self

    def download(self, dtype, count: int) -> np.ndarray:
        out = np.empty(count, dtype=dtype)
        assert out.nbytes <= self.nbytes
        check(self.ctx.lib.gsx_dev_download(self.ctx.handle, out.ctypes.data, self.ptr, out.nbytes), "gsx_dev_download")
        return out

    def download_at(self, byte_offset: int, dtype, count: int) -> np.ndarray:
        """`count` elements starting `byte_offset` bytes into the buffer"""
        out = np.empty(count, dtype=dtype)
        assert 0 <= byte_offset and byte_offset + out.nbytes <= self.nbytes
        check(self.ctx.lib.gsx_dev_download(self.ctx.handle, out.ctypes.data, self.ptr + byte_offset, out.nbytes), "gsx_dev_download")
        return out

    def free(self):
        if self.ptr:
            self.ctx.lib.gsx_dev_free(self.ctx.handle, self.ptr)
            self.ptr = None


class Context:
    """gsx_ctx wrapper: device-resident entry points (pointers are plain ints)."""

    def __init__(self, device: int = 0, stream: int | None = None, own_stream: bool = False):
        self.lib = require_hip()
        h = C.c_void_p()
        check(self.lib.gsx_ctx_create(int(device), C.byref(h)), "gsx_ctx_create")
        self.handle = h
        if stream is not None:
            self.set_stream(stream)
        elif own_stream:   # a non-blocking stream of its own: contexts created like this run concurrently on one GPU
            check(self.lib.gsx_ctx_own_stream(self.handle), "gsx_ctx_own_stream")

    def close(self):
        if self.handle:
            self.lib.gsx_ctx_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_stream(self, stream: int):
        check(self.lib.gsx_ctx_set_stream(self.handle, C.c_void_p(int(stream))), "gsx_ctx_set_stream")

    def synchronize(self):
        check(self.lib.gsx_ctx_synchronize(self.handle), "gsx_ctx_synchronize")

    def check(self):
        """synchronise + raise GsxError if a _dev call met non-finite coordinates since the last check"""
        check(self.lib.gsx_ctx_check(self.handle), "gsx_ctx_check")

    def last_knn_algo(self) -> int:
        """KNN_* of the path this context's last KNN call took (adaptive mode picks GRID or TREE)"""
        return int(self.lib.gsx_ctx_last_knn_algo(self.handle))

    def set_param(self, name: str, value: float):
        check(self.lib.gsx_ctx_set_param(self.handle, name.encode(), float(value)), "gsx_ctx_set_param")

    def set_timing(self, on: bool):
        check(self.lib.gsx_ctx_set_timing(self.handle, 1 if on else 0), "gsx_ctx_set_timing")

    def reset_timing(self):
        check(self.lib.gsx_ctx_reset_timing(self.handle), "gsx_ctx_reset_timing")

    def timing(self, slot: int):
        n = C.c_uint64()
        ms = C.c_double()
        check(self.lib.gsx_ctx_get_timing(self.handle, int(slot), C.byref(n), C.byref(ms)), "gsx_ctx_get_timing")
        return int(n.value), float(ms.value)

    def alloc(self, nbytes: int) -> DeviceArray:
        return DeviceArray(self, nbytes)

    def sor_knn(self, x: int, y: int, z: int, stride: int, n_ref: int, q_begin: int, q_count: int, k: int,
                mean_out: int, algo: int = KNN_AUTO, want_info: bool = False):
        info = SorInfo() if want_info else None
        check(self.lib.gsx_sor_knn_dev(self.handle, x, y, z, stride, n_ref, q_begin, q_count, int(k), int(algo),
                                       mean_out, C.byref(info) if want_info else None), "gsx_sor_knn_dev")
        return info.as_dict() if want_info else None

    def sor_knn_share(self, x: int, y: int, z: int, stride: int, n: int, k: int, share: int, nshares: int,
                      mean_out: int, algo: int = KNN_AUTO, want_info: bool = False):
        info = SorInfo() if want_info else None
        check(self.lib.gsx_sor_knn_share_dev(self.handle, x, y, z, stride, n, int(k), int(algo), int(share),
                                             int(nshares), mean_out, C.byref(info) if want_info else None),
              "gsx_sor_knn_share_dev")
        return info.as_dict() if want_info else None

    def sor_stats(self, mean_dists: int, n: int, threshold_factor: float, stats_out: int):
        check(self.lib.gsx_sor_stats_dev(self.handle, mean_dists, n, float(threshold_factor), stats_out),
              "gsx_sor_stats_dev")

    def sor_mask(self, mean_dists: int, n: int, threshold_ptr: int, mask_out: int):
        check(self.lib.gsx_sor_mask_dev(self.handle, mean_dists, n, threshold_ptr, mask_out), "gsx_sor_mask_dev")


class DeviceArena:
    """A long-lived context + named GROW-ONLY device buffers, one per device and process (``arena(device)``).

    hipMalloc / hipFree are not cheap at the writers' sizes: a call that allocates its ~6 GB of work buffers afresh spends
    anything from 3 to 300 ms in the allocator (measured on MI355X, 20 allocations of a 10M-splat SOG encode: 3 ms when the
    runtime still holds the address ranges of the previous call's frees, 308 ms when it does not), and the first DMA into a
    freshly mapped range runs at half the link rate.  A writer that is called again (a converter working through a directory
    of scenes, bench.py's repetitions) finds its buffers here; ``release_arenas()`` gives the memory back."""

    def __init__(self, device: int = 0):
        self.device = int(device)
        self.ctx = Context(device)
        self._side = None
        self._bufs = {}
        self._pinned = {}
        self._leases = set()

    @property
    def side(self) -> "Context":
        """a second context with a stream of its own (copies that overlap the first one's kernels)"""
        if self._side is None:
            self._side = Context(self.device, own_stream=True)
        return self._side

    def buf(self, name: str, nbytes: int) -> DeviceArray:
        nbytes = max(int(nbytes), 16)
        cur = self._bufs.get(name)
        if cur is None or cur.nbytes < nbytes:
            if cur is not None:
                cur.free()
            cur = self.ctx.alloc(nbytes + 256)
            self._bufs[name] = cur
        return cur

    def pinned(self, name: str, nbytes: int) -> np.ndarray:
        """a named grow-only PAGE-LOCKED host buffer as a uint8 array (hipHostMalloc: the DMA engines read it at link rate)"""
        nbytes = max(int(nbytes), 16)
        cur = self._pinned.get(name)
        if cur is None or cur[1] < nbytes:
            if cur is not None:
                self.ctx.lib.gsx_host_pinned_free(self.ctx.handle, C.c_void_p(cur[0]))
                del self._pinned[name]
            p = C.c_void_p()
            check(self.ctx.lib.gsx_host_pinned_alloc(self.ctx.handle, nbytes, C.byref(p)), "gsx_host_pinned_alloc")
            cur = self._pinned[name] = (p.value, nbytes)
        return np.ctypeslib.as_array((C.c_uint8 * cur[1]).from_address(cur[0]))

    def lease(self, who: str) -> bool:
        """one user of the buffers named after `who` at a time (a second DeviceChain alive at the same moment gets a context and
        buffers of its own)"""
        if who in self._leases:
            return False
        self._leases.add(who)
        return True

    def unlease(self, who: str):
        self._leases.discard(who)

    def held_bytes(self) -> int:
        return sum(b.nbytes for b in self._bufs.values())

    def held_pinned_bytes(self) -> int:
        return sum(nb for _, nb in self._pinned.values())

    def release(self):
        for b in self._bufs.values():
            b.free()
        self._bufs.clear()
        for ptr, _ in self._pinned.values():
            if self.ctx is not None and self.ctx.handle:
                self.ctx.lib.gsx_host_pinned_free(self.ctx.handle, C.c_void_p(ptr))
        self._pinned.clear()
        for c in (self._side, self.ctx):
            if c is not None:
                c.close()
        self._side = self.ctx = None


_arenas = {}


def arena(device: int = 0) -> DeviceArena:
    a = _arenas.get(int(device))
    if a is None or a.ctx is None:
        a = _arenas[int(device)] = DeviceArena(device)
    return a


def release_arenas():
    """free every cached work buffer and context of the writers' arenas (they are re-created on the next call)"""
    for dev, a in list(_arenas.items()):
        if a._leases:            # a device chain is alive on this arena's context (a lazy DataProcessor with pending filters): it keeps it
            continue
        if a.ctx is not None:
            a.release()
        del _arenas[dev]


class DeviceChain:
    """Coordinates resident in HBM across consecutive filters (device half).

    The reference ends every filter with ``self.data = vertices[mask]`` on the host table (data_processor.py:114,149)
    and re-gathers ``coords`` at the start of the next one (:38,139).  Here the (n,3) rows are uploaded once, every
    filter leaves its mask in HBM, ``gsx_compact_rows_dev`` compacts the rows there, and the composed survivor list comes
    back once -- the 248-byte host rows are compacted once, after the last filter."""

    def __init__(self, xyz_rows: "np.ndarray | None" = None, device: int = 0, keep_pristine: bool = False, table: "np.ndarray | None" = None):
        """xyz_rows: (n, 3) coordinates, or table: the structured splat table itself -- its x, y, z are then gathered (threaded, C ABI
        gsx_host_gather_f32) straight into a page-locked staging buffer of the arena: no 12n-byte temporary whose pages are faulted
        in, copied from at pageable rate and unmapped again (9 + 5 + 9 ms of configs.dropin_e2e_10m's 37)"""
        if table is not None:
            require_hip()
            ar0 = arena(device)
            n_t = len(table)
            if n_t >= 4096 and "chain" not in ar0._leases:
                a = host_gather_xyz(table, out=ar0.pinned("chain_xyz", 12 * n_t)[:12 * n_t].view(np.float32).reshape(n_t, 3))
            else:
                a = host_gather_xyz(table)
        else:
            a = np.ascontiguousarray(xyz_rows, dtype=np.float32)
        if a.ndim != 2 or a.shape[1] != 3:
            raise ValueError("Requires 3D data")
        # round 6: the process-wide arena's context and named buffers when no other chain holds them -- a chain per filter run used
        # to create a context, hipMalloc its ~37 bytes per row and, inside the context, the whole KNN workspace, and free it all
        # again at close(): 10-12 of the 37 ms of configs.dropin_e2e_10m.  release_arenas() gives the memory back.
        require_hip()
        ar = arena(device)
        self._ar = ar if ar.lease("chain") else None
        self.ctx = None
        self.rows = self.spare = self.orig = self.mask = self.pristine = self._md = self._st = None
        self._pool = []
        try:
            self._setup(a, device, keep_pristine, table is not None)
        except BaseException:
            self.close()                    # (gives the lease back; a private context and its buffers are freed)
            raise

    def _setup(self, a, device, keep_pristine, pinned_source):
        ar = self._ar
        self.ctx = ar.ctx if self._ar is not None else Context(device)
        self._names = iter(range(1 << 30))
        self.ctx.set_param("adaptive", 1)   # every step below synchronises anyway
        self.n0 = self.n = int(a.shape[0])
        self.rows = self._alloc(max(a.nbytes, 16), "rows")
        if a.nbytes >= (32 << 20) and not pinned_source:      # through the pinned staging lanes: a caller's array is pageable memory
            check(self.ctx.lib.gsx_dev_upload_staged(self.ctx.handle, self.rows.ptr, a.ctypes.data, a.nbytes), "gsx_dev_upload_staged")
        else:
            self.rows.upload(a)
        self.spare = self._alloc(max(a.nbytes, 16), "spare")
        # bench.py only: a second device copy of the uploaded rows, so that restart() can run the chain again without PCIe
        self.pristine = self._alloc(max(a.nbytes, 16), "pristine") if keep_pristine else None
        if self.pristine is not None:
            check(self.ctx.lib.gsx_dev_copy(self.ctx.handle, self.pristine.ptr, self.rows.ptr, a.nbytes), "gsx_dev_copy")
        self.orig = None                    # None = identity
        self._pool = []                     # survivor-list buffers not in use (4 n0 bytes each; at most two ever exist)
        self.empty = False                  # keep_none(): no survivor, whatever self.orig says
        self.mask = self._alloc(self.n0 + 16, "mask")
        self._md = self._st = None          # SOR work buffers (mean distances, statistics), allocated on first use
        self._pristine_box = None           # box of the uploaded rows, when it was measured before any filter ran (restart() keeps it)
        self._box = None                    # box of the rows at the first density_filter() (a superset of every LATER state of the
                                            # chain; restart() drops it).  gsx_density_filter_dev re-derives the frame itself when
                                            # a row falls outside the box it was given (status word `oob` of the device result)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _alloc(self, nbytes: int, name: str):
        """a named grow-only buffer of the arena while this chain holds its lease, an allocation of its own otherwise"""
        if self._ar is not None:
            return self._ar.buf("chain_" + name, nbytes)
        return self.ctx.alloc(nbytes)

    def _free(self, b):
        if b is not None and self._ar is None:
            b.free()

    def restart(self):
        """back to the state right after the upload (needs keep_pristine): device-to-device copy, identity survivor list"""
        if self.pristine is None:
            raise ValueError("DeviceChain(keep_pristine=True) is needed for restart()")
        check(self.ctx.lib.gsx_dev_copy(self.ctx.handle, self.rows.ptr, self.pristine.ptr, 12 * self.n0), "gsx_dev_copy")
        if self.orig is not None:
            self._pool.append(self.orig)
        self.orig = None
        self.n, self.empty = self.n0, False
        # the cached box is the box of the rows that were current at the first density_filter() -- after a restart the
        # pristine rows may reach beyond it: back to the box of the PRISTINE rows if that is what was
        # measured (no filter had run yet), else to "not measured"
        self._box = self._pristine_box

    def _xyz(self):
        p = self.rows.ptr
        return p, p + 4, p + 8, 3

    def bbox(self):
        """per-axis (min, max) of the surviving rows (gsx_slab_bbox_dev: the multi-GPU path's box kernel) -> ([3], [3]) float32"""
        out = self._alloc(32, "bbox")
        try:
            x, y, z, st = self._xyz()
            check(self.ctx.lib.gsx_slab_bbox_dev(self.ctx.handle, x, y, z, st, self.n, out.ptr), "gsx_slab_bbox_dev")
            b = out.download(np.float32, 7)
        finally:
            self._free(out)
        self.bbox_nonfinite = bool(b[6] != 0)      # a NaN or an infinity among the coordinates (numpy's min / max would propagate a NaN)
        return [-b[0], -b[1], -b[2]], [b[3], b[4], b[5]]

    def density_voxels(self, voxel_size: float, min_points: int):
        n = self.n
        dense_cap = int(min(n, n // max(int(min_points), 1) + 1))
        keys = np.empty((max(dense_cap, 1), 3), dtype=np.int64)
        counts = np.empty(max(dense_cap, 1), dtype=np.int64)
        nu, nd = C.c_int64(), C.c_int64()
        x, y, z, st = self._xyz()
        check(self.ctx.lib.gsx_density_voxels_dev(self.ctx.handle, x, y, z, st, n, float(voxel_size), int(min_points), dense_cap,
                                                  C.byref(nu), C.byref(nd), keys.ctypes.data, counts.ctypes.data), "gsx_density_voxels_dev")
        m = int(nd.value)
        return {"n_unique": int(nu.value), "dense_keys": keys[:m].copy(), "dense_counts": counts[:m].copy()}

    def density_filter(self, voxel_size: float, min_points: int, keep_multicluster: bool):
        """the whole density filter on the device (gsx_density_filter_dev) -> dict(status, n_unique, kept_clusters, largest,
        left: rows after the compaction) ; status DENSITY_HOST: nothing was applied, take the host path"""
        if self._box is None:     # box of the rows as they are now: a superset of whatever survives later filters
            lo, hi = self.bbox()
            self._box = np.array(list(lo) + list(hi), dtype=np.float32)
            if self.orig is None and not self.empty:     # nothing has been removed yet: this IS the box of the uploaded rows
                self._pristine_box = self._box
        info = DensityInfo()
        x, y, z, st = self._xyz()
        check(self.ctx.lib.gsx_density_filter_dev(self.ctx.handle, x, y, z, st, self.n, float(voxel_size), int(min_points),
                                                  1 if keep_multicluster else 0, self._box.ctypes.data, self.mask.ptr, C.byref(info)),
              "gsx_density_filter_dev")
        out = {"status": int(info.status), "n_unique": int(info.n_unique), "kept_clusters": int(info.kept_clusters),
               "largest": int(info.largest), "left": None}
        if info.status == DENSITY_OK:
            out["left"] = self._compact()
        return out

    def _compact(self) -> int:
        out = self._pool.pop() if self._pool else self._alloc(4 * self.n0 + 16, "list%d" % next(self._names))
        n_out = C.c_int64()
        check(self.ctx.lib.gsx_compact_rows_dev(self.ctx.handle, self.rows.ptr, self.orig.ptr if self.orig is not None else None,
                                                self.mask.ptr, self.n, self.spare.ptr, out.ptr, C.byref(n_out)),
              "gsx_compact_rows_dev")
        self.rows, self.spare = self.spare, self.rows
        if self.orig is not None:
            self._pool.append(self.orig)    # the previous survivor list is free again
        self.orig = out
        self.n = int(n_out.value)
        return self.n

    def density_keep(self, voxel_size: float, kept_keys: np.ndarray) -> int:
        kk = np.ascontiguousarray(kept_keys, dtype=np.int64).reshape(-1, 3)
        x, y, z, st = self._xyz()
        check(self.ctx.lib.gsx_density_mask_dev(self.ctx.handle, x, y, z, st, self.n, float(voxel_size), kk.ctypes.data, len(kk),
                                                self.mask.ptr), "gsx_density_mask_dev")
        return self._compact()

    def keep_none(self):
        """a filter removed every row (reference: ``self.data = self.data[:0]``, data_processor.py:54-57,91-93).
        ``orig`` may still be None (= identity) when this is the first filter of the chain, so emptiness is its own flag."""
        self.n = 0
        self.empty = True

    def bbox_keep(self, bounds6) -> int:
        """crop_by_bbox (data_processor.py:215-224): bounds as numpy would see them -- Python floats are weak scalars
        (rounded to float32 before the comparison), numpy float64 scalars promote the comparison to f64"""
        b = np.array([float(np.float32(v)) if type(v) in (float, int) else float(v) for v in bounds6], dtype=np.float64)
        check(self.ctx.lib.gsx_mask_bbox_dev(self.ctx.handle, self.rows.ptr, self.n, b.ctypes.data, self.mask.ptr), "gsx_mask_bbox_dev")
        return self._compact()

    def ge_keep(self, column: np.ndarray, threshold: float) -> int:
        """rows whose value in ``column`` (float32, one per row of the table the chain started from) is >= threshold"""
        col = np.ascontiguousarray(column, dtype=np.float32)
        if col.shape != (self.n0,):
            raise ValueError("column must have one value per original row")
        dev = self._alloc(max(col.nbytes, 16), "column").upload(col)
        try:
            check(self.ctx.lib.gsx_mask_ge_dev(self.ctx.handle, dev.ptr, self.orig.ptr if self.orig is not None else None, self.n,
                                               float(threshold), self.mask.ptr), "gsx_mask_ge_dev")
            return self._compact()
        finally:
            self._free(dev)

    def sor_keep(self, k: int, threshold_factor: float):
        numpy_reduction_selfcheck(self.ctx)
        n = self.n
        if self._md is None:                # sized for the whole table once: no allocation in later calls
            self._md, self._st = self._alloc(4 * self.n0 + 16, "md"), self._alloc(16, "st")
        md, st = self._md, self._st
        x, y, z, stride = self._xyz()
        self.ctx.sor_knn(x, y, z, stride, n, 0, n, int(k), md.ptr)
        self.ctx.sor_stats(md.ptr, n, float(threshold_factor), st.ptr)
        self.ctx.sor_mask(md.ptr, n, st.ptr + 8, self.mask.ptr)
        self.ctx.check()
        stats = st.download(np.float32, 3)
        kept = self._compact()
        return {"mean": stats[0], "std": stats[1], "threshold": stats[2], "kept": kept}

    def survivors(self) -> np.ndarray:
        """indices (ascending) of the surviving rows in the table the chain started from"""
        if self.empty or self.n == 0:
            return np.zeros(0, np.uint32)
        if self.orig is None:
            return np.arange(self.n0, dtype=np.uint32)
        return self.orig.download(np.uint32, self.n) if self.n else np.zeros(0, np.uint32)

    def close(self):
        if self._ar is not None:             # the arena keeps the buffers and the context (and its KNN workspace) for the next chain
            ar, self._ar = self._ar, None
            healthy = False
            if self.ctx is not None and self.ctx.handle:
                try:
                    self.ctx.set_param("adaptive", 0)
                    self.ctx.synchronize()
                    healthy = True
                except GsxError:
                    pass
            ar.unlease("chain")
            self.ctx = None
            if not healthy:                  # a failed HIP call somewhere in this chain: do not hand the context on
                release_arenas()
            return
        if self.ctx is None:
            return
        for b in (self.rows, self.spare, self.orig, self.mask, self.pristine, self._md, self._st, *self._pool):
            if b is not None:
                b.free()
        self.ctx.close()
        self.ctx = None
