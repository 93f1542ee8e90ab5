"""``write_compressed_ply`` -- the reference's ``CompressedPlyFormat.write`` (formats/compressed_ply.py:128-243) with its
numeric core on the MI355X.

  | step (formats/compressed_ply.py)                   | here                                                        |
  |----------------------------------------------------|-------------------------------------------------------------|
  | :245-291 recursive Morton sort                      | ``gsx_morton_order_dev`` (radix sort per recursion level,    |
  |                                                    | every group of a level at once)                              |
  | :139-171 active SH degree                           | numpy on the host, as the reference                          |
  | :200-203 sigmoid of the opacity                     | numpy on the host: numpy's float32 ``exp`` is a SIMD routine |
  |                                                    | a device ``exp`` cannot reproduce bit for bit                |
  | :205-234 chunk bounds + the three packers           | ``gsx_cply_pack_dev``: one workgroup per 256-splat chunk      |
  | :236-241 SH bytes                                   | ``gsx_cply_sh_dev``                                          |
  | :385-390 PLY container                              | the reference's ``_write_ply_file`` when plyfile is there,    |
  |                                                    | else the same three binary elements written directly         |

Bit-exact chunk records and packed words given the same splat order.  The order: the sequence of Morton codes is the
reference's; splats with EQUAL code keep ascending input index here, while ``np.argsort`` (not stable) leaves them in an
order that depends on numpy's build and the CPU (csrc/cply.hip header).
"""
from __future__ import annotations

import numpy as np

from .. import _lib
from ..utils import debug_print

CHUNK_SIZE = 256   # :12

CHUNK_DTYPE = np.dtype([(p + a, "f4") for p, axes in (("min_", "xyz"), ("max_", "xyz")) for a in axes] +
                       [(p + "scale_" + a, "f4") for p in ("min_", "max_") for a in "xyz"] +
                       [(p + a, "f4") for p in ("min_", "max_") for a in "rgb"])            # :167-174
VERTEX_DTYPE = np.dtype([("packed_position", "u4"), ("packed_rotation", "u4"), ("packed_scale", "u4"), ("packed_color", "u4")])


def active_sh_names(data, last_active_idx=None):
    """:139-171 -- the f_rest_* columns up to the highest degree that has a non-zero coefficient.  last_active_idx: the highest
    f_rest index that holds a non-zero, when the caller already knows it (the device's pass over resident rows); None = scan"""
    names = data.dtype.names
    if last_active_idx is not None:
        pass
    elif any(n.startswith("f_rest_") for n in names):
        last_active_idx = -1
        for i in range(44, -1, -1):
            fname = f"f_rest_{i}"
            if fname in names and np.any(data[fname] != 0):
                last_active_idx = i
                break
    else:
        last_active_idx = -1
    if last_active_idx >= 24:
        target_degree = 3
    elif last_active_idx >= 9:
        target_degree = 2
    elif last_active_idx >= 0:
        target_degree = 1
    else:
        target_degree = 0
    needed = {3: 45, 2: 24, 1: 9, 0: 0}[target_degree]
    sh_names = [f"f_rest_{i}" for i in range(needed) if f"f_rest_{i}" in names]
    debug_print(f"[DEBUG] Compressed PLY SH Detection: Max Index={last_active_idx}, Degree={target_degree}, Coeffs Count={len(sh_names)}")
    return sh_names


def encode(data: np.ndarray, order=None):
    """-> (chunk_data, vertex_data, sh_data or None, order): the three structured arrays the reference hands to
    ``_write_ply_file``.  order: a precomputed splat order (e.g. the reference's own) instead of the Morton sort."""
    n = len(data)
    found = []

    def resolve(last_active_idx):      # (called once by cply_pack_table: with the device's answer on resident rows, with None otherwise)
        found[:] = active_sh_names(data, last_active_idx)
        return found
    # round 5: one threaded gather of every column the writer needs, one upload, Morton order + packers + SH bytes on the device
    # (_lib.cply_pack_table; the per-column path below it -- _lib.morton_order + _lib.cply_pack -- is kept for callers with columns)
    chunks, verts, sh, order, levels = _lib.cply_pack_table(data, resolve, order)
    sh_names = list(found)
    if levels is not None:
        debug_print(f"[DEBUG] Morton order: {levels} recursion level(s)")
    chunk_data = np.ascontiguousarray(chunks).view(CHUNK_DTYPE).reshape(-1)
    vertex_data = np.ascontiguousarray(verts).view(VERTEX_DTYPE).reshape(-1)
    sh_data = None
    if sh_names:
        sh_data = np.ascontiguousarray(sh).view(np.dtype([(name, "u1") for name in sh_names])).reshape(-1)
    assert len(vertex_data) == n
    return chunk_data, vertex_data, sh_data, order


_PLY_TYPES = {"f4": "float", "u4": "uint", "u1": "uchar"}


def _write_ply(path, elements):
    """binary little-endian PLY with the given (name, structured array) elements -- the layout plyfile produces for
    ``PlyData(elements, text=False, byte_order='<')`` (:385-390)"""
    with open(path, "wb") as f:
        head = ["ply", "format binary_little_endian 1.0"]
        for name, arr in elements:
            head.append(f"element {name} {len(arr)}")
            for field in arr.dtype.names:
                head.append(f"property {_PLY_TYPES[arr.dtype[field].str[1:]]} {field}")
        head.append("end_header")
        f.write(("\n".join(head) + "\n").encode("ascii"))
        for _, arr in elements:
            f.write(np.ascontiguousarray(arr).astype(arr.dtype.newbyteorder("<"), copy=False).tobytes())


def write_compressed_ply(data: np.ndarray, path: str, **kwargs) -> None:
    debug_print(f"[DEBUG] Writing Compressed PLY file to {path}")
    chunk_data, vertex_data, sh_data, _ = encode(data)
    try:
        from gsconverter.formats.compressed_ply import CompressedPlyFormat   # type: ignore
        from plyfile import PlyElement
        if not hasattr(PlyElement, "describe"):
            raise ImportError("plyfile is a stub")
        CompressedPlyFormat()._write_ply_file(path, chunk_data, vertex_data, sh_data)
    except ImportError:
        elements = [("chunk", chunk_data), ("vertex", vertex_data)] + ([("sh", sh_data)] if sh_data is not None else [])
        _write_ply(path, elements)
    debug_print(f"Compressed PLY write completed. {len(vertex_data)} points in {len(chunk_data)} chunks.")
