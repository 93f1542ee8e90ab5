"""``write_sog`` -- the reference's ``SogFormat.write`` (formats/sog.py:249-639) with its numeric core on the MI355X.

Same bundle layout and the same bytes wherever the reference is deterministic; what runs where:

  | step (formats/sog.py)                         | here                                                        |
  |-----------------------------------------------|-------------------------------------------------------------|
  | :264 spatial order ``np.lexsort((z, y, x))``   | ``gsx_lexsort3`` (three stable radix passes on the GPU)     |
  | :279-309 log-transformed u16 positions        | ``gsx_sog_positions`` / ``gsx_sog_alpha``: float64 log / exp |
  | :457-459 sigmoid of the opacity                | on the GPU + a rounding certificate; numpy (whose float32    |
  |                                               | SIMD log/exp bits a device cannot reproduce) only evaluates  |
  |                                               | the ~1 % of texels next to a rounding boundary: same bytes    |
  | :315-386 quaternion smallest-three packing    | ``gsx_sog_quats``, byte-exact                                |
  | :392-449 scale / colour codebooks + quantiser | ``gpu_ops.kmeans(init="k-means++")`` -> ``gsx_kmeans1d`` (sorted-run |
  |                                               | solver: the quality of the reference's scikit-learn path, not |
  |                                               | of its random-init Taichi path) + ``gsx_quantize_sorted_codebook`` |
  | :496-552 SH palette, 64 independent chunks    | ``dist_palette.palette_kmeans`` (matrix-core assign; one GPU |
  |                                               | or dealt out across the GPUs of a node)                     |
  | :561 256-scalar codebook of the centroids     | ``gsx_kmeans1d`` over the flattened palette (the reference is |
  |                                               | hard-wired to sklearn ``MiniBatchKMeans`` there, unseeded: any |
  |                                               | codebook of at least that quality is admissible), quantised  |
  |                                               | on the GPU                                                   |
  | :269-276, 566-639 WebP textures, meta, zip     | pillow / zipfile, as the reference (out of scope: packaging) |

Round 6: the table is DEVICE-RESIDENT for the whole core (formats/sog_device.py, csrc/sog_table.hip): the raw rows are uploaded
once, lexsort / ``data[indices]`` / texels / codebooks / palette run in HBM, texels come back.  The host-staged core below (one
upload / download per stage) remains for the tables that path does not take (non-float32 fields, fewer than 1024 rows,
non-finite coordinates) and for a palette dealt out across GPUs.

``install(sog_writer=True)`` rebinds ``gsconverter.formats.sog.SogFormat.write`` to this function, which also makes the
GPU quantiser reachable from the reference's CLI (in the reference it is a closure inside ``write``).
"""
from __future__ import annotations

import io
import json
import zipfile

import numpy as np

from .. import _lib
from ..processing import gpu_ops
from ..utils import debug_print, status_print
from .. import dist_palette
from . import sog_device


def _texture_size(n):
    width = int(np.ceil(np.sqrt(n) / 4) * 4)            # :260-261
    height = int(np.ceil(n / width / 4) * 4)
    return width, height


def _webp_bytes(pixels_rgba, w, h):
    from PIL import Image
    bio = io.BytesIO()
    Image.frombytes("RGBA", (w, h), pixels_rgba.tobytes()).save(bio, format="WEBP", lossless=True, quality=100, method=1)   # :273-275
    return bio.getvalue()


def _webp_all(zf, jobs):
    """jobs: (entry name, (texels, 4) uint8, w, h) in the reference's order of write_webp calls.  The lossless encodes are independent
    and libwebp runs outside the GIL: one thread per image (six or seven; the same bytes per image, the same entry order in the
    archive -- at 10M splats the encodes are seconds each and the whole of the writer's time once its numeric core takes 0.1 s)"""
    from concurrent.futures import ThreadPoolExecutor
    if len(jobs) > 1:
        with ThreadPoolExecutor(max_workers=len(jobs), thread_name_prefix="gsx-webp") as ex:
            blobs = list(ex.map(lambda j: _webp_bytes(j[1], j[2], j[3]), jobs))
    else:
        blobs = [_webp_bytes(j[1], j[2], j[3]) for j in jobs]
    for j, blob in zip(jobs, blobs):
        zf.writestr(j[0], blob)


def _cols(ds, names):
    """the named columns as contiguous arrays: ONE threaded pass over the rows for float32 fields (numpy's strided copies of a
    248-byte-stride view cost ~5 ms per column and million rows), the views themselves for any other dtype"""
    fields = ds.dtype.fields or {}
    if len(ds) >= 4096 and all(nm in fields and fields[nm][0] == np.dtype("<f4") for nm in names):
        return list(_lib.host_gather_columns(ds, list(names)))
    return [ds[nm] for nm in names]


def _rows(ds, names):
    """np.column_stack of the named columns -> (n, len(names)): row-major straight out of the threaded gather for float32 fields"""
    fields = ds.dtype.fields or {}
    if len(ds) >= 4096 and all(nm in fields and fields[nm][0] == np.dtype("<f4") for nm in names):
        return _lib.host_gather_xyz(ds, list(names))
    return np.column_stack([ds[nm] for nm in names])


def _positions(ds):
    """:279-309 -- GPU transcendental + certificate, numpy for the texels next to a rounding boundary (_lib.sog_positions)"""
    u16, mins, maxs = [], [], []
    for col in _cols(ds, "xyz"):
        u, mn, mx = _lib.sog_positions(col)
        u16.append(u)
        mins.append(mn)
        maxs.append(mx)
    return u16, mins, maxs


def _scalar_codebook(columns, ds, label):
    """:392-423 / :435-449: 256-entry codebook of the 3N scalars (fit on a 50 000-sample), then nearest-entry indices"""
    cols = _cols(ds, columns)
    flat = np.concatenate(cols)
    status_print(label)
    fit = flat
    if len(flat) > 50000:
        fit = flat[np.random.choice(len(flat), 50000, replace=False)]
    # the reference calls gpu_ops.kmeans(fit.reshape(-1, 1), 256, max_iter=20): random-row init with Taichi, k-means++-seeded
    # sklearn without.  The scalar solver gives the better of the two qualities deterministically.
    cent, _ = gpu_ops.kmeans(fit.reshape(-1, 1), 256, max_iter=20, init="k-means++")
    codebook = np.array(sorted(cent.flatten()))
    idx = [gpu_ops.quantize_to_codebook(np.ascontiguousarray(c), codebook) if len(codebook) > 1
           else np.zeros(len(ds), np.uint8) for c in cols]
    return codebook, idx


def _sh_bands(data, ds):
    """:461-493: bands present in the dtype, downgraded when the trailing coefficients are all zero"""
    names = data.dtype.names
    if "f_rest_0" not in names:
        return 0
    count = sum(1 for i in range(45) if "f_rest_%d" % i in names)
    bands = 3 if count >= 45 else 2 if count >= 24 else 1 if count >= 9 else 0
    if bands > 0:
        last = -1
        for i in range({3: 44, 2: 23, 1: 8}[bands], -1, -1):
            fn = "f_rest_%d" % i
            if fn in names and np.any(ds[fn] != 0):
                last = i
                break
        bands = 3 if last >= 24 else 2 if last >= 9 else 1 if last >= 0 else 0
    debug_print(f"[DEBUG] SOG Write: Effective SH Bands detected: {bands}")
    return bands


def _encode_host(data: np.ndarray, level: int, comm=None, be=None) -> dict:
    """The numeric core with the table on the HOST: one upload / download per stage (round 3's writer).  Takes every table the
    reference takes (any dtype, tiny tables with their k >= N shortcuts, a palette dealt out across GPUs through comm / be);
    returns what sog_device.encode returns."""
    n = len(data)
    width, height = _texture_size(n)
    texels = width * height
    zyx = _lib.host_gather_columns(data, ["z", "y", "x"])                     # one threaded pass instead of three strided numpy copies
    order = _lib.lexsort3(zyx[0], zyx[1], zyx[2])                             # :264
    ds = data[order]
    out = {"n": n, "width": width, "height": height, "textures": {}, "stats": {}}

    # positions: low / high byte textures (:300-312)
    u16, mins, maxs = _positions(ds)
    lo = np.full((texels, 4), 255, np.uint8)
    hi = np.full((texels, 4), 255, np.uint8)
    for c in range(3):
        lo[:n, c] = u16[c] & 0xff
        hi[:n, c] = u16[c] >> 8
    out["textures"]["means_l"], out["textures"]["means_u"] = lo, hi
    out["mins"], out["maxs"] = mins, maxs

    # rotations (:315-386)
    quats = np.full((texels, 4), 255, np.uint8)
    quats[:n] = _lib.sog_quats(_rows(ds, ("rot_0", "rot_1", "rot_2", "rot_3")))
    out["textures"]["quats"] = quats

    # scales (:388-431) and colours + opacity (:433-459)
    scale_cb, (s0, s1, s2) = _scalar_codebook(("scale_0", "scale_1", "scale_2"), ds, "Clustering Scales...")
    scales = np.zeros((texels, 4), np.uint8)
    scales[:n, 0], scales[:n, 1], scales[:n, 2], scales[:n, 3] = s0, s1, s2, 255
    out["textures"]["scales"], out["scale_codebook"] = scales, scale_cb
    color_cb, (d0, d1, d2) = _scalar_codebook(("f_dc_0", "f_dc_1", "f_dc_2"), ds, "Clustering Colors...")
    sh0 = np.zeros((texels, 4), np.uint8)
    sh0[:n, 0], sh0[:n, 1], sh0[:n, 2] = d0, d1, d2
    sh0[:n, 3] = _lib.sog_alpha(_cols(ds, ("opacity",))[0])                                    # :457-459
    out["textures"]["sh0"], out["color_codebook"] = sh0, color_cb

    # SH-N palette (:496-600)
    bands = out["bands"] = _sh_bands(data, ds)
    if bands > 0:
        coeffs = [0, 9, 24, 45][bands]
        sh = _rows(ds, ["f_rest_%d" % i for i in range(coeffs)]).astype(np.float32, copy=False)
        status_print(f"SOG Write Quality Level: {level} (0=Max, 9=Min)")
        status_print(f"SH Clustering: K={dist_palette.palette_plan(n, level)['target_k']}, Points={n}. Strategy: GPU (HIP gfx950)")
        centroids, labels = dist_palette.palette_kmeans(sh, level, 10, comm=comm, be=be)
        palette = out["palette"] = len(centroids)
        status_print("Clustering SH Centroids into Codebook...")
        flat = np.ascontiguousarray(centroids, dtype=np.float32).reshape(-1)
        if len(flat) > 256:                                                                               # :561
            codebook = _lib.kmeans1d(flat, 256, iters=100).astype(np.float64)
        else:   # fewer scalars than codebook entries (sklearn would raise): every scalar is its own entry
            codebook = np.array(sorted(flat.tolist()))
        out["shn_codebook"] = codebook
        out["shn_centroid_index"] = gpu_ops.quantize_to_codebook(centroids.flatten(), codebook)
        limg = np.zeros((texels, 4), np.uint8)
        l16 = labels.astype(np.uint16)
        limg[:n, 0], limg[:n, 1], limg[:n, 2], limg[:n, 3] = l16 & 0xff, l16 >> 8, 0, 255
        out["textures"]["shN_labels"] = limg
        assert palette * coeffs == len(out["shn_centroid_index"])
    return out


DEVICE_RESIDENT = None   # default of encode(device_resident=...): None = whenever eligible (the CPU-only call-sequence test of the
                         # host-staged stages, tests/test_e2e_reference_dropin.py, sets False)


def encode(data: np.ndarray, level: int = 0, comm=None, be=None, device_resident=None, profile: bool = False) -> dict:
    """every array SogFormat.write hands to write_webp + the numbers of its meta.json (sog.py:249-600 without the packaging).
    device_resident: None = the device-resident core (formats/sog_device.py) whenever the table and the request allow it,
    True = insist (NotEligible propagates), False = the host-staged core."""
    if device_resident is None:
        device_resident = DEVICE_RESIDENT
    if device_resident is not False and comm is None:
        try:
            return sog_device.encode(data, level, profile=profile)
        except sog_device.NotEligible as e:
            if device_resident:
                raise
            debug_print(f"[DEBUG] SOG Write: host-staged core ({e})")
    return _encode_host(data, level, comm=comm, be=be)


def write_sog(data: np.ndarray, path: str, comm=None, be=None, **kwargs):
    """data: the reference's structured splat table.  comm / be: optional communicator + buffer backend of dist_slab for
    the SH palette across GPUs (every rank passes the same table; rank 0's file is the result)."""
    n = len(data)
    debug_print(f"[DEBUG] Writing .sog file to {path}")
    try:
        level = int(kwargs.get("compression_level", 0))
    except Exception:
        level = 0
    core = encode(data, level, comm=comm, be=be, device_resident=kwargs.get("device_resident"))
    width, height, tex = core["width"], core["height"], core["textures"]
    zf = zipfile.ZipFile(path, "w", zipfile.ZIP_STORED)
    jobs = [(name + ".webp", tex[name], width, height) for name in ("means_l", "means_u", "quats", "scales", "sh0")]

    shn_meta = None
    bands = core["bands"]
    if bands > 0:
        coeffs = [0, 9, 24, 45][bands]
        palette, codebook, cidx = core["palette"], core["shn_codebook"], core["shn_centroid_index"]
        w_c, h_c = 64 * coeffs, int(np.ceil(palette / 64))
        cimg = np.full((w_c * h_c, 4), 255, np.uint8)
        per = cidx.reshape(palette, 3, coeffs // 3).transpose(0, 2, 1).reshape(-1, 3)                 # (P, 3, C) -> (P*C, 3), :580-590
        cimg[:len(per), :3] = per
        jobs.append(("shN_centroids.webp", cimg, w_c, h_c))
        jobs.append(("shN_labels.webp", tex["shN_labels"], width, height))
        shn_meta = {"count": int(palette), "bands": int(bands), "codebook": [float(c) for c in codebook],
                    "files": ["shN_centroids.webp", "shN_labels.webp"]}
    _webp_all(zf, jobs)

    meta = {"version": 2, "asset": {"generator": "gsconverter-sog"}, "count": n,
            "means": {"mins": [float(m) for m in core["mins"]], "maxs": [float(m) for m in core["maxs"]], "files": ["means_l.webp", "means_u.webp"]},
            "scales": {"codebook": [float(c) for c in core["scale_codebook"]], "files": ["scales.webp"]},
            "quats": {"files": ["quats.webp"]},
            "sh0": {"codebook": [float(c) for c in core["color_codebook"]], "files": ["sh0.webp"]}}
    if shn_meta:
        meta["shN"] = shn_meta
    zf.writestr("meta.json", json.dumps(meta))
    zf.close()
    status_print(f"SOG write completed to {path}. {n} points bundled.")
