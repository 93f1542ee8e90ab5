"""SOG writer numeric core -- numpy restatement of the reference's lines (formats/sog.py:264-386, 457-459).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Pinned by the textures of bundles the reference itself wrote
(tests/golden/kmeans_ref.npz: ``sog_*__means_l / means_u / quats / sh0[...,3]``, oracle/make_golden_kmeans.py).
The code in ``SogFormat.write`` is straight-line numpy inside one method, so it is restated line by line here.
"""
from __future__ import annotations

import numpy as np


def order(data):
    """sog.py:264"""
    return np.lexsort((data["z"], data["y"], data["x"]))


def positions(data_s):
    """sog.py:279-309 -> (means_l (N,3) u8, means_u (N,3) u8, mins, maxs)"""
    def log_transform(v):
        return np.sign(v) * np.log(np.abs(v) + 1.0)
    l = [log_transform(data_s[a]) for a in "xyz"]
    mins = [np.min(v) for v in l]
    maxs = [np.max(v) for v in l]
    u = []
    for i in range(3):
        n = (l[i] - mins[i]) / (maxs[i] - mins[i])
        u.append(np.clip(n * 65535, 0, 65535).astype(np.uint16))
    lo = np.stack([v & 0xff for v in u], 1).astype(np.uint8)
    hi = np.stack([v >> 8 for v in u], 1).astype(np.uint8)
    return lo, hi, mins, maxs


def quats(data_s):
    """sog.py:315-386 -> (N,4) u8: three non-maximal components + 252 + argmax"""
    q = np.column_stack((data_s["rot_0"], data_s["rot_1"], data_s["rot_2"], data_s["rot_3"]))
    qn = q / np.linalg.norm(q, axis=1, keepdims=True)
    max_idx = np.abs(qn).argmax(axis=1)
    max_val = np.take_along_axis(qn, max_idx[:, None], axis=1).flatten()
    qn *= np.sign(max_val).reshape(-1, 1)
    qn *= np.sqrt(2.0)

    def quantize_vec(v):
        return np.clip((v * 0.5 + 0.5) * 255.0, 0, 255).astype(np.uint8)
    out = np.empty((len(q), 4), np.uint8)
    keep = {0: (1, 2, 3), 1: (0, 2, 3), 2: (0, 1, 3), 3: (0, 1, 2)}
    for m, cols in keep.items():
        sel = max_idx == m
        for j, c in enumerate(cols):
            out[sel, j] = quantize_vec(qn[sel, c])
    out[:, 3] = 252 + max_idx.astype(np.uint8)
    return out


def opacity_u8(data_s):
    """sog.py:457-459"""
    op_sig = 1.0 / (1.0 + np.exp(-data_s["opacity"]))
    return np.clip(op_sig * 255, 0, 255).astype(np.uint8)


def write_core_without_kmeans(data, scale_codebook, color_codebook):
    """Every statement of SogFormat.write between the table and its texel arrays EXCEPT the K-Means fits (sog.py:264-265,
    279-312, 315-386, 391, 408-431, 434, 446-459, 499-503), with the two scalar codebooks given: what the reference's single
    numpy thread does around its clustering calls.  -> dict of (texels, 4) uint8 images + the (N, 45) SH matrix + mins / maxs.
    bench.py times it as the CPU baseline of configs.sog_write_core_10m and compares its images with the device's."""
    N = len(data)
    width = int(np.ceil(np.sqrt(N) / 4) * 4)
    height = int(np.ceil(N / width / 4) * 4)
    indices = np.lexsort((data["z"], data["y"], data["x"]))              # :264
    data_s = data[indices]                                               # :265
    lo, hi, mins, maxs = positions(data_s)                               # :279-298
    means_l = np.full((height * width, 4), 255, dtype=np.uint8)          # :300-312
    means_u = np.full((height * width, 4), 255, dtype=np.uint8)
    means_l[:N, :3] = lo
    means_u[:N, :3] = hi
    quats_img = np.full((height * width, 4), 255, dtype=np.uint8)        # :340
    quats_img[:N] = quats(data_s)                                        # :315-386

    def quantize_to_codebook(vals, cb):                                  # :408-419
        if len(cb) == 1:
            return np.zeros_like(vals, dtype=np.uint8)
        idx = np.searchsorted(cb, vals)
        idx = np.clip(idx, 0, len(cb) - 1)
        left = np.maximum(idx - 1, 0)
        d_idx = np.abs(vals - cb[idx])
        d_left = np.abs(vals - cb[left])
        use_left = d_left < d_idx
        idx[use_left] = left[use_left]
        return idx.astype(np.uint8)
    s_data = np.concatenate([data_s["scale_0"], data_s["scale_1"], data_s["scale_2"]])      # :391 (what the 50 000-sample is drawn from)
    scb = np.array(scale_codebook)
    scales_img = np.zeros((height * width, 4), dtype=np.uint8)           # :421-431
    for c in range(3):
        scales_img[:N, c] = quantize_to_codebook(data_s["scale_%d" % c], scb)
    scales_img[:N, 3] = 255
    dc_data = np.concatenate([data_s["f_dc_0"], data_s["f_dc_1"], data_s["f_dc_2"]])        # :434
    ccb = np.array(color_codebook)
    sh0_img = np.zeros((height * width, 4), dtype=np.uint8)              # :446-459
    for c in range(3):
        sh0_img[:N, c] = quantize_to_codebook(data_s["f_dc_%d" % c], ccb)
    sh0_img[:N, 3] = opacity_u8(data_s)
    sh = None
    if "f_rest_44" in data.dtype.names:
        sh = np.column_stack([data_s["f_rest_%d" % i] for i in range(45)]).astype(np.float32)   # :499-503
    return {"means_l": means_l, "means_u": means_u, "quats": quats_img, "scales": scales_img, "sh0": sh0_img, "sh": sh,
            "mins": mins, "maxs": maxs, "n_concat": len(s_data) + len(dc_data)}
