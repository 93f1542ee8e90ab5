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
