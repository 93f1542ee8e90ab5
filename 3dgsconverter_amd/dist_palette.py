"""SOG SH-palette K-Means across the GPUs of one node: independent chunks, no collective in the compute.

The reference clusters the SH coefficients chunk by chunk -- ``num_chunks`` (<= 64) independent K-Means problems
(formats/sog.py:527-552) -- so the scale-out is to deal the chunks out, eight per GPU on an 8-GPU node, and
all-gather the (small) results: SURVEY.md 8(e) row 3.  ``palette_kmeans`` is a drop-in for that loop: it returns what
the loop leaves in ``centroids`` / ``labels`` (sog.py:554-555).  The initial centroids are drawn exactly like the
reference does (one ``np.random.choice`` per chunk from numpy's global stream, gpu_ops.py:182), by EVERY rank and
for EVERY chunk, so the result does not depend on how many GPUs share the work.
"""
from __future__ import annotations

import numpy as np


def palette_plan(n: int, compression_level: int = 0) -> dict:
    """formats/sog.py:513-529: palette size and chunking of the SH-N K-Means"""
    official_standard_k = min(64, 2 ** int(np.floor(np.log2(n / 1024)))) * 1024
    if compression_level <= 3:
        target_k = min(65536, official_standard_k)
    elif compression_level <= 6:
        target_k = min(16384, official_standard_k)
    else:
        target_k = min(4096, official_standard_k)
    target_k = max(256, target_k)
    num_chunks = max(1, min(64, n // 1024))
    chunk_size = int(np.ceil(n / num_chunks))
    k_per_chunk = max(16, int(np.ceil(target_k / num_chunks)))
    return {"target_k": int(target_k), "num_chunks": int(num_chunks), "chunk_size": chunk_size, "k_per_chunk": k_per_chunk}


def palette_kmeans(sh_data: np.ndarray, compression_level: int = 0, max_iter: int = 10, kmeans=None, comm=None, be=None):
    """sh_data: (N, D) float32, the same array on every rank.  kmeans(data, k, max_iter, init_centroids) -> (centroids,
    labels): ``gpu_ops.kmeans`` by default.  comm / be: a communicator and buffer backend of dist_slab (RcclComm +
    HipSlabBackend, or the host stand-ins of the tests); None = one GPU does every chunk.
    -> (centroids f32 [P, D], labels int64 [N] into the concatenated palette)"""
    lanes_default = kmeans is None   # the product path: this rank's chunks on a few concurrent contexts (_lib.kmeans_lloyd_many)
    if kmeans is None:
        from .processing import gpu_ops
        kmeans = lambda d, k, it, init: gpu_ops.kmeans(d, k, max_iter=it, init_centroids=init)
    n, d = sh_data.shape
    plan = palette_plan(n, compression_level)
    rank, world = (comm.rank, comm.world) if comm is not None else (0, 1)
    bounds = []
    for i in range(plan["num_chunks"]):
        start, end = i * plan["chunk_size"], min((i + 1) * plan["chunk_size"], n)
        if start >= end:
            break
        bounds.append((start, end))
    ks = [min(e - s, plan["k_per_chunk"]) for s, e in bounds]
    # every rank draws every chunk's initial centroids from the global stream, in chunk order (see module docstring);
    # a chunk whose k >= its size takes the reference's shortcut (gpu_ops.py:30-31) and draws nothing
    inits = [np.random.choice(e - s, k, replace=False) if k < e - s else None for (s, e), k in zip(bounds, ks)]
    mine = [i for i in range(len(bounds)) if i % world == rank]
    results = {}
    if lanes_default:
        from . import _lib
        todo = [i for i in mine if inits[i] is not None]
        chunks = {i: np.ascontiguousarray(sh_data[bounds[i][0]:bounds[i][1]], dtype=np.float32) for i in todo}
        done = _lib.kmeans_lloyd_many([(chunks[i], chunks[i][inits[i]]) for i in todo], max_iter)
        results.update(dict(zip(todo, done)))
        for i in mine:
            if inits[i] is None:   # k >= rows: the reference's shortcut (gpu_ops.py:30-31)
                s, e = bounds[i]
                results[i] = kmeans(np.ascontiguousarray(sh_data[s:e], dtype=np.float32), ks[i], max_iter, None)
    else:
        for i in mine:
            s, e = bounds[i]
            chunk = np.ascontiguousarray(sh_data[s:e], dtype=np.float32)
            init = chunk[inits[i]] if inits[i] is not None else None
            results[i] = kmeans(chunk, ks[i], max_iter, init)
    if world > 1:
        # all-gather of padded per-rank blocks: [slots, kmax, d] centroids and [slots, chunk_size] labels
        slots = -(-len(bounds) // world)
        kmax, cs = max(ks), plan["chunk_size"]
        cen = np.zeros((slots, kmax, d), np.float32)
        lab = np.zeros((slots, cs), np.int32)
        for j, i in enumerate(mine):
            c, l = results[i]
            cen[j, :len(c)] = c
            lab[j, :len(l)] = l
        out = {}
        for name, arr in (("pal_cen", cen), ("pal_lab", lab)):
            sb, rb = be.buf(name + "_s", arr.nbytes), be.buf(name + "_r", arr.nbytes * world)
            be.from_host(sb, arr)
            comm.all_gather(sb, rb, arr.nbytes)
            out[name] = be.to_host(rb, arr.dtype, arr.size * world).reshape((world,) + arr.shape)
        for i in range(len(bounds)):
            if i % world != rank:
                s, e = bounds[i]
                results[i] = (out["pal_cen"][i % world, i // world, :ks[i]].copy(), out["pal_lab"][i % world, i // world, :e - s].copy())
    all_centroids, all_labels, offset = [], [], 0
    for i in range(len(bounds)):   # sog.py:546-552
        c, l = results[i]
        all_labels.append(l.astype(np.int64) + offset)
        all_centroids.append(c)
        offset += len(c)
    return np.vstack(all_centroids), np.concatenate(all_labels)
