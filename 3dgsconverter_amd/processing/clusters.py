"""Host side of the voxel-density filter: connected components of the dense voxels.

Mirrors data_processor.py:57-106.  At most 100/threshold% voxels are dense (<= 1000 for
sensitivity 0, <= 181 for 0.5), so this stays on the host exactly like the reference; the
O(N) parts (voxel keys, occupancy counts, per-point membership) run on the GPU.

Tie-break parity: when two clusters have the same voxel count and keep_multicluster is
False the reference keeps whichever its python ``set`` iteration meets first (the sort at
data_processor.py:95 is stable).  The same set is rebuilt here -- same tuples of ints
inserted in np.unique's lexicographic order -- so the iteration order is identical.
"""
from __future__ import annotations

from collections import deque

_NEIGH = ((-1, 0, 0), (1, 0, 0), (0, -1, 0), (0, 1, 0), (0, 0, -1), (0, 0, 1))


def connected_clusters(dense_keys_lex):
    """dense_keys_lex: iterable of (x,y,z) int tuples in lexicographic order -> list of sets."""
    dense = set(dense_keys_lex)
    seen = set()
    clusters = []
    for v in dense:  # set iteration order == the reference's (same elements, same insertion order)
        if v in seen:
            continue
        seen.add(v)
        comp = {v}
        todo = deque((v,))
        while todo:
            cx, cy, cz = todo.popleft()
            for dx, dy, dz in _NEIGH:
                nb = (cx + dx, cy + dy, cz + dz)
                if nb in dense and nb not in seen:
                    seen.add(nb)
                    comp.add(nb)
                    todo.append(nb)
        clusters.append(comp)
    return clusters


def select_clusters(clusters, keep_multicluster: bool):
    """data_processor.py:95-106 -> (kept voxel set, kept cluster count, largest size)."""
    if not clusters:
        return set(), 0, 0
    ordered = sorted(clusters, key=len, reverse=True)
    largest = len(ordered[0])
    floor = largest * 0.05 if keep_multicluster else largest
    kept = set()
    n_kept = 0
    for comp in ordered:
        if len(comp) >= floor:
            kept |= comp
            n_kept += 1
            if not keep_multicluster:
                break
    return kept, n_kept, largest
