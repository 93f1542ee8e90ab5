"""Generate tests/golden/cply_ref.npz from the REFERENCE's compressed-PLY writer (needs /root/reference; run here, the
fixture travels).  For every case: the splat order and the three elements the reference's own
``CompressedPlyFormat.write`` produced (oracle/refload.py:reference_cply -- only the plyfile container is intercepted),
and the order its ``_sort_morton_order`` gives when ``np.argsort`` is made stable (the only change).

    python -m oracle.make_golden_cply
"""
from __future__ import annotations

import hashlib
import os

import numpy as np

from . import cply, refload

CASES = [("plain", 3000, 1), ("clustered", 9000, 2), ("degree1", 777, 3), ("flat_scale", 1500, 4), ("plain", 256, 5), ("plain", 1, 6)]


def main():
    out = {}
    names = []
    for kind, n, seed in CASES:
        d = cply.cply_scene(n, seed, kind)
        own = refload.reference_cply(d)
        stable = refload.reference_cply(d, stable_ties=True)
        tag = "%s_%d_%d" % (kind, n, seed)
        names.append(tag)
        out[tag + "/order_ref"] = own["order"]
        out[tag + "/order_stable"] = stable["order"]
        out[tag + "/chunk"] = own["chunk"].view(np.float32).reshape(-1, 18)
        out[tag + "/vertex"] = own["vertex"].view(np.uint32).reshape(-1, 4)
        sh = own["sh"]
        out[tag + "/sh_names"] = np.array([] if sh is None else list(sh.dtype.names))
        out[tag + "/sh_sha256"] = np.array("" if sh is None else hashlib.sha256(sh.tobytes()).hexdigest())
        # the stable-order run's elements too: what the GPU writer must produce end to end
        out[tag + "/chunk_stable"] = stable["chunk"].view(np.float32).reshape(-1, 18)
        out[tag + "/vertex_stable"] = stable["vertex"].view(np.uint32).reshape(-1, 4)
        out[tag + "/sh_stable_sha256"] = np.array("" if stable["sh"] is None else hashlib.sha256(stable["sh"].tobytes()).hexdigest())
        _, depth = cply.morton_order(d["x"], d["y"], d["z"])
        print(tag, "ties differ:", not np.array_equal(own["order"], stable["order"]), "depth", depth)
    out["cases"] = np.array(names)
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "cply_ref.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
