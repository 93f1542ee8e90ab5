import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
gsx = importlib.import_module("3dgsconverter_amd")
L = gsx._lib
from oracle import sor as osor
import test_sor_tree_gpu as T
only = sys.argv[1] if len(sys.argv) > 1 else None
for name, xyz, k in T._shapes(np.random.default_rng(5)):
    if only and name != only: continue
    xyz = np.ascontiguousarray(xyz)
    print("shape", name, len(xyz), k, flush=True)
    res = L.sor_filter(xyz, k, 1.0, algo=3, want_info=True)
    ref = osor.mean_dists_ckdtree(xyz, k)
    print("   ", T._explain(res["mean_dists"], ref), res["info"], flush=True)
