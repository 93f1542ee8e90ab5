"""does a process that used the library's RCCL communicator exit cleanly?  usage: rccl_exit_probe.py [destroy|leak|del]"""
import importlib, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
mode = sys.argv[1] if len(sys.argv) > 1 else "destroy"
if "torch" in sys.argv:
    import torch
slab = importlib.import_module("3dgsconverter_amd.dist_slab")
be = slab.HipSlabBackend(0)
comm = slab.RcclComm(be.ctx, 0, 1, slab.RcclComm.unique_id())
a = be.buf("a", 64)
be.from_host(a, np.arange(8, dtype=np.float32))
comm.all_reduce(a, 8, slab.KIND_F32_MAX)
print(be.to_host(a, np.float32, 8))
if mode == "destroy":
    comm.close()
elif mode == "leak":
    import atexit
    atexit.unregister(comm.close)
    comm.ctx = None
    be.ctx.lib.gsx_comm_destroy  # not called
    be.ctx.handle = None  # never destroy the context either
elif mode == "del":
    comm.close()
    be.ctx.close()
print("leaving", mode, flush=True)
