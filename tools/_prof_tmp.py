import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
from tools.probe_sog import table
L = importlib.import_module("3dgsconverter_amd._lib")
w = importlib.import_module("3dgsconverter_amd.formats.sog_writer")
sd = importlib.import_module("3dgsconverter_amd.formats.sog_device")
n = 10_000_000
base = table(n, 7)
colors = np.zeros((n, 3), np.uint8)
wide = L.host_append_u8_columns(base, ("red", "green", "blue"), colors)
print(wide.dtype.itemsize)
for _ in range(3):
    t = time.perf_counter(); rows, lay = sd.table_layout(wide); t1 = time.perf_counter()
    print("table_layout (host pack) %.1f ms" % ((t1 - t) * 1e3)); del rows
for _ in range(3):
    t = time.perf_counter(); core = w.encode(wide, 2, device_resident=True); print("encode 251-byte rows %.1f ms" % ((time.perf_counter() - t) * 1e3)); del core
