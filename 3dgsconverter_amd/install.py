"""Make the reference package use this implementation without editing it.

``install()`` rebinds the three names the reference's orchestrator and SOG writer resolve
at call time:
    gsconverter.converter.DataProcessor            (converter.py:10 -> used at :150; bound to ChainedDataProcessor,
                                                    which keeps the coordinates in HBM across density -> SOR and whose
                                                    filter methods return None -- the orchestrator ignores them)
    gsconverter.processing.DataProcessor / gsconverter.processing.data_processor.DataProcessor
                                                   (the EAGER class: every method returns self.data like the reference)
    gsconverter.formats.sog.gpu_ops                (sog.py:11 -> used at :402,443,524,544)
    gsconverter.processing.gpu_ops                 (data_processor.py:142 imports it lazily)
so ``gsconverter``'s CLI (main.py) and every flag in SURVEY.md 8(b) keep working.
"""
from __future__ import annotations

import importlib
import sys

_saved = {}


def install(sog_writer: bool = True):
    """sog_writer: also rebind ``gsconverter.formats.sog.SogFormat.write`` to formats/sog_writer.py:write_sog (spatial
    sort, quaternion packing, codebook quantiser and SH palette on the GPU; identical bytes where the reference is
    deterministic) and ``gsconverter.formats.compressed_ply.CompressedPlyFormat.write`` to
    formats/compressed_ply_writer.py:write_compressed_ply (Morton order, chunk bounds and packers on the GPU)."""
    from . import processing
    from .processing import gpu_ops
    # the orchestrator ignores the filters' return values (converter.py:196-236), so ITS name gets the lazy class (coordinates
    # stay in HBM across filters); the public names keep the eager class, whose methods return self.data like the reference's
    from .processing.data_processor import ChainedDataProcessor, DataProcessor
    import gsconverter.processing as rp  # type: ignore  (raises ImportError if the reference is absent)
    import gsconverter.processing.data_processor as rdp  # type: ignore
    from .processing import data_processor as mine
    if rdp.DataProcessor is not DataProcessor and rdp.DataProcessor.__module__.startswith("gsconverter"):
        mine._REFERENCE_CLASS = rdp.DataProcessor  # methods with no device implementation are forwarded to it
    targets = [(rp, "DataProcessor", DataProcessor), (rdp, "DataProcessor", DataProcessor),
               (rp, "gpu_ops", gpu_ops)]
    for modname, attr, val in (("gsconverter.converter", "DataProcessor", ChainedDataProcessor),
                               ("gsconverter.formats.sog", "gpu_ops", gpu_ops)):
        try:
            targets.append((importlib.import_module(modname), attr, val))
        except ImportError:
            pass  # plyfile / pillow missing: that module of the REFERENCE is not importable, nothing to rebind
    for mod, attr, val in targets:
        _saved.setdefault((mod.__name__, attr), getattr(mod, attr, None))
        setattr(mod, attr, val)
    if sog_writer:
        # only the REFERENCE's module may be missing (pillow / plyfile absent: its own writer cannot run either);
        # a failure importing this package's writers propagates -- no silent return to the CPU writer
        try:
            sogmod = importlib.import_module("gsconverter.formats.sog")
        except ImportError:
            sogmod = None
        if sogmod is not None:
            from .formats.sog_writer import write_sog
            _saved.setdefault(("sogformat", "write"), sogmod.SogFormat.write)
            sogmod.SogFormat.write = lambda self, data, path, **kw: write_sog(data, path, **kw)
        try:
            cpmod = importlib.import_module("gsconverter.formats.compressed_ply")
        except ImportError:
            cpmod = None
        if cpmod is not None:
            from .formats.compressed_ply_writer import write_compressed_ply
            _saved.setdefault(("cplyformat", "write"), cpmod.CompressedPlyFormat.write)
            cpmod.CompressedPlyFormat.write = lambda self, data, path, **kw: write_compressed_ply(data, path, **kw)
    _saved.setdefault(("sys.modules", "gsconverter.processing.gpu_ops"),
                      sys.modules.get("gsconverter.processing.gpu_ops"))
    sys.modules["gsconverter.processing.gpu_ops"] = gpu_ops
    return processing


def uninstall():
    for (modname, attr), val in list(_saved.items()):
        if modname == "sogformat":
            importlib.import_module("gsconverter.formats.sog").SogFormat.write = val
            continue
        if modname == "cplyformat":
            importlib.import_module("gsconverter.formats.compressed_ply").CompressedPlyFormat.write = val
            continue
        if modname == "sys.modules":
            if val is None:
                sys.modules.pop(attr, None)
            else:
                sys.modules[attr] = val
        elif val is not None:
            setattr(importlib.import_module(modname), attr, val)
    _saved.clear()
