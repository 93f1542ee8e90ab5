"""3dgsconverter_amd -- MI355X (gfx950) implementation of 3dgsconverter's point-cloud
filtering hot path (SOR exact-KNN mean distance, voxel-density clustering, SOG K-Means
codebook) behind the reference's own ``gsconverter.processing`` API.

The directory name starts with a digit, so import it with
``importlib.import_module("3dgsconverter_amd")``.

    gsx = importlib.import_module("3dgsconverter_amd")
    gsx.processing.DataProcessor(data).remove_flyers(k, sigma)        # drop-in
    gsx.install()   # make the reference's converter.py / sog.py use this package

Host code is numpy + ctypes over the C ABI in include/gsx_hip.h; kernels are hand-written
HIP in csrc/.  No CPU fallback: without libgsx_hip.so and a gfx950 device the filter
entry points raise ``GsxError``.
"""
from . import _lib
from ._lib import GsxError, has_hip, device_count, release_arenas as release_device_cache  # noqa: F401
from . import processing  # noqa: F401
from .processing import DataProcessor, gpu_ops  # noqa: F401
from .install import install, uninstall  # noqa: F401

__all__ = ["processing", "DataProcessor", "gpu_ops", "GsxError", "has_hip", "device_count", "install", "uninstall", "release_device_cache"]
