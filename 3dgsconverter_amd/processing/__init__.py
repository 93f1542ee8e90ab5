"""Drop-in for ``gsconverter.processing`` (reference: gsconverter/processing/__init__.py)."""
from .data_processor import DataProcessor
from . import gpu_ops

__all__ = ["DataProcessor", "gpu_ops"]
