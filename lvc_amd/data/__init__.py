"""Test-time input transform of the detector (the part of detectron2/data that sits on the device here)."""
from .transforms import ResizeShortestEdge, ResizeTransform, resample_coeffs

__all__ = ["ResizeShortestEdge", "ResizeTransform", "resample_coeffs"]
