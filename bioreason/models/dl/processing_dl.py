"""bioreason/models/dl/processing_dl.py:36-300 -> bioreason_amd.processing"""
from bioreason_amd.processing import DLProcessor  # noqa: F401
