"""collate helpers of SURVEY §8f N1 (bioreason/dataset/kegg.py:252-327 label masking) -> bioreason_amd.collate"""
from bioreason_amd import collate  # noqa: F401
