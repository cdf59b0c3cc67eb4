"""bioreason/dataset/kegg.py:223-333 (the SFT collate on the hot path's input side) -> bioreason_amd.collate"""
from bioreason_amd.collate import qwen_dna_collate_fn  # noqa: F401
