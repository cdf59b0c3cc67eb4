"""bioreason/dna_modules/dna_module.py:5-49 -> bioreason_amd.dna_modules"""
from bioreason_amd.dna_modules import DNABaseModule  # noqa: F401
