"""bioreason/dna_modules/nucleotide_module.py:16-263 -> bioreason_amd.dna_modules"""
from bioreason_amd.dna_modules import NucleotideDNAModule  # noqa: F401
