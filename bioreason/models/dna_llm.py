"""bioreason/models/dna_llm.py:18-306 -> bioreason_amd.dna_llm"""
from bioreason_amd.dna_llm import DNALLMModel  # noqa: F401
