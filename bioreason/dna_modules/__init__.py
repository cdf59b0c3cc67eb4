from .dna_module import DNABaseModule
from .nucleotide_module import NucleotideDNAModule

__all__ = ["DNABaseModule", "NucleotideDNAModule"]
