from .dna_llm import DNALLMModel

__all__ = ["DNALLMModel"]
