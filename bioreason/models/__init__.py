from .dna_llm import DNALLMModel
from .evo2_tokenizer import Evo2Tokenizer

__all__ = ["DNALLMModel", "Evo2Tokenizer"]
