"""bioreason/models/evo2_tokenizer.py:16-218 -> bioreason_amd.evo2_tokenizer"""
from bioreason_amd.evo2_tokenizer import CharLevelTokenizer, Evo2Tokenizer, register_evo2_tokenizer  # noqa: F401
