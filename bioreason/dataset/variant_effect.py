"""bioreason/dataset/variant_effect.py:14-97 -> bioreason_amd.datasets"""
from bioreason_amd.datasets import (  # noqa: F401
    clean_variant_effect_example,
    clean_variant_effect_non_snv_example,
    format_variant_effect_for_dna_llm,
    format_variant_effect_for_llm,
    get_format_variant_effect_function,
)
