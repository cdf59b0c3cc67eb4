"""the names bioreason/dataset/__init__.py:1-11 exports; collate helpers of SURVEY §8f N1 live in bioreason_amd.collate"""
from bioreason_amd import collate  # noqa: F401
from .kegg import KEGGDataset, split_kegg_dataset
from .utils import torch_to_hf_dataset, truncate_dna
from .variant_effect import get_format_variant_effect_function

__all__ = ["KEGGDataset", "split_kegg_dataset", "torch_to_hf_dataset", "truncate_dna", "get_format_variant_effect_function"]
