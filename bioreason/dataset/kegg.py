"""bioreason/dataset/kegg.py -> bioreason_amd.datasets (records, :14-220, :336-382) and bioreason_amd.collate (the SFT collate on
the hot path's input side, :223-333)"""
from bioreason_amd.collate import qwen_dna_collate_fn  # noqa: F401
from bioreason_amd.datasets import (  # noqa: F401
    KEGGDataset,
    create_kegg_dataloader,
    dna_collate_fn,
    format_kegg_for_dna_llm,
    format_kegg_for_llm,
    get_format_kegg_function,
    split_kegg_dataset,
)
