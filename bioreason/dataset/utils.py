"""bioreason/dataset/utils.py:6-59 -> bioreason_amd.datasets"""
from bioreason_amd.datasets import torch_to_hf_dataset, truncate_dna  # noqa: F401
