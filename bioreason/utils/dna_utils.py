"""bioreason/utils/dna_utils.py:7-11: the input type alias of the DNA processor"""
from typing import Union

import numpy as np
import torch

DNAInput = Union[str, list[int], np.ndarray, torch.Tensor, list[str], list[list[int]], list[np.ndarray], list[torch.Tensor]]
