"""CPU ORACLE support (test infrastructure): the REFERENCE'S OWN data-format functions and Evo2 tokenizer wrapper.

`bioreason/dataset/{kegg,variant_effect}.py` import `trl` (absent), so the modules cannot be imported; their functions are plain
Python: this file parses the sources, compiles the AST nodes of the named functions / classes unmodified and hands them out.
`bioreason/models/evo2_tokenizer.py` imports cleanly and is loaded as a module from its path.  Nothing is restated here.
Only tests/ and oracle/ scripts import this file; it needs /root/reference (build container only).
"""
from __future__ import annotations

import ast
import importlib.util
import json
import os
import random
from typing import Any, Dict, List, Tuple

import torch
from torch.utils.data import DataLoader, Dataset

REF = "/root/reference"
FILES = {
    "utils": "bioreason/dataset/utils.py",
    "kegg": "bioreason/dataset/kegg.py",
    "variant_effect": "bioreason/dataset/variant_effect.py",
}


def available() -> bool:
    return os.path.exists(os.path.join(REF, FILES["kegg"]))


def namespace(which: str, names) -> Dict[str, Any]:
    """the top-level functions / classes `names` of one reference file, compiled from their own source lines"""
    path = os.path.join(REF, FILES[which])
    tree = ast.parse(open(path).read())
    from datasets import Dataset as HFDataset
    ns = {"json": json, "os": os, "random": random, "torch": torch, "Dataset": Dataset, "DataLoader": DataLoader, "Any": Any,
          "Dict": Dict, "List": List, "Tuple": Tuple, "HFDataset": HFDataset, "TorchDataset": Dataset, "Union": None}
    want = set(names)
    body = [n for n in tree.body if isinstance(n, (ast.FunctionDef, ast.ClassDef)) and n.name in want]
    assert {n.name for n in body} == want, (want, [n.name for n in body])
    exec(compile(ast.Module(body=body, type_ignores=[]), path, "exec"), ns)
    return ns


def evo2_tokenizer_module():
    """the reference's evo2_tokenizer.py as a module object (it only needs transformers)"""
    path = os.path.join(REF, "bioreason/models/evo2_tokenizer.py")
    spec = importlib.util.spec_from_file_location("_ref_evo2_tokenizer", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod
