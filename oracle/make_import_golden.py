"""CPU ORACLE support (test infrastructure): the `bioreason.*` names the reference's entry scripts import.

Walks the module-level AST of `reason.py` and `train_dna_qwen.py` (and the trainer / dataset modules they pull in) and records
every `from bioreason... import name` / `import bioreason...` with its line — the drop-in boundary of SURVEY §8(b) spelled out
as data.  Written to tests/golden/reference_imports.json so that the GPU box (no /root/reference) checks the same list;
`tests/test_reference_imports.py` resolves every entry with THIS repository first on PYTHONPATH, in a fresh interpreter.

    python oracle/make_import_golden.py            # rewrites the fixture (build container only)
"""
import ast
import json
import os

REF = "/root/reference"
SCRIPTS = ["reason.py", "train_dna_qwen.py", "bioreason/trainer/grpo_trainer.py", "bioreason/dna_modules/nucleotide_module.py",
           "bioreason/models/dna_llm.py", "bioreason/dataset/kegg.py", "bioreason/dataset/variant_effect.py",
           "bioreason/dataset/__init__.py", "bioreason/trainer/__init__.py", "bioreason/dna_modules/__init__.py"]
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "reference_imports.json")


def package_of(path: str) -> str:
    """dotted package a file's relative imports are resolved against"""
    parts = path[:-3].split("/")
    return ".".join(parts[:-1])


def bioreason_imports(path: str):
    tree = ast.parse(open(os.path.join(REF, path)).read())
    out = []
    for node in ast.walk(tree):
        if isinstance(node, ast.ImportFrom):
            mod = node.module or ""
            if node.level:                                     # relative import inside the bioreason package
                base = package_of(path).split(".")
                base = base[: len(base) - (node.level - 1)]
                mod = ".".join(base + ([mod] if mod else []))
            if mod == "bioreason" or mod.startswith("bioreason."):
                out.append({"file": path, "line": node.lineno, "module": mod, "names": [a.name for a in node.names]})
        elif isinstance(node, ast.Import):
            for a in node.names:
                if a.name == "bioreason" or a.name.startswith("bioreason."):
                    out.append({"file": path, "line": node.lineno, "module": a.name, "names": []})
    return sorted(out, key=lambda e: e["line"])


def collect():
    return [e for p in SCRIPTS for e in bioreason_imports(p)]


if __name__ == "__main__":
    data = collect()
    with open(OUT, "w") as fh:
        json.dump(data, fh, indent=1)
    print(f"{len(data)} import statements -> {OUT}")
