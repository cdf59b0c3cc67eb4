"""bioreason_amd — MI355X-native (gfx950) hot path of BioReason's DNA-LLM.

Hand-written HIP kernels (bioreason_amd/csrc -> libbioreason_hip.so, C ABI in
include/bioreason_hip.h) under a Python host layer that mirrors the reference's
`bioreason.models` / `bioreason.trainer` interfaces.
"""
__version__ = "0.1.0"
