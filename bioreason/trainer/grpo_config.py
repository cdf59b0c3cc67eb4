"""bioreason/trainer/grpo_config.py:21-365 -> bioreason_amd.grpo_trainer"""
from bioreason_amd.grpo_trainer import DNALLMGRPOConfig  # noqa: F401
