"""bioreason/trainer/grpo_trainer.py:72-119, 122-904 -> bioreason_amd.grpo_trainer"""
from bioreason_amd.grpo_trainer import DNALLMGRPOTrainer, RepeatRandomSampler  # noqa: F401
