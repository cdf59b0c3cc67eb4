from .grpo_config import DNALLMGRPOConfig
from .grpo_trainer import DNALLMGRPOTrainer

__all__ = ["DNALLMGRPOConfig", "DNALLMGRPOTrainer"]
