"""`DNABaseModule` / `NucleotideDNAModule` — the adapter object the reference's GRPO trainer asks for model class,
processor class, input keywords and prompt preparation (bioreason/dna_modules/dna_module.py:5-49,
nucleotide_module.py:16-262; SURVEY §8b "other preserved classes").  Thin host glue: the answers are the reference's,
the classes they name are this package's.

`prepare_prompt` applies the chat template the way trl's `maybe_apply_chat_template` does for a prompt-only
conversational example (trl is absent here): `apply_chat_template(messages, tokenize=False, add_generation_prompt=True)`;
a plain-string prompt passes through."""
import re
from abc import ABC, abstractmethod
from typing import Any, Callable, Dict, List, Type


class DNABaseModule(ABC):
    def __init__(self):
        super().__init__()

    @abstractmethod
    def get_dnallm_key(self):
        ...

    @abstractmethod
    def get_model_class(self, model_id: str, model_init_kwargs: dict):
        ...

    def post_model_init(self, model, processing_class):
        pass

    def is_embeds_input(self):
        return False

    @abstractmethod
    def get_processing_class(self):
        ...

    @abstractmethod
    def get_dnallm_modules_keywords(self):
        ...

    @abstractmethod
    def get_custom_multimodal_keywords(self):
        ...

    @abstractmethod
    def get_non_generate_params(self):
        ...

    @abstractmethod
    def get_custom_processing_keywords(self):
        ...

    @abstractmethod
    def prepare_prompt(self, processing_class, inputs):
        ...

    @abstractmethod
    def prepare_model_inputs(self, processing_class, model, prompts_text, batch_dna_sequences, return_tensors, padding,
                             padding_side, add_special_tokens):
        ...


class NucleotideDNAModule(DNABaseModule):
    def get_dnallm_key(self) -> str:                                   # nucleotide_module.py:28-35
        return "qwen"

    def get_model_class(self, model_id: str, model_init_kwargs: Dict[str, Any]) -> Type:   # :37-55
        if "DNALLM" in model_id:
            from .dna_llm import DNALLMModel
            return DNALLMModel
        raise ValueError(f"Unsupported model: {model_id}")

    def get_processing_class(self) -> Type:                            # :68-75
        from .processing import DLProcessor
        return DLProcessor

    def get_dnallm_modules_keywords(self) -> List[str]:                # :77-86 (kept out of the LoRA targets)
        return ["dna"]

    def get_custom_multimodal_keywords(self) -> List[str]:             # :88-95
        return ["dna_tokenized", "batch_idx_map"]

    def get_non_generate_params(self) -> List[str]:                    # :97-104
        return []

    def get_custom_processing_keywords(self) -> List[tuple]:           # :106-113
        return [("dna_tokenizer", "max_length")]

    def is_embeds_input(self) -> bool:                                 # :178-186: generate() returns completions only
        return True

    def prepare_prompt(self, processing_class: Any, inputs: List[Dict[str, Any]]) -> List[str]:      # :115-132
        out = []
        for ex in inputs:
            p = ex["prompt"]
            if isinstance(p, str):
                out.append(p)
                continue
            tok = getattr(processing_class, "tokenizer", processing_class)
            kw = {}
            if getattr(processing_class, "chat_template", None) is not None:
                kw["chat_template"] = processing_class.chat_template
            out.append(tok.apply_chat_template(p, tokenize=False, add_generation_prompt=True, **kw))
        return out

    def prepare_model_inputs(self, processing_class: Any, model: Any, prompts_text: List[str],
                             batch_dna_sequences: List[List[str]], return_tensors: str = "pt", padding: bool = True,
                             padding_side: str = "left", add_special_tokens: bool = False) -> Dict[str, Any]:   # :134-176
        m = model.module if hasattr(model, "module") else model
        return processing_class(text=prompts_text, batch_dna_sequences=batch_dna_sequences, return_tensors=return_tensors,
                                padding=padding, padding_side=padding_side, add_special_tokens=add_special_tokens,
                                max_length_text=m.max_length_text, max_length_dna=m.max_length_dna)

    @staticmethod
    def get_question_template() -> str:                                # :188-196
        return "{Question}"

    @staticmethod
    def format_reward_rec(completions: List[Any], **kwargs) -> List[float]:               # :198-234 (debug log file omitted)
        pattern = r"<think>.*?</think>\s*<answer>.*?\{.*\[\d+,\s*\d+,\s*\d+,\s*\d+\].*\}.*?</answer>"
        return [1.0 if re.search(pattern, c[0]["content"], re.DOTALL) is not None else 0.0 for c in completions]

    @staticmethod
    def select_reward_func(func: str, task_type: str) -> Callable:     # :236-262 (`iou_reward` does not exist in the reference either)
        if func == "format" and task_type == "rec":
            return NucleotideDNAModule.format_reward_rec
        raise ValueError(f"Unsupported reward function: {func}")
