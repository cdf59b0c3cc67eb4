"""`bioreason` import paths of the reference (reason.py:35-39, train_dna_qwen.py:27-36, grpo_trainer.py:65-66) resolved to the
MI355X implementation in `bioreason_amd`: with this repository on PYTHONPATH, `from bioreason.models.dna_llm import
DNALLMModel` etc. give the HIP-backed classes and the reference's scripts need no edit for the hot path.
Only the hot-path modules exist here; `bioreason.dataset` carries the collate helpers of row N1."""
