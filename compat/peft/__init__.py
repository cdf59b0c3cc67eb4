"""`peft` for PYTHONPATH (put `<repo>/compat` in front): everything an installed peft exports, with `get_peft_model`,
`prepare_model_for_kbit_training`, `PeftModel.from_pretrained` dispatched on the model type — a bioreason_amd HIP text model gets the
adapters its engine executes, any other model goes to the real peft unchanged (bioreason_amd/peft_compat.py).  Without an installed
peft the names the reference's entry points import (`reason.py:25`, `train_dna_qwen.py:16`) are provided for HIP models alone."""
__bioreason_amd_shim__ = True

from bioreason_amd import peft_compat as _c

_real = _c._real_peft()
if _real is not None:
    for _k in dir(_real):
        if not _k.startswith("__"):
            globals()[_k] = getattr(_real, _k)
    __version__ = getattr(_real, "__version__", "0")

    class PeftModel(_real.PeftModel):                # keeps isinstance(x, peft.PeftModel) for real wrappers
        @classmethod
        def from_pretrained(cls, model, model_id, *a, **kw):
            if _c.is_hip_text_model(model):
                return _c.PeftModel.from_pretrained(model, model_id, *a, **kw)
            return _real.PeftModel.from_pretrained(model, model_id, *a, **kw)
else:
    __version__ = "0+bioreason_amd"
    PeftModel = _c.PeftModel
    LoraConfig = _c.LoraConfig
    PeftConfig = _c.LoraConfig

get_peft_model = _c.get_peft_model
prepare_model_for_kbit_training = _c.prepare_model_for_kbit_training
