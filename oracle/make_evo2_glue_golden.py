"""CPU ORACLE support (test infrastructure): the reference's OWN Evo2 branch of `DNALLMModel` on a stand-in encoder.

`evo2` is absent, so StripedHyena-2 itself has no oracle; the GLUE around it does: this script imports the reference's
unmodified `DNALLMModel` (oracle/make_golden.py:import_reference), sets `dna_is_evo2 = True`, `dna_embedding_layer`, plugs in the toy
causal encoder of tests/test_evo2_glue.py (Evo2's call interface) and records `process_dna_embeddings` (dna_llm.py:123-179: the call
per sequence, the first-`sum(mask)`-rows slice over LEFT-padded rows) and the logits of `forward` in fp32 and in bf16, into
tests/golden/evo2_glue.pt.        python oracle/make_evo2_glue_golden.py          (build container only)
"""
import os
import sys

import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import dna_llm_oracle as O           # noqa: E402
from oracle.make_golden import import_reference, init_weights   # noqa: E402

TEXT = dict(vocab_size=512, hidden_size=128, intermediate_size=256, num_hidden_layers=2, num_attention_heads=4, num_key_value_heads=2,
            head_dim=32, rope_theta=1e6, max_position_embeddings=512)


def compute():
    import test_evo2_glue as T
    DNALLMModel = import_reference()
    torch.manual_seed(0)
    text = O.make_qwen3(TEXT, "eager")
    init_weights(text, 11)
    text.tie_weights()
    text.eval()
    enc = T.StandInEvo2().eval()
    proj = nn.Linear(T.HD, TEXT["hidden_size"])
    init_weights(proj, 12)
    ref = DNALLMModel.__new__(DNALLMModel)
    nn.Module.__init__(ref)
    ref.text_model, ref.dna_model, ref.dna_projection = text, enc, proj
    ref.dna_is_evo2, ref.dna_embedding_layer = True, T.LAYER
    ref.text_hidden_size, ref.dna_hidden_size, ref.dna_token_id = TEXT["hidden_size"], T.HD, 500
    batch, _ = T.make_batch(500)
    # bf16-representable weights, as in make_golden.py
    for mod in (text, proj):
        for p in mod.parameters():
            p.data = p.data.to(torch.bfloat16).float()
    with torch.no_grad():
        per_item = ref.process_dna_embeddings(batch["dna_tokenized"], batch["batch_idx_map"], 2)
        logits32 = ref(input_ids=batch["input_ids"].clone(), attention_mask=batch["attention_mask"], dna_tokenized=batch["dna_tokenized"],
                       batch_idx_map=batch["batch_idx_map"]).logits.float()
        n_calls = len(enc.calls)
        ref.to(torch.bfloat16)
        logits16 = ref(input_ids=batch["input_ids"].clone(), attention_mask=batch["attention_mask"], dna_tokenized=batch["dna_tokenized"],
                       batch_idx_map=batch["batch_idx_map"]).logits.float()
        ref.float()
    return {"batch": batch, "per_item": [t.clone() for t in per_item], "logits_fp32": logits32, "logits_bf16": logits16,
            "encoder": {k: v.clone() for k, v in enc.state_dict().items()},
            "text": {k: v.clone().to(torch.bfloat16) for k, v in text.state_dict().items()},
            "proj": {k: v.clone().to(torch.bfloat16) for k, v in proj.state_dict().items()}, "reference_calls_per_forward": n_calls // 2}


if __name__ == "__main__":
    fix = compute()
    torch.save(fix, os.path.join(ROOT, "tests", "golden", "evo2_glue.pt"))
    print("reference encoder calls per pass:", fix["reference_calls_per_forward"], "logits", tuple(fix["logits_fp32"].shape))
