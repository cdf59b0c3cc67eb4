"""stamps inside bra_dec_attn_both during a real rollout (full-size model, 1 prompt x 8): where its ~11 us go"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bioreason_amd import configs
from bioreason_amd import _lib as _bra_lib
_bra_lib.use_debug_library()          # knobs / probes / persistent step: libbioreason_hip_debug.so (include/bioreason_hip_debug.h)
from bioreason_amd._lib import get_lib
from bioreason_amd.dna_llm import DNALLMModel
from bioreason_amd.synth import synth_prompt_batch
dev = torch.device("cuda:0")
m = DNALLMModel(configs.qwen3_config(), configs.nt_v2_config(), device=dev)
m.text_model.init_weights(0.02, seed=1); m.dna_model.init_weights(0.02, seed=2)
m.text_model.apply_lora(r=32, alpha=64.0, arena=m.arena)
b = synth_prompt_batch(B=8, n_unique=1, dna_token_id=m.dna_token_id, device=dev)
kw = dict(input_ids=b["input_ids"], attention_mask=b["attention_mask"], dna_tokenized=b["dna_tokenized"], batch_idx_map=b["batch_idx_map"],
          dna_alias=b["dna_alias"], prompt_alias=b["prompt_alias"], do_sample=True, temperature=0.6, top_k=20, top_p=0.95, eos_token_id=None,
          use_graph=False)
probe = torch.zeros(16, dtype=torch.int64, device=dev)
m.generate(max_new_tokens=8, **kw)
get_lib().call("bra_debug_set_probe", probe)
for C in (40, 200):
    m.generate(max_new_tokens=C, **kw); torch.cuda.synchronize()
    p = probe.cpu().tolist()
    print(f"C={C}: prompt-part wave: entry->issued {0.01*(p[1]-p[0]):.2f}  ->q arrived {0.01*(p[2]-p[1]):.2f}  ->q rotated {0.01*(p[3]-p[2]):.2f}  "
          f"->scores {0.01*(p[4]-p[3]):.2f}  ->partials issued {0.01*(p[5]-p[4]):.2f}  total {0.01*(p[5]-p[0]):.2f} us | completion-part wave: "
          f"prologue {0.01*(p[9]-p[8]):.2f}  scores {0.01*(p[10]-p[9]):.2f}  pv+store {0.01*(p[11]-p[10]):.2f}  total {0.01*(p[11]-p[8]):.2f} us; "
          f"entry skew completion-prompt {0.01*(p[8]-p[0]):.2f} us", flush=True)
get_lib().call("bra_debug_set_probe", None)
