"""Rollout-only probe (full-size models, B=8, P=2180): prints ms per decode step."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bioreason_amd import configs
from bioreason_amd.dna_llm import DNALLMModel
from bioreason_amd.synth import synth_prompt_batch

dev = torch.device("cuda:0")
C = int(os.environ.get("PROBE_C", "64"))
m = DNALLMModel(configs.qwen3_config(), configs.nt_v2_config(), device=dev)
m.text_model.init_weights(0.02, seed=1); m.dna_model.init_weights(0.02, seed=2)
if os.environ.get("PROBE_LORA", "1") == "1":
    m.text_model.apply_lora(r=32, alpha=64.0, arena=m.arena)
b = synth_prompt_batch(B=8, n_unique=1, dna_token_id=m.dna_token_id, device=dev)
kw = dict(input_ids=b["input_ids"], attention_mask=b["attention_mask"], dna_tokenized=b["dna_tokenized"], batch_idx_map=b["batch_idx_map"],
          dna_alias=b["dna_alias"], prompt_alias=b["prompt_alias"], do_sample=True, temperature=0.6, top_k=20, top_p=0.95, eos_token_id=None)
MODES = {"sg": (True, True), "se": (True, False), "ng": (False, True), "ne": (False, False)}
for shared_dec, graph in [MODES[k] for k in os.environ.get("PROBE_MODES", "sg,se,ng,ne").split(",")]:
    kw["shared_prefix_decode"], kw["use_graph"] = shared_dec, graph
    for it in range(2):
        torch.cuda.synchronize(); t0 = time.time()
        m.generate(max_new_tokens=1, **kw); torch.cuda.synchronize(); t1 = time.time()
        m.generate(max_new_tokens=C, **kw); torch.cuda.synchronize(); t2 = time.time()
        print("shared", shared_dec, "graph", graph, "prefill+1 ms %.1f" % ((t1 - t0) * 1e3), "gen", C, "ms %.1f" % ((t2 - t1) * 1e3),
              "per decode step ms %.3f" % (((t2 - t1) - (t1 - t0)) * 1e3 / (C - 1)), flush=True)

prof = {}
kw["shared_prefix_decode"], kw["use_graph"] = True, True
torch.cuda.synchronize(); t0 = time.time()
m.generate(max_new_tokens=C, profile=prof, **kw); torch.cuda.synchronize()
print("generate total ms %.1f" % ((time.time() - t0) * 1e3), {k: round(v, 2) for k, v in prof.items()}, flush=True)

# cost of rebuilding the merged rollout weights (done once per training step, after the optimizer touched the adapters)
from bioreason_amd import generation
eng = m.text_model.engine
for it in range(2):
    eng._rollout = None
    torch.cuda.synchronize(); t0 = time.time()
    generation.rollout_weights(m.text_model); torch.cuda.synchronize()
    print("rollout_weights rebuild ms %.2f" % ((time.time() - t0) * 1e3), flush=True)
