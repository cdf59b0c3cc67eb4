"""does completion-independent MFMA work hide under the token loop?
Full-size models, one prompt x 8 rollouts (cfg-3).  Three timings on one box:
  (a) the rollout alone (prefill + C tokens),
  (b) the side work alone: the reference pass's PROMPT chain (adapters off, 2180 rows through 28 layers; what a policy prompt
      chain or the next batch's frozen encoder would also look like: under-filled 144-workgroup MFMA grids),
  (c) both at once: the side work queued on a second stream before the token loop is issued,
      optionally (PROBE_PRIO=1) with the loop's stream at high priority.
Prints ms for each and the net effect (c) - (a) against (b): hidden = (a) + (b) - (c)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bioreason_amd import configs, ops
from bioreason_amd.dna_llm import DNALLMModel
from bioreason_amd.engine import BF16, SeqMeta
from bioreason_amd.synth import synth_prompt_batch

dev = torch.device("cuda:0")
C = int(os.environ.get("PROBE_C", "128"))
REPS = int(os.environ.get("PROBE_SIDE_REPS", "2"))          # prompt chains queued on the side stream (2 ~ reference + policy)
m = DNALLMModel(configs.qwen3_config(), configs.nt_v2_config(), device=dev)
m.text_model.init_weights(0.02, seed=1); m.dna_model.init_weights(0.02, seed=2)
m.text_model.apply_lora(r=32, alpha=64.0, arena=m.arena)
b = synth_prompt_batch(B=8, n_unique=1, dna_token_id=m.dna_token_id, device=dev)
kw = dict(input_ids=b["input_ids"], attention_mask=b["attention_mask"], dna_tokenized=b["dna_tokenized"], batch_idx_map=b["batch_idx_map"],
          dna_alias=b["dna_alias"], prompt_alias=b["prompt_alias"], do_sample=True, temperature=0.6, top_k=20, top_p=0.95, eos_token_id=None,
          use_graph=False, max_new_tokens=C)
eng = m.text_model.ensure_packed()
P = b["input_ids"].shape[1]
xp = (torch.randn(P, eng.H, device=dev) * 0.02).to(BF16)
mp = SeqMeta(B=1, S=P, pos=torch.arange(P, dtype=torch.int32, device=dev), kmask=torch.ones(1, P, dtype=torch.uint8, device=dev),
             lora_on=False, max_pos=P + C)


@torch.no_grad()
def side_work():
    for _ in range(REPS):
        x = xp
        for li in range(eng.L):
            kc = torch.empty((1, eng.Hkv, P, eng.hd), dtype=BF16, device=dev)
            vc = torch.empty_like(kc)
            x, _ = eng.layer_fwd(li, x, mp, save=False, kv_out=(kc, vc, 0))
    return x


def timed(fn):
    torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3


prio = os.environ.get("PROBE_PRIO", "0") == "1"
side = torch.cuda.Stream(device=dev, priority=0)
main_hi = torch.cuda.Stream(device=dev, priority=-1) if prio else None

def rollout():
    if main_hi is not None:
        main_hi.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(main_hi):
            m.generate(**kw)
        torch.cuda.current_stream(dev).wait_stream(main_hi)
    else:
        m.generate(**kw)

def both():
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        side_work()
    rollout()
    torch.cuda.current_stream(dev).wait_stream(side)

for it in range(3):
    a = timed(rollout)
    s = timed(side_work)
    c = timed(both)
    print(f"iter {it}: rollout alone {a:.1f} ms | side work alone ({REPS} prompt chains) {s:.1f} ms | both {c:.1f} ms | "
          f"hidden {a + s - c:.1f} ms of {s:.1f} (prio={int(prio)})", flush=True)
