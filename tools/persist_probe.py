"""Persistent decode step (k_persist.hip) against the launched step on the GPU: bit-equality of the logits of every decode step
and time per token.   python tools/persist_probe.py [layers] [tokens]
Model: Qwen3-1.7B dimensions with `layers` decoder layers (default 28) and random weights; 1 prompt x 8 rollouts, P = 2180."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bioreason_amd import configs, generation
from bioreason_amd.modeling import Qwen3ForCausalLM

L = int(sys.argv[1]) if len(sys.argv) > 1 else 28
T = int(sys.argv[2]) if len(sys.argv) > 2 else 24
P = int(os.environ.get("PP_P", "2180"))
dev = torch.device("cuda:0")
m = Qwen3ForCausalLM(configs.qwen3_config(num_hidden_layers=L), device=dev)
m.init_weights(0.02, seed=1)
m.apply_lora(r=32, alpha=64.0, dropout=0.0)
g = torch.Generator().manual_seed(3)
for n, p in m.named_parameters():
    if "lora_B" in n:
        p.data.copy_((torch.randn(p.shape, generator=g) * 0.01).to(dev))
if getattr(m, "arena", None) is not None:
    m.arena.pack()
emb = (torch.randn(1, P, 2048, generator=g) * 0.02).to(torch.bfloat16).to(dev).repeat(8, 1, 1)
mask = torch.ones(8, P, dtype=torch.long, device=dev)
mask[:, :5] = 0
kw = dict(max_new_tokens=T, do_sample=True, temperature=0.6, top_k=20, top_p=0.95, eos_token_id=None, seed=11, prompt_alias=[0] * 8,
          use_graph=False)


def run(mode, trace=True):
    os.environ["BRA_DEC_PERSIST"] = mode
    tr = [] if trace else None
    out = generation.generate(m, emb, mask, trace_logits=tr, **kw)
    return out, tr


def timed(mode, n=3):
    os.environ["BRA_DEC_PERSIST"] = mode
    best = 1e9
    for _ in range(n):
        prof = {}
        generation.generate(m, emb, mask, profile=prof, **kw)
        best = min(best, prof["decode_loop"] / (T - 1))
    return best


ref_out, ref_tr = run("0")
print("launched: tokens", ref_out[0, :8].tolist(), flush=True)
for mode in os.environ.get("PP_MODES", "1,2,3").split(","):
    out, tr = run(mode)
    same_tok = bool(torch.equal(out, ref_out))
    nbad = [int((a != b).sum().item()) for a, b in zip(tr, ref_tr)]
    maxd = max(float((a - b).abs().max().item()) for a, b in zip(tr, ref_tr))
    print(f"persist mode {mode}: tokens equal {same_tok}; steps with differing logits {sum(1 for x in nbad if x)} of {len(nbad)}; "
          f"differing words per step (first 6) {nbad[:6]}; max |d| {maxd:.3e}", flush=True)
if os.environ.get("PP_TIME", "1") == "1":
    for mode in ["0"] + os.environ.get("PP_MODES", "1,2,3").split(","):
        print(f"mode {mode}: {timed(mode) * 1e3:.1f} us per token step (eager, {L} layers)", flush=True)
