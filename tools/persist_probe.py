"""Persistent decode step (k_persist.hip) against the launched step on the GPU: bit-equality of the logits of every decode step
and time per token.   python tools/persist_probe.py [layers] [tokens]
Model: Qwen3-1.7B dimensions with `layers` decoder layers (default 28) and random weights; 1 prompt x 8 rollouts, P = 2180."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bioreason_amd import configs, generation
from bioreason_amd import _lib as _bra_lib
_bra_lib.use_debug_library()          # knobs / probes / persistent step: libbioreason_hip_debug.so (include/bioreason_hip_debug.h)
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
out2, tr2 = run("0")
print("launched twice: identical", all(torch.equal(a, b) for a, b in zip(tr2, ref_tr)), flush=True)
for mode in [m_ for m_ in os.environ.get("PP_MODES", "1,2,3").split(",") if m_]:
    stop = "0"
    if ":" in mode:
        mode, stop = mode.split(":")
    os.environ["BRA_DEC_PERSIST_STOP"] = stop
    out, tr = run(mode)
    os.environ["BRA_DEC_PERSIST_STOP"] = "0"
    for st_, (a_, b_) in enumerate(zip(tr, ref_tr)):
        if not torch.equal(a_, b_):
            rows = (a_ != b_).any(dim=1).nonzero().flatten().tolist()
            print(f"   mode {mode} stop {stop}: first differing step {st_}, rows {rows}, max |d| {float((a_ - b_).abs().max()):.3e}", flush=True)
            break
    same_tok = bool(torch.equal(out, ref_out))
    nbad = [int((a != b).sum().item()) for a, b in zip(tr, ref_tr)]
    maxd = max(float((a - b).abs().max().item()) for a, b in zip(tr, ref_tr))
    print(f"persist mode {mode}: tokens equal {same_tok}; steps with differing logits {sum(1 for x in nbad if x)} of {len(nbad)}; "
          f"differing words per step (first 6) {nbad[:6]}; max |d| {maxd:.3e}", flush=True)
if os.environ.get("PP_TIME", "1") == "1":
    for mode in ["0"] + [m_ for m_ in [m_ for m_ in os.environ.get("PP_MODES", "1,2,3").split(",") if m_] if ":" not in m_]:
        print(f"mode {mode}: {timed(mode) * 1e3:.1f} us per token step (eager, {L} layers)", flush=True)
if os.environ.get("PP_STAMPS", "0") == "1":
    from bioreason_amd._lib import get_lib
    names = ["qkv", "items", "merge", "o", "gu", "down"]
    for mode in [m_ for m_ in os.environ.get("PP_MODES", "1,2,3").split(",") if m_]:
        if ":" in mode:
            continue
        st = torch.zeros(6 * L * 4, dtype=torch.int64, device=dev)
        get_lib().call("bra_persist_set_stamps", st)
        os.environ["BRA_DEC_PERSIST"] = mode
        generation.generate(m, emb, mask, **{**kw, "max_new_tokens": 6})
        torch.cuda.synchronize()
        get_lib().call("bra_persist_set_stamps", None)
        s4 = st.view(6 * L, 4).cpu().double() / 100.0          # us
        t0 = s4[0, 0]
        print(f"--- mode {mode}: stamps of workgroup 0 (last replayed token), us since the first phase started; per phase: start | work+stores issued | drained | next start")
        for l in (1, L // 2):
            for p_ in range(6):
                i = l * 6 + p_
                nxt = s4[i + 1, 0] if i + 1 < 6 * L else float("nan")
                print(f"   layer {l} {names[p_]:6s}: start {s4[i,0]-t0:8.2f}  work {s4[i,2]-s4[i,0]:6.2f}  drain {s4[i,3]-s4[i,2]:6.2f}  barrier {nxt-s4[i,3]:6.2f}  total {nxt-s4[i,0]:6.2f}")
        tot = (s4[6 * L - 1, 3] - t0)
        print(f"   whole layer loop: {tot:.1f} us; per layer {tot / L:.2f}")
