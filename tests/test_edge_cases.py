"""Edge cases of the reference's forward / generate that the golden fixtures do not contain, checked directly against
the CPU oracle rebuilt from the fixture weights (fp32, eager attention): text-only batches (dna_llm.py:208-211 — no DNA
branch), a batch where one sample has no DNA sequence (ragged `batch_idx_map`, dna_llm.py:163-177), and the EOS /
padding bookkeeping of `generate` (TF:generation/utils.py:2897-2925: finished rows emit pad, the loop stops when every
row has finished)."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from test_model_parity import GOLD, build, rel, to_dev   # noqa: E402
from test_oracle import rebuild                          # noqa: E402


def _fix(name):
    return torch.load(os.path.join(GOLD, f"{name}.pt"), weights_only=False)


def rel_valid(got, want, mask):
    """relative distance over the attended positions: the rows of left-padding queries see no key at all and are
    unspecified in the reference (its eager softmax of an all-masked row), as in test_model_parity"""
    w = mask.bool().cpu()[..., None]
    return rel(got.float().cpu() * w, want.float().cpu() * w)


@pytest.mark.parametrize("name", ["tiny_a", "tiny_b"])
def test_text_only_batch(backend, name):
    fix = _fix(name)
    cfg = fix["config"]
    ora = rebuild(fix, True)
    m = build(fix, backend, True)
    ids = fix["batch"]["input_ids"].clone()
    ids[ids == cfg["dna_token_id"]] = 5                     # plain text: no placeholder rows at all
    mask = fix["batch"]["attention_mask"]
    labels = fix["batch"]["labels"]
    with torch.no_grad():
        want = ora(input_ids=ids, attention_mask=mask, labels=labels)
    got = m(input_ids=ids.to(backend), attention_mask=mask.to(backend), labels=labels.to(backend))
    assert rel_valid(got.logits, want.logits, mask) < 2e-2
    assert abs(got.loss.item() - want.loss.item()) < 3e-2 * max(1.0, abs(want.loss.item()))
    # and with the DNA arguments present but empty (reference: `if dna_tokenized is not None and batch_idx_map`)
    got2 = m(input_ids=ids.to(backend), attention_mask=mask.to(backend), dna_tokenized=None, batch_idx_map=[])
    assert rel_valid(got2.logits, got.logits, mask) == 0


def test_sample_without_dna_among_samples_with_dna(backend):
    """tiny_a holds two DNA sequences per sample; drop the second sample's sequences: its placeholder rows become text"""
    fix = _fix("tiny_a")
    cfg = fix["config"]
    b = fix["batch"]
    bim = list(b["batch_idx_map"])
    keep = [i for i, s in enumerate(bim) if s == 0]
    assert keep and len(keep) < len(bim)
    ids = b["input_ids"].clone()
    ids[1][ids[1] == cfg["dna_token_id"]] = 7
    dna = {k: v[keep] for k, v in b["dna_tokenized"].items()}
    ora = rebuild(fix, False)
    with torch.no_grad():
        want = ora(input_ids=ids, attention_mask=b["attention_mask"], dna_tokenized=dna, batch_idx_map=[0] * len(keep))
    m = build(fix, backend, False)
    got = m(input_ids=ids.to(backend), attention_mask=b["attention_mask"].to(backend),
            dna_tokenized={k: v.to(backend) for k, v in dna.items()}, batch_idx_map=[0] * len(keep))
    assert rel_valid(got.logits, want.logits, b["attention_mask"]) < 2e-2


@pytest.mark.parametrize("decode_impl", ["fused", "unfused"])
@pytest.mark.parametrize("row,step", [(0, 7), (1, 2)])
def test_generate_eos_and_padding(backend, decode_impl, row, step):
    """Pick as EOS the token the reference's greedy decode emits for `row` first at `step` (tiny_a with adapters:
    row 0 = 450 x7 then 42.., row 1 = 13 13 170..): that row must stop there and be padded, the other row continues,
    exactly as HF's unfinished_sequences bookkeeping does."""
    fix = _fix("tiny_a")
    cfg = fix["config"]
    ref_ids = fix["fp32_lora"]["greedy_ids"]
    eos = int(ref_ids[row, step])
    assert eos not in ref_ids[row, :step].tolist() and eos not in ref_ids[1 - row].tolist()
    pad = 1
    ora = rebuild(fix, True)
    b = fix["batch"]
    gb = {k: v for k, v in b.items() if k != "labels"}
    want = ora.generate(**gb, max_new_tokens=cfg["gen_tokens"], do_sample=False, eos_token_id=eos, pad_token_id=pad)
    m = build(fix, backend, True)
    d = to_dev(b, backend)
    d.pop("labels")
    got = m.generate(**d, max_new_tokens=cfg["gen_tokens"], do_sample=False, eos_token_id=eos, pad_token_id=pad,
                     check_every=1, decode_impl=decode_impl)
    assert got.shape == want.shape, (got.shape, want.shape)
    g = got.cpu()
    # the reference's own switch to this token is a near-tie (row 0: 450 -> 42), so the bf16 path may reach it one step
    # earlier or later (margins are checked in test_greedy_decode_and_logps); the bookkeeping must hold wherever it lands
    hits = (g[row] == eos).nonzero().flatten().tolist()
    assert len(hits) == 1 and abs(hits[0] - step) <= 2, (hits, g[row].tolist())
    assert (g[row, hits[0] + 1:] == pad).all() and (g[row, :hits[0]] != pad).all()
    assert eos not in g[1 - row].tolist() and pad not in g[1 - row].tolist()
    assert int((g[1 - row] != want[1 - row]).sum()) <= 2


class _FixedMask(torch.nn.Module):
    """stands in for a LoraLayer's nn.Dropout with a given keep mask: x * mask / (1 - p)"""

    def __init__(self, mask, p):
        super().__init__()
        self.mask, self.p = mask, p

    def forward(self, x):
        return x * self.mask.to(x.dtype).view(x.shape) / (1.0 - self.p)


@pytest.mark.parametrize("name", ["tiny_a", "tiny_b"])
def test_lora_dropout_matches_oracle_under_the_same_masks(backend, name):
    """training-mode LoRA dropout (reason.py:266: lora_dropout 0.05; here 0.2 for signal): every target module has its own
    mask stream, as PEFT's per-layer nn.Dropout; the masks the HIP kernels regenerate on the fly are exported and injected
    into the oracle, whose loss, logits and LoRA / projection gradients must then agree as in test_forward_backward"""
    from bioreason_amd import ops
    from bioreason_amd.engine import lora_drop_seeds
    from oracle import dna_llm_oracle as O
    p = 0.2
    fix = _fix(name)
    b = fix["batch"]
    m = build(fix, backend, False)
    m.text_model.apply_lora(r=32, alpha=64.0, dropout=p, arena=m.arena)
    own = dict(m.text_model.named_parameters())
    for k, v in fix["state"]["lora"].items():
        own[k].data.copy_(v.to(backend))
    m.arena.pack()
    m.train()
    m.text_model.set_dropout_seed(123)
    pass_seed = (123 * 0x9E3779B1 + 1 * 0x85EBCA6B) & 0xFFFFFFFF            # first forward after set_dropout_seed
    ora = rebuild(fix, True)
    B, S = b["input_ids"].shape
    where = {"q_proj": ("qkv", 0, 3), "k_proj": ("qkv", 1, 3), "v_proj": ("qkv", 2, 3), "o_proj": ("o", 0, 1),
             "gate_proj": ("gu", 0, 2), "up_proj": ("gu", 1, 2), "down_proj": ("d", 0, 1)}
    for li, layer in enumerate(ora.text_model.model.layers):
        for holder in (layer.self_attn, layer.mlp):
            for nm, (grp, j, n) in where.items():
                mod = getattr(holder, nm, None)
                if isinstance(mod, O.LoraLinear):
                    K = mod.base_layer.in_features
                    seed = lora_drop_seeds(pass_seed, li, grp, n)[j]
                    mod.dropout = _FixedMask(ops.dropout_mask(B * S, K, p, seed, backend).cpu().view(B, S, K), p)
    want = ora(**{k: (v.clone() if torch.is_tensor(v) else v) for k, v in b.items()})
    want.loss.backward()
    m.arena.zero_grad()
    got = m(**to_dev(b, backend))
    keep = b["attention_mask"].bool()
    tol = 2.5e-2
    assert rel(got.logits.float().cpu()[keep], want.logits.detach()[keep]) < tol
    assert abs(got.loss.item() - want.loss.item()) < tol * max(1.0, abs(want.loss.item()))
    got.loss.backward()
    assert rel(m.dna_projection.weight.grad, ora.dna_projection.weight.grad) < 3 * tol
    l0, r0 = m.text_model.model.layers[0], ora.text_model.model.layers[0]
    for nm, mod, ref in (("q", l0.self_attn.q_proj, r0.self_attn.q_proj), ("k", l0.self_attn.k_proj, r0.self_attn.k_proj),
                         ("up", l0.mlp.up_proj, r0.mlp.up_proj), ("down", l0.mlp.down_proj, r0.mlp.down_proj)):
        assert rel(mod.lora_A["default"].weight.grad, ref.lora_A["default"].weight.grad) < 3 * tol, nm
        assert rel(mod.lora_B["default"].weight.grad, ref.lora_B["default"].weight.grad) < 3 * tol, nm
    # the masks matter: without them the oracle's gradients are measurably different (guards against a silent no-op)
    ora2 = rebuild(fix, True)
    w2 = ora2(**{k: (v.clone() if torch.is_tensor(v) else v) for k, v in b.items()})
    w2.loss.backward()
    assert rel(r0.mlp.down_proj.lora_A["default"].weight.grad, ora2.text_model.model.layers[0].mlp.down_proj.lora_A["default"].weight.grad) > 0.1
    # eval mode: no dropout (nn.Dropout is the identity)
    m.eval()
    ev = m(**to_dev(b, backend))
    assert rel(ev.logits.float().cpu()[keep], w2.logits.detach()[keep]) < tol


def test_dna_embedding_cache_is_exact(backend):
    """encoder outputs cached per distinct sequence reproduce the uncached forward bit for bit — the rows
    of a batched encoder pass do not depend on which other sequences share the batch — and repeated calls hit the cache"""
    fix = _fix("tiny_a")
    m = build(fix, backend, False)
    b = to_dev(fix["batch"], backend)
    b.pop("labels")
    want = m(**b).logits
    m.enable_dna_cache(max_entries=8)
    got1 = m(**b).logits
    n = b["dna_tokenized"]["input_ids"].shape[0]
    assert m.dna_cache_misses == n and m.dna_cache_hits == 0
    got2 = m(**b).logits
    assert m.dna_cache_misses == n and m.dna_cache_hits == n           # nothing recomputed
    assert torch.equal(got1.cpu(), want.cpu()) and torch.equal(got2.cpu(), want.cpu())
    m._dna_cache_cap = 2                                               # least-recently-used rows are dropped
    assert torch.equal(m(**b).logits.cpu(), want.cpu()) and len(m._dna_cache) == 2
    # a different order / subset of the same sequences is served from the cache where possible, still exact
    keep = [i for i, s in enumerate(b["batch_idx_map"]) if s == 0]
    sub = {k: v[keep] for k, v in b["dna_tokenized"].items()}
    ids = b["input_ids"].clone()
    ids[1][ids[1] == fix["config"]["dna_token_id"]] = 7
    m.disable_dna_cache()
    w2 = m(input_ids=ids, attention_mask=b["attention_mask"], dna_tokenized=sub, batch_idx_map=[0] * len(keep)).logits
    m.enable_dna_cache()
    m(**b)
    g2 = m(input_ids=ids, attention_mask=b["attention_mask"], dna_tokenized=sub, batch_idx_map=[0] * len(keep)).logits
    assert torch.equal(g2.cpu(), w2.cpu())


def test_batch_without_a_supervised_position(backend):
    """every label -100 (the assistant span fell to truncation, kegg.py:252-327 + max_length_text): HF's CE mean over zero
    positions is NaN (TF:loss/loss_utils.py:32-46) — so is this loss, and its backward runs (zero gradients) instead of raising"""
    fix = _fix("tiny_b")
    ora = rebuild(fix, True)
    m = build(fix, backend, True)
    b = to_dev(fix["batch"], backend)
    lab = torch.full_like(fix["batch"]["labels"], -100)
    with torch.no_grad():
        want = ora(input_ids=fix["batch"]["input_ids"], attention_mask=fix["batch"]["attention_mask"], labels=lab,
                   dna_tokenized=fix["batch"]["dna_tokenized"], batch_idx_map=fix["batch"]["batch_idx_map"])
    assert torch.isnan(want.loss)
    m.train()
    m.arena.zero_grad()
    out = m(input_ids=b["input_ids"], attention_mask=b["attention_mask"], labels=lab.to(backend), dna_tokenized=b["dna_tokenized"],
            batch_idx_map=b["batch_idx_map"])
    assert torch.isnan(out.loss) and out.logits.shape == want.logits.shape
    out.loss.backward()
    assert float(m.arena.grads.abs().max()) == 0.0
    # one supervised position is an ordinary batch again
    lab[0, -1] = fix["batch"]["input_ids"][0, -1]
    with torch.no_grad():
        want = ora(input_ids=fix["batch"]["input_ids"], attention_mask=fix["batch"]["attention_mask"], labels=lab,
                   dna_tokenized=fix["batch"]["dna_tokenized"], batch_idx_map=fix["batch"]["batch_idx_map"])
    out = m(input_ids=b["input_ids"], attention_mask=b["attention_mask"], labels=lab.to(backend), dna_tokenized=b["dna_tokenized"],
            batch_idx_map=b["batch_idx_map"])
    assert abs(out.loss.item() - want.loss.item()) < 3e-2 * max(1.0, abs(want.loss.item()))
