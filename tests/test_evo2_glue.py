"""SURVEY §8f N3, the slice that needs no Evo2 oracle: the glue of `DNALLMModel.process_dna_embeddings` around an Evo2-interface
encoder (dna_llm.py:85-90, 123-146, 168) — tokenizer, batched call, and the left-padded valid-row quirk — against the REFERENCE'S
OWN `process_dna_embeddings` / `forward` executed on the same stand-in encoder.

The stand-in is NOT StripedHyena: a tiny causal network with Evo2's call signature (`model(input_ids, return_embeddings=True,
layer_names=[...]) -> (logits, {layer: [n, S, H]})`, `.tokenizer`, `.model.config.hidden_size`).  Causal on purpose: with left padding
the pad tokens precede the sequence and influence it, so a per-sequence call and a batched call only agree if the glue hands over the
padded rows unchanged — which is what the reference does (dna_llm.py:129: `input_ids[seq_idx:seq_idx+1]`)."""
import os
import sys
import types
import typing

import pytest
import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from bioreason_amd import configs                                   # noqa: E402
from bioreason_amd.dna_llm import DNALLMModel                       # noqa: E402
from bioreason_amd.evo2_tokenizer import CharLevelTokenizer, Evo2Tokenizer   # noqa: E402

LAYER = "blocks.2.mlp.l3"
HD = 32
GOLD = os.path.join(ROOT, "tests", "golden", "evo2_glue.pt")
HAVE_REF = os.path.isdir("/root/reference/bioreason")


class StandInEvo2(nn.Module):
    """Evo2's call interface over a toy causal mixer (embedding -> causal running mean -> tanh(linear))"""

    supports_batch = True

    def __init__(self, hidden=HD, seed=0):
        super().__init__()
        g = torch.Generator().manual_seed(seed)
        self.emb = nn.Embedding(512, hidden)
        self.lin = nn.Linear(hidden, hidden)
        with torch.no_grad():
            self.emb.weight.copy_(torch.randn(512, hidden, generator=g).to(torch.bfloat16).float())
            self.lin.weight.copy_((torch.randn(hidden, hidden, generator=g) * hidden ** -0.5).to(torch.bfloat16).float())
            self.lin.bias.zero_()
        self.tokenizer = CharLevelTokenizer(512)
        self.model = types.SimpleNamespace(config=types.SimpleNamespace(hidden_size=hidden))
        self.calls = []

    def forward(self, input_ids, return_embeddings=False, layer_names=None):
        self.calls.append(tuple(input_ids.shape))
        x = self.emb(input_ids.to(self.emb.weight.device))
        run = x.cumsum(1) / torch.arange(1, x.shape[1] + 1, device=x.device)[None, :, None]       # causal
        h = torch.tanh(self.lin(run))
        return None, {n: h for n in (layer_names or [])}


def tiny_text_cfg():
    return configs.qwen3_config(vocab_size=512, hidden_size=128, intermediate_size=256, num_hidden_layers=2, num_attention_heads=4,
                                num_key_value_heads=2, head_dim=32, rope_theta=1e6, max_position_embeddings=512)


def make_batch(dna_token_id=500):
    tok = Evo2Tokenizer(CharLevelTokenizer(512))
    seqs = ["ACGTACGTAC", "ACG", "TTGACA", "G"]                          # ragged -> left padding in every row but the longest
    enc = tok(seqs, padding=True, truncation=True, max_length=64, return_tensors="pt")
    lens = enc["attention_mask"].sum(1).tolist()
    B, P = 2, 40
    g = torch.Generator().manual_seed(3)
    ids = torch.randint(0, 400, (B, P), generator=g)
    ids[0, 5:5 + lens[0] + lens[1]] = dna_token_id
    ids[1, 2:2 + lens[2] + lens[3]] = dna_token_id
    mask = torch.ones(B, P, dtype=torch.long)
    mask[1, :2] = 0
    return {"input_ids": ids, "attention_mask": mask, "dna_tokenized": {k: v for k, v in enc.items()}, "batch_idx_map": [0, 0, 1, 1]}, lens


def build(dev, batched=True, dna_token_id=500):
    enc = StandInEvo2().to(dev)
    enc.supports_batch = batched
    m = DNALLMModel(tiny_text_cfg(), enc, device=dev, dna_is_evo2=True, dna_embedding_layer=LAYER, dna_token_id=dna_token_id)
    return m, enc


def to_dev(b, dev):
    return {"input_ids": b["input_ids"].to(dev), "attention_mask": b["attention_mask"].to(dev),
            "dna_tokenized": {k: v.to(dev) for k, v in b["dna_tokenized"].items()}, "batch_idx_map": list(b["batch_idx_map"])}


def rel(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


def test_constructor_wires_the_evo2_pieces(backend):
    m, enc = build(backend)
    assert m.dna_is_evo2 and m.dna_embedding_layer == LAYER and m.dna_model is enc
    assert isinstance(m.dna_tokenizer, Evo2Tokenizer) and m.dna_tokenizer.evo2_tokenizer is enc.tokenizer       # dna_llm.py:87
    assert m.dna_hidden_size == HD and m.dna_config.hidden_size == HD                                           # dna_llm.py:88,93
    assert m.dna_projection.in_features == HD and m.dna_projection.out_features == 128
    with pytest.raises(ImportError, match="evo2"):
        DNALLMModel(tiny_text_cfg(), "evo2_1b_base", device=backend, dna_is_evo2=True, dna_embedding_layer=LAYER)
    m2, _ = build(backend)
    m2.dna_embedding_layer = None
    b, _ = make_batch()
    with pytest.raises(ValueError, match="dna_embedding_layer"):
        m2.process_dna_embeddings(to_dev(b, backend)["dna_tokenized"], b["batch_idx_map"], 2)


def test_one_batched_call_equals_a_call_per_sequence(backend):
    b, lens = make_batch()
    bd = to_dev(b, backend)
    mb, eb = build(backend, batched=True)
    ms, es = build(backend, batched=False)
    got_b = mb.process_dna_embeddings(bd["dna_tokenized"], bd["batch_idx_map"], 2)
    got_s = ms.process_dna_embeddings(bd["dna_tokenized"], bd["batch_idx_map"], 2)
    assert eb.calls == [(4, 10)] and es.calls == [(1, 10)] * 4                     # dna_llm.py:127-141 issues four
    for x, y in zip(got_b, got_s):
        assert torch.equal(x.cpu(), y.cpu())
    assert [t.shape[0] for t in got_b] == [lens[0] + lens[1], lens[2] + lens[3]]
    empty = {"input_ids": torch.zeros((0, 10), dtype=torch.long), "attention_mask": torch.zeros((0, 10), dtype=torch.long)}
    out = mb.process_dna_embeddings(empty, [], 3)                                  # dna_llm.py:145-146
    assert [tuple(t.shape) for t in out] == [(0, 128)] * 3


def test_left_padded_rows_follow_the_reference_quirk(backend):
    """dna_llm.py:168 keeps rows [0, attention_mask.sum()) of every sequence; under LEFT padding (evo2_tokenizer.py:138-146) those
    are the pad rows first — the reference's behaviour, kept: the rows written over the <|dna_pad|> positions are exactly those"""
    b, lens = make_batch()
    bd = to_dev(b, backend)
    m, enc = build(backend)
    per_item = m.process_dna_embeddings(bd["dna_tokenized"], bd["batch_idx_map"], 2)
    with torch.no_grad():
        _, emb = enc(bd["dna_tokenized"]["input_ids"], return_embeddings=True, layer_names=[LAYER])
        proj = emb[LAYER].float() @ m.dna_projection.weight.float().T + m.dna_projection.bias.float()
    want0 = torch.cat([proj[0, :lens[0]], proj[1, :lens[1]]])                      # rows from position 0, pads included
    assert rel(per_item[0], want0) < 1e-2
    valid1 = proj[1, 10 - lens[1]:]                                                # what a mask-aware gather would have taken
    assert rel(per_item[0][lens[0]:], valid1) > 0.1                                # ... and is NOT what the reference takes
    emb_rows = m._inputs_embeds(bd["input_ids"], bd["dna_tokenized"], bd["batch_idx_map"])
    pos = (bd["input_ids"] == 500)
    assert rel(emb_rows[0][pos[0]], per_item[0]) < 1e-6 and rel(emb_rows[1][pos[1]], per_item[1]) < 1e-6


def _golden():
    return torch.load(GOLD, weights_only=False)


def test_logits_equal_the_reference_glue_golden(backend):
    """the reference's own DNALLMModel.forward (dna_is_evo2=True branch) on the stand-in encoder, recorded by
    oracle/make_evo2_glue_golden.py: per-item embeddings and logits"""
    fix = _golden()
    m, enc = build(backend)
    enc.load_state_dict(fix["encoder"])
    m.text_model.load_state_dict(fix["text"], strict=False)
    m.dna_projection.weight.data.copy_(fix["proj"]["weight"].float())
    m.dna_projection.bias.data.copy_(fix["proj"]["bias"].float())
    m.arena.pack()
    bd = to_dev(fix["batch"], backend)
    per_item = m.process_dna_embeddings(bd["dna_tokenized"], bd["batch_idx_map"], 2)
    for got, want in zip(per_item, fix["per_item"]):
        assert tuple(got.shape) == tuple(want.shape) and rel(got, want) < 1e-2
    out = m(**bd)
    keep = fix["batch"]["attention_mask"].bool()
    e_hip, e_ref = rel(out.logits.float().cpu()[keep], fix["logits_fp32"][keep]), rel(fix["logits_bf16"][keep], fix["logits_fp32"][keep])
    assert e_hip <= (1.25 if backend.type == "cuda" else 1.6) * e_ref, (e_hip, e_ref)


@pytest.mark.skipif(not HAVE_REF, reason="reference checkout not present")
def test_golden_is_what_the_reference_class_returns_today():
    from oracle import make_evo2_glue_golden as G
    fresh, fix = G.compute(), _golden()
    for a, b in zip(fresh["per_item"], fix["per_item"]):
        assert torch.equal(a, b)
    assert torch.equal(fresh["logits_fp32"], fix["logits_fp32"])
