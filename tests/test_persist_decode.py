"""The persistent decode step (k_persist.hip: all decoder layers of a token step in ONE launch, grid barriers + write-through
hand-offs between its phases, weights of the following phases requested ahead) against the launched step it replaces
(bra_qwen_decode_step_one: six launches per layer).  Same tiles, K split, reduction order and epilogues: the fp32 logits of every
decode step must be BIT-IDENTICAL, for every prefetch level, across a 64-key chunk boundary of the completion cache, with a
left-padded prompt.  GPU only (the emulator runs workgroups one after another: nothing to synchronise)."""
import os

import pytest
import torch


def _setup(dev, layers, vocab, P, pad):
    from bioreason_amd import configs
    from bioreason_amd.modeling import Qwen3ForCausalLM
    m = Qwen3ForCausalLM(configs.qwen3_config(num_hidden_layers=layers, vocab_size=vocab), device=dev)      # Qwen3-1.7B dimensions
    m.init_weights(0.02, seed=1)
    m.apply_lora(r=32, alpha=64.0, dropout=0.0)
    g = torch.Generator().manual_seed(3)
    for n, p in m.named_parameters():
        if "lora_B" in n:
            p.data.copy_((torch.randn(p.shape, generator=g) * 0.01).to(dev))
    m.arena.pack()
    emb = (torch.randn(1, P, 2048, generator=g) * 0.02).to(torch.bfloat16).to(dev)
    mask = torch.ones(1, P, dtype=torch.long, device=dev)
    mask[:, :pad] = 0
    return m, emb, mask


@pytest.mark.gpu
@pytest.mark.parametrize("copies,P,T", [(8, 333, 70), (5, 130, 12)])
def test_persistent_step_is_bit_identical_to_launched_step(hip_debug_device, copies, P, T, monkeypatch):
    hip_device = hip_debug_device
    from bioreason_amd import generation
    dev = hip_device
    if torch.cuda.get_device_properties(0).multi_processor_count < 256:
        pytest.skip("needs one CU per workgroup (256)")
    m, emb, mask = _setup(dev, 2, 8192, P, 7)
    emb, mask = emb.repeat(copies, 1, 1), mask.repeat(copies, 1)
    kw = dict(max_new_tokens=T, do_sample=True, temperature=0.6, top_k=20, top_p=0.95, eos_token_id=None, seed=11,
              prompt_alias=[0] * copies, use_graph=False)
    runs = {}
    for mode in ("0", "1", "2", "3"):
        monkeypatch.setenv("BRA_DEC_PERSIST", mode)
        tr = []
        out = generation.generate(m, emb, mask, trace_logits=tr, **kw)
        assert len(tr) == T - 1
        runs[mode] = (out, tr)
    ref_out, ref_tr = runs["0"]
    for mode in ("1", "2", "3"):
        out, tr = runs[mode]
        for step, (a, b) in enumerate(zip(tr, ref_tr)):
            assert torch.equal(a, b), f"prefetch level {int(mode) - 1}: logits differ at decode step {step} ({int((a != b).sum())} words)"
        assert torch.equal(out, ref_out)
    # the persistent path really ran (an unsupported shape falls back silently to the launched kernels)
    monkeypatch.setenv("BRA_DEC_PERSIST", "2")
    st = {}
    orig = generation.SharedDecodeState.step

    def spy(self, *a, **k):
        r = orig(self, *a, **k)
        st["ok"] = self.persist["ok"] if self.persist is not None else None
        return r
    monkeypatch.setattr(generation.SharedDecodeState, "step", spy)
    generation.generate(m, emb, mask, **{**kw, "max_new_tokens": 3})
    assert st["ok"] is True


@pytest.mark.gpu
def test_persistent_step_under_graph_replay(hip_debug_device, monkeypatch):
    hip_device = hip_debug_device
    """the token loop replayed from a hipGraph (memset node of the barrier record + the persistent launch + lm_head + sampler):
    same tokens as eager issue of the launched kernels"""
    from bioreason_amd import generation
    if torch.cuda.get_device_properties(0).multi_processor_count < 256:
        pytest.skip("needs one CU per workgroup (256)")
    m, emb, mask = _setup(hip_device, 2, 8192, 200, 3)
    emb, mask = emb.repeat(8, 1, 1), mask.repeat(8, 1)
    kw = dict(max_new_tokens=40, do_sample=True, temperature=0.6, top_k=20, top_p=0.95, eos_token_id=None, seed=5, prompt_alias=[0] * 8)
    monkeypatch.setenv("BRA_DEC_PERSIST", "0")
    want = generation.generate(m, emb, mask, use_graph=False, **kw)
    monkeypatch.setenv("BRA_DEC_PERSIST", "2")
    got = generation.generate(m, emb, mask, use_graph=True, **kw)
    assert torch.equal(got, want)
