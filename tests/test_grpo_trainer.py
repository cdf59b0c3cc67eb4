"""SURVEY §8b boundary: the reference's import paths resolve to the HIP classes, `RepeatRandomSampler` equals the reference
class (ast-extracted from grpo_trainer.py:72-119 — the module itself needs trl), `DNALLMGRPOConfig` carries the reference's
fields and defaults, and `DNALLMGRPOTrainer(model, reward_funcs, args, dna_module, train_dataset, peft_config, ...)` runs the
text path of a training step (chat template -> DLProcessor -> rollout -> decode -> python rewards -> loss -> optimiser) with the
reference's rollout buffering over gradient accumulation and num_iterations (grpo_trainer.py:757-762)."""
import ast
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF_TRAINER = "/root/reference/bioreason/trainer/grpo_trainer.py"
REF_CONFIG = "/root/reference/bioreason/trainer/grpo_config.py"


# (the import-path test lives in tests/test_reference_imports.py: a fresh interpreter per run, the reference's own import lists)


SAMPLER_KNOWN = {   # (n, mini, batch, repeat, seed) -> stream of the reference class (recorded in the build container)
    (7, 2, 3, 2, 42): [1, 1, 6, 6, 3, 3, 1, 1, 6, 6, 3, 3, 5, 5, 4, 4, 0, 0, 5, 5, 4, 4, 0, 0],
    (4, 3, 1, 1, 0): [0, 0, 0, 1, 1, 1, 3, 3, 3, 2, 2, 2],
}


@pytest.mark.parametrize("key", list(SAMPLER_KNOWN))
def test_repeat_random_sampler_known_answers(key):
    from bioreason_amd.grpo_trainer import RepeatRandomSampler
    from bioreason_amd.grpo import repeat_sampler_indices
    n, mini, bs, rep, seed = key
    got = list(RepeatRandomSampler(range(n), mini, bs, rep, seed))
    assert got == SAMPLER_KNOWN[key]
    assert repeat_sampler_indices(n, mini, bs, rep, seed) == got


@pytest.mark.skipif(not os.path.exists(REF_TRAINER), reason="reference checkout not present")
def test_repeat_random_sampler_equals_reference_class():
    from typing import Optional, Sized
    from torch.utils.data import Sampler
    from bioreason_amd.grpo_trainer import RepeatRandomSampler
    tree = ast.parse(open(REF_TRAINER).read())
    node = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "RepeatRandomSampler")
    ns = {"Sampler": Sampler, "Sized": Sized, "Optional": Optional, "torch": torch}
    exec(compile(ast.Module(body=[node], type_ignores=[]), REF_TRAINER, "exec"), ns)
    Ref = ns["RepeatRandomSampler"]
    for n, mini, bs, rep, seed in [(7, 2, 3, 2, 42), (4, 3, 1, 1, 0), (16, 8, 2, 1, 42), (5, 1, 2, 3, 7), (3, 4, 4, 1, 1)]:
        want = list(Ref(range(n), mini, bs, rep, seed))
        assert list(RepeatRandomSampler(range(n), mini, bs, rep, seed)) == want
        assert len(RepeatRandomSampler(range(n), mini, bs, rep, seed)) == len(Ref(range(n), mini, bs, rep, seed))
        if (n, mini, bs, rep, seed) in SAMPLER_KNOWN:
            assert want == SAMPLER_KNOWN[(n, mini, bs, rep, seed)]


@pytest.mark.skipif(not os.path.exists(REF_CONFIG), reason="reference checkout not present")
def test_config_fields_and_defaults_equal_reference():
    """every GRPO field the reference's dataclass declares (name and default) exists on DNALLMGRPOConfig"""
    from bioreason_amd.grpo_trainer import DNALLMGRPOConfig
    tree = ast.parse(open(REF_CONFIG).read())
    cls = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "DNALLMGRPOConfig")
    ours = DNALLMGRPOConfig(output_dir="/tmp/_cfg", report_to="none")
    n = 0
    for st in cls.body:
        if isinstance(st, ast.AnnAssign) and isinstance(st.value, ast.Call):
            name = st.target.id
            default = None
            for kw in st.value.keywords:
                if kw.arg == "default":
                    default = ast.literal_eval(kw.value)
            if name == "report_to":      # TrainingArguments post-processes it
                continue
            assert hasattr(ours, name), name
            assert getattr(ours, name) == default, (name, getattr(ours, name), default)
            n += 1
    assert n >= 25


# ----------------------------------------------------------------------------------------------- end-to-end on the emulator
def _toy_tokenizers(tmp_path, vocab_size):
    from tokenizers import Tokenizer, models, pre_tokenizers
    from transformers import EsmTokenizer, GPT2TokenizerFast
    words = ["<|endoftext|>", "[UNK]", "<think>", "</think>", "answer", "Which", "pathway", "?", "ALS", "system", "user", "assistant",
             "<|im_start|>", "<|im_end|>"]
    vocab = {w: i for i, w in enumerate(words)}
    for i in range(len(vocab), vocab_size - 3):
        vocab[f"w{i}"] = i
    tk = Tokenizer(models.WordLevel(vocab, unk_token="[UNK]"))
    tk.pre_tokenizer = pre_tokenizers.WhitespaceSplit()
    tok = GPT2TokenizerFast(tokenizer_object=tk, eos_token="<|endoftext|>", unk_token="[UNK]")
    tok.add_special_tokens({"additional_special_tokens": ["<|dna_start|>", "<|dna_pad|>", "<|dna_end|>"]})
    vf = tmp_path / "vocab.txt"
    vf.write_text("\n".join(["<cls>", "<pad>", "<eos>", "<unk>", "A", "C", "G", "T", "N", "<mask>"]))
    return tok, EsmTokenizer(str(vf))


def test_trainer_text_path_with_buffering(backend, tmp_path):
    if backend.type != "cpu":
        pytest.skip("host-logic test: emulator run is enough")
    from bioreason.dna_modules import NucleotideDNAModule
    from bioreason.trainer import DNALLMGRPOConfig, DNALLMGRPOTrainer
    from bioreason_amd import configs, rewards
    from bioreason_amd.chat_template import CHAT_TEMPLATE
    from bioreason_amd.dna_llm import DNALLMModel
    tc = configs.qwen3_config(vocab_size=256, hidden_size=64, intermediate_size=128, num_hidden_layers=2, num_attention_heads=4,
                              num_key_value_heads=2, head_dim=32, max_position_embeddings=512)
    dc = configs.nt_v2_config(vocab_size=16, hidden_size=64, intermediate_size=128, num_hidden_layers=1, num_attention_heads=2,
                              max_position_embeddings=64)
    tok, dtok = _toy_tokenizers(tmp_path, 256)
    tok.chat_template = CHAT_TEMPLATE
    m = DNALLMModel(tc, dc, device=backend, dna_token_id=tok.convert_tokens_to_ids("<|dna_pad|>"))
    m.text_model.init_weights(0.05, seed=1)
    m.dna_model.init_weights(0.05, seed=2)
    m.text_tokenizer, m.dna_tokenizer, m.max_length_text, m.max_length_dna = tok, dtok, 64, 16
    data = [{"prompt": [{"role": "user", "content": [{"type": "dna"}, {"type": "text", "text": f"Which pathway w{20 + i} ?"}]}],
             "dna_sequences": ["ACGTAC" + "GT" * i], "answer": "ALS"} for i in range(4)]
    args = DNALLMGRPOConfig(output_dir=str(tmp_path / "out"), report_to="none", per_device_train_batch_size=2, num_generations=2,
                            max_completion_length=4, gradient_accumulation_steps=2, num_iterations=2, learning_rate=1e-3,
                            logging_steps=1, save_strategy="no", max_steps=4, seed=3, use_cpu=True)
    calls = []

    def spy_reward(prompts, completions, **kw):
        calls.append((len(prompts), [c[0]["content"] for c in completions], sorted(kw)))
        return [float(len(c[0]["content"]) % 3) for c in completions]

    tr = DNALLMGRPOTrainer(model=m, reward_funcs=[spy_reward, rewards.xmlcount_reward_func], args=args, dna_module=NucleotideDNAModule(),
                           train_dataset=data, peft_config={"r": 32, "lora_alpha": 64, "lora_dropout": 0.0})
    # sampler wiring (:883-897): effective batch = 2 x 1 rank x 2 accumulation = 4 rows = 2 unique prompts x G=2, x mu=2 repeats
    s = tr._get_train_sampler()
    assert (s.mini_repeat_count, s.batch_size, s.repeat_count) == (2, 2, 2)
    p0 = m.arena.params.clone()
    res = tr.train()
    assert res.global_step == 4
    # 4 optimiser steps x 2 micro-batches; rollouts are generated only on optimiser steps 0 and 2 (global_step % mu == 0)
    assert tr.runner._step == 8 and tr.runner.step_idx == 4, (tr.runner._step, tr.runner.step_idx)
    assert len(calls) == 4 and all(n == 2 and "answer" in kw and "dna_sequences" in kw for n, _, kw in calls)
    assert (m.arena.params - p0).abs().max() > 0
    assert len(tr.log_history) == 4
    for k in ("completion_length", "reward", "reward_std", "kl", "clip_ratio", "rewards/spy_reward", "rewards/xmlcount_reward_func", "loss"):
        assert k in tr.log_history[-1], k
    # the G copies of a prompt were recognised (shared prefill / encoder work)
    b = tr._prepare_batch([data[0], data[0]])
    assert b["prompt_alias"] == [0, 0] and b["dna_alias"] == [0, 0]
    assert (b["input_ids"] == m.dna_token_id).sum().item() == 2 * int((b["dna_tokenized"]["attention_mask"][0]).sum())
    tr.save_model(str(tmp_path / "ck"))
    sd = torch.load(str(tmp_path / "ck" / "pytorch_model.bin"), weights_only=True)
    assert any(k.startswith("dna_projection.") for k in sd) and any("lora_A.default.weight" in k for k in sd)


def test_old_logps_path_matches_oracle_math(backend):
    """num_iterations > 1: the clipped ratio uses the stored sampling-policy log-probs (grpo_trainer.py:620-626, :786)"""
    from oracle import grpo_math as GM
    from bioreason_amd import grpo
    dev = backend
    g = torch.Generator().manual_seed(0)
    B, C = 3, 7
    lp = (torch.randn(B, C, generator=g) * 0.3 - 1.0)
    old = lp + torch.randn(B, C, generator=g) * 0.4
    ref = lp + torch.randn(B, C, generator=g) * 0.2
    adv = torch.randn(B, generator=g)
    mask = (torch.rand(B, C, generator=g) > 0.2).int()
    mask[:, 0] = 1
    want, _, _ = GM.grpo_loss(lp, old, ref, adv, mask.float(), 0.2, 0.3, 0.04)
    got, stats = grpo.grpo_loss(lp.to(dev).requires_grad_(True), old.to(dev), ref.to(dev), adv.to(dev), mask.to(dev), 0.2, 0.3, 0.04)
    assert abs(got.item() - want.item()) < 1e-5


# ----------------------------------------------------------------------------------------------- host logic of the trainer shell
class _Args:
    """the TrainingArguments fields `_lr_schedule` reads"""
    def __init__(self, **kw):
        self.__dict__.update(kw)

    def get_warmup_steps(self, n):
        import math
        return self.warmup_steps if self.warmup_steps > 0 else math.ceil(n * self.warmup_ratio)


@pytest.mark.parametrize("kind,warmup_steps,warmup_ratio", [("linear", 0, 0.0), ("linear", 3, 0.0), ("cosine", 0, 0.1),
                                                            ("constant", 0, 0.0), ("constant_with_warmup", 2, 0.0)])
def test_lr_schedule_equals_hf_get_scheduler(kind, warmup_steps, warmup_ratio):
    """the rate handed to the fused AdamW at optimiser step s is the one HF Trainer's scheduler has in force at s
    (Trainer.create_scheduler -> get_scheduler; default `linear`: decay to 0 over the run, after warm-up)"""
    from transformers.optimization import get_scheduler
    from bioreason_amd.grpo_trainer import DNALLMGRPOTrainer
    a = _Args(learning_rate=2e-4, lr_scheduler_type=kind, warmup_steps=warmup_steps, warmup_ratio=warmup_ratio, lr_scheduler_kwargs={})
    n = 20
    sched = DNALLMGRPOTrainer._lr_schedule(type("T", (), {"args": a})(), n)
    opt = torch.optim.AdamW([torch.zeros(1, requires_grad=True)], lr=a.learning_rate)
    ref = get_scheduler(kind, optimizer=opt, num_warmup_steps=a.get_warmup_steps(n), num_training_steps=n)
    want = []
    for _ in range(n):
        want.append(opt.param_groups[0]["lr"])      # the rate the optimiser uses for this step
        opt.step()
        ref.step()
    got = [sched(s) for s in range(n)]
    assert got == pytest.approx(want, rel=0, abs=1e-12)
    assert sched(7) == got[7]                        # random access after the fact
    if kind == "linear" and warmup_steps == 0:
        assert got[0] == a.learning_rate and got[-1] == pytest.approx(a.learning_rate / n) and got[10] < got[5]
    if kind == "constant":
        assert all(g == a.learning_rate for g in got)


def test_sampler_generator_state_carries_across_epochs():
    """ONE sampler per run: iterating it again continues its generator (a new permutation per epoch, as the reference's
    dataloader does); re-creating it per epoch would replay the first permutation"""
    from bioreason_amd.grpo_trainer import RepeatRandomSampler
    s = RepeatRandomSampler(range(16), 1, 1, 1, seed=5)
    e0, e1 = list(s), list(s)
    assert sorted(e0) == sorted(e1) == list(range(16)) and e0 != e1
    assert list(RepeatRandomSampler(range(16), 1, 1, 1, seed=5)) == e0
    import inspect
    from bioreason_amd import grpo_trainer as GT
    src = inspect.getsource(GT.DNALLMGRPOTrainer.train)
    assert src.index("_get_train_sampler()") < src.index("for epoch in range")


def test_text_reward_fn_wraps_only_conversational_prompts():
    """grpo_trainer.py:643-650: `is_conversational(inputs[0])` decides whether completions are [{role, content}] or strings"""
    from bioreason_amd.trainer import text_reward_fn

    class Proc:
        def batch_decode(self, ids, skip_special_tokens=True):
            return [f"t{int(r[0])}" for r in ids]
    seen = []

    def rf(prompts, completions, **kw):
        seen.append(completions)
        return [1.0] * len(completions)
    ids = torch.tensor([[3, 1], [4, 1]])
    conv = [[{"role": "user", "content": "q"}]] * 2
    out = text_reward_fn(Proc(), [rf], prompts=conv)(ids, torch.ones_like(ids))
    assert seen[-1] == [[{"role": "assistant", "content": "t3"}], [{"role": "assistant", "content": "t4"}]] and out.shape == (2, 1)
    text_reward_fn(Proc(), [rf], prompts=["plain q", "plain q"])(ids, torch.ones_like(ids))
    assert seen[-1] == ["t3", "t4"]


def test_call_rc_compares_the_status_for_equality():
    from bioreason_amd._lib import BRA_ERR_UNSUPPORTED, KernelError, KernelLibrary
    lib = KernelLibrary.__new__(KernelLibrary)
    for status, raises in [(-2, False), (-20, True), (-1, True), (2, True)]:
        def call(name, *a, _s=status):
            raise KernelError(name, _s)
        lib.call = call
        if raises:
            with pytest.raises(KernelError) as ei:
                lib.call_rc("bra_x")
            assert ei.value.status == status
        else:
            assert lib.call_rc("bra_x") == BRA_ERR_UNSUPPORTED


def test_length_bucketed_sampler_is_the_reference_sampler_when_off_and_a_permutation_when_on():
    """SURVEY section 8f N4 (length-bucketed scheduling across ranks), opt-in: `bucket_batches = 1` or no costs reproduces
    RepeatRandomSampler index for index; with buckets every epoch is still a permutation (each index `mini x repeat` times, copies
    consecutive), every rank draws the same stream, and the spread of costs inside a global batch shrinks"""
    import random
    from bioreason_amd.grpo_trainer import LengthBucketedRepeatSampler, RepeatRandomSampler, prompt_cost
    n, mini, bs, rep, seed = 103, 4, 8, 2, 11
    rng = random.Random(3)
    costs = [rng.choice([200, 900, 2000, 4000]) + rng.random() for _ in range(n)]
    ref = list(RepeatRandomSampler(range(n), mini, bs, rep, seed))
    assert list(LengthBucketedRepeatSampler(range(n), mini, bs, rep, seed, costs=costs, bucket_batches=1)) == ref
    assert list(LengthBucketedRepeatSampler(range(n), mini, bs, rep, seed, costs=None, bucket_batches=6)) == ref
    a = list(LengthBucketedRepeatSampler(range(n), mini, bs, rep, seed, costs=costs, bucket_batches=6))
    b = list(LengthBucketedRepeatSampler(range(n), mini, bs, rep, seed, costs=costs, bucket_batches=6))
    assert a == b and len(a) == len(ref)                                          # same seed -> same stream on every rank
    from collections import Counter
    assert Counter(a) == Counter(ref) or set(a) <= set(range(n))
    cnt = Counter(a)
    assert all(v == mini * rep for v in cnt.values()) and len(cnt) == (n // bs) * bs
    for lo in range(0, len(a), bs * mini * rep):                                  # one global batch: bs unique prompts, each mini copies in a row, rep times
        blk = a[lo:lo + bs * mini * rep]
        uniq = blk[:bs * mini:mini]
        assert blk == [i for _ in range(rep) for i in uniq for _ in range(mini)]

    def spread(stream):
        out = []
        for lo in range(0, len(stream), bs * mini * rep):
            c = [costs[i] for i in stream[lo:lo + bs * mini:mini]]
            out.append(max(c) - min(c))
        return sum(out) / len(out)
    assert spread(a) < 0.5 * spread(ref)
    # the step waits for its slowest rank: mean over steps of max(cost) drops, the total work does not change
    def mean_max(stream):
        return sum(max(costs[i] for i in stream[lo:lo + bs * mini:mini]) for lo in range(0, len(stream), bs * mini * rep))
    assert mean_max(a) < 0.85 * mean_max(ref)
    s2 = LengthBucketedRepeatSampler(range(n), mini, bs, rep, seed, costs=costs, bucket_batches=6)
    e1, e2 = list(s2), list(s2)
    assert e1 != e2 and Counter(e1).keys() != set()                                # a new shuffle per epoch, like the reference's
    with pytest.raises(ValueError):
        s2.set_costs([1.0] * (n - 1))
    ex = {"dna_sequences": ["ACGT" * 10, "AC"], "prompt": [{"role": "user", "content": [{"type": "dna", "text": None}, {"type": "text", "text": "why?"}]}]}
    assert prompt_cost(ex) == 42 + 4 and prompt_cost({"prompt": "abc"}) == 3.0
