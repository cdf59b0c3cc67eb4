"""Data-parallel GRPO step on 2 ranks (gloo, CPU, kernel-source emulator): the two exchange steps of the path — the
reward all-gather (grpo_trainer.py:679) and the gradient reduction (one flat all-reduce over the TrainableArena,
replacing DDP / ZeRO-2 buckets) — keep the replicas identical and use the gathered rewards for the group statistics."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


def _worker(rank, world, port, emu_path, q, grouped=False):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["BRA_EMU_THREADS"] = "2"
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from bioreason_amd import _lib
    _lib.use_library_for_tests(emu_path)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_model_parity import build, to_dev
    from bioreason_amd.trainer import GRPOConfig, GRPOStepRunner
    fix = torch.load(os.path.join(GOLD, "tiny_a.pt"), weights_only=False)
    m = build(fix, torch.device("cpu"), True)
    b = to_dev(fix["batch"], torch.device("cpu"))
    b.pop("labels")
    if grouped:
        # RepeatRandomSampler's layout: consecutive copies of a prompt -> the shared-prompt rollout, reference pass and (differentiable)
        # policy pass; rank r trains on 2 copies of prompt r of the fixture
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from test_shared_policy import _group_batch
        ids, mask, mm, alias = _group_batch(fix, torch.device("cpu"), 2)
        rows = [2 * rank, 2 * rank + 1]
        dsel = [i for i, s_ in enumerate(mm["batch_idx_map"]) if s_ in rows]
        b = {"input_ids": ids[rows], "attention_mask": mask[rows],
             "dna_tokenized": {k: v[dsel] for k, v in mm["dna_tokenized"].items()},
             "batch_idx_map": [mm["batch_idx_map"][i] - rows[0] for i in dsel], "prompt_alias": [0, 0]}
    seen = {}

    def reward(ids, mask):
        r = torch.stack([(ids[:, 0] % 5).float() + rank, (ids[:, 1] % 3).float()], dim=1)
        seen["local"] = r.clone()
        return r

    G = 2 if grouped else 2 * world          # (ungrouped) one group spans both ranks: 2 local rows per rank, G = 4 (exercises the cross-rank gather)
    gdt = os.environ.get("BRA_TEST_GRAD_DTYPE", "fp32")
    runner = GRPOStepRunner(m, GRPOConfig(num_generations=G, max_completion_length=4, eos_token_id=None, seed=7, learning_rate=1e-3,
                                          grad_allreduce_dtype=gdt), reward)
    p0 = m.arena.params.clone()
    out = runner.step(b)
    if gdt == "bf16":        # every summed gradient went through a bf16 image (the fp32 metric slot excepted)
        from bioreason_amd.trainer import METRIC_SLOT
        g_ = m.arena.grads.clone()
        o_ = m.arena._offsets[METRIC_SLOT]
        g_[o_:o_ + 64] = 0
        assert torch.equal(g_, g_.to(torch.bfloat16).float()) and float(g_.abs().max()) > 0
    gathered = [torch.empty_like(seen["local"]) for _ in range(world)]
    dist.all_gather(gathered, seen["local"])
    allr = torch.cat(gathered, 0).sum(1)
    if grouped:
        grp = allr.view(-1, 2)
        want_adv = ((allr - grp.mean(1).repeat_interleave(2)) / (grp.std(1).repeat_interleave(2) + 1e-4))[rank * 2:(rank + 1) * 2]
    else:
        mean, std = allr.mean(), allr.std()
        want_adv = ((allr - mean) / (std + 1e-4))[rank * 2:(rank + 1) * 2]
    params = [torch.empty_like(m.arena.params) for _ in range(world)]
    dist.all_gather(params, m.arena.params)
    grads = [torch.empty_like(m.arena.grads) for _ in range(world)]
    dist.all_gather(grads, m.arena.grads)
    got_adv = runner._buffered_inputs[0]["advantages"]
    # the packed metric record: every rank must hold the same global means (completion length, reward, group std,
    # and loss / KL / clip ratio averaged over ranks through the gradient bucket's spare slot)
    mets = [torch.empty_like(out["metrics_t"]) for _ in range(world)]
    dist.all_gather(mets, out["metrics_t"])
    losses = [torch.empty_like(out["loss_t"].reshape(1)) for _ in range(world)]
    dist.all_gather(losses, out["loss_t"].reshape(1))
    want_loss = float(torch.cat(losses).mean())
    q.put((rank, bool(torch.equal(params[0], params[1])), bool(torch.equal(grads[0], grads[1])),
           float((m.arena.params - p0).abs().max()), float(out["loss_t"]), want_adv.tolist(), got_adv.tolist(),
           bool(torch.equal(mets[0], mets[1])), out["metrics_t"].tolist(), want_loss, float(allr.mean()), len(runner._cuts)))
    dist.destroy_process_group()


def test_grpo_step_two_ranks(emu_lib_path):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, emu_lib_path, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=900) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, same_p, same_g, moved, loss, want_adv, got_adv, same_m, mets, want_loss, want_reward, ncuts in res:
        assert same_p, "replicas diverged after the optimiser step"
        assert same_g, "gradient bucket differs across ranks after the all-reduce"
        assert moved > 0, "parameters did not move"
        assert loss == loss
        # the advantages every rank trains on = its slice of the statistics of the GATHERED rewards (grpo_trainer.py:679-699)
        assert torch.allclose(torch.tensor(got_adv), torch.tensor(want_adv), atol=1e-5), (rank, got_adv, want_adv)
        assert same_m, "metric record differs across ranks"
        # metric_names = completion_length, reward, reward_std, loss, kl, clip_ratio
        assert abs(mets[0] - 4.0) < 1e-6 and abs(mets[1] - want_reward) < 1e-5
        assert abs(mets[3] - want_loss) < 1e-5 * max(1.0, abs(want_loss))
        assert ncuts >= 1, "the gradient reduction was not cut into overlapped buckets"


def test_grpo_step_two_ranks_bf16_gradient_transport(emu_lib_path, monkeypatch):
    """GRPOConfig.grad_allreduce_dtype = "bf16": each bucket travels as a bf16 image (half the xGMI bytes), the per-rank loss / KL / clip
    ratio in the spare slot stay fp32; replicas identical, metrics exact, the bucket cuts unchanged"""
    monkeypatch.setenv("BRA_TEST_GRAD_DTYPE", "bf16")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 36500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, emu_lib_path, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=900) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, same_p, same_g, moved, loss, want_adv, got_adv, same_m, mets, want_loss, want_reward, ncuts in res:
        assert same_p and same_g and same_m and moved > 0 and loss == loss
        assert abs(mets[3] - want_loss) < 1e-5 * max(1.0, abs(want_loss)) and ncuts >= 1


def test_grpo_step_two_ranks_shared_prompt_groups(emu_lib_path):
    """the same step on the layout the bench and the trainer produce: every rank holds consecutive copies of its prompt, so the
    rollout, the reference pass and the differentiable policy pass all run the prompt once per group — replicas identical after
    the bucketed gradient mean, advantages = statistics of the gathered rewards per group"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, emu_lib_path, q, True)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=900) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, same_p, same_g, moved, loss, want_adv, got_adv, same_m, mets, want_loss, want_reward, ncuts in res:
        assert same_p and same_g and same_m and moved > 0 and loss == loss
        assert torch.allclose(torch.tensor(got_adv), torch.tensor(want_adv), atol=1e-5), (rank, got_adv, want_adv)
        assert abs(mets[3] - want_loss) < 1e-5 * max(1.0, abs(want_loss)) and ncuts >= 1


def _worker_uneven(rank, world, port, emu_path, q):
    """4 ranks x 3 local rows, G = 4: the three groups of the global batch span the ranks UNEVENLY (rows 0-3 = rank 0 + one row of
    rank 1; rows 4-7 = two of rank 1 + two of rank 2; rows 8-11 = one of rank 2 + rank 3) — the layout accelerate's contiguous
    per-rank slices of RepeatRandomSampler's stream give when per_device_train_batch_size is not a multiple of G
    (grpo_trainer.py:428-437 only asks (W x B_local) % G == 0; the advantages of :682-699 use the GATHERED rewards)"""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), BRA_EMU_THREADS="1")
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from bioreason_amd import _lib
    _lib.use_library_for_tests(emu_path)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_model_parity import build, to_dev
    from bioreason_amd.trainer import GRPOConfig, GRPOStepRunner
    fix = torch.load(os.path.join(GOLD, "tiny_a.pt"), weights_only=False)
    m = build(fix, torch.device("cpu"), True)
    b0 = to_dev(fix["batch"], torch.device("cpu"))
    b0.pop("labels")
    rows = [(rank + j) % 2 for j in range(3)]                       # three local rows drawn from the fixture's two samples
    dsel, bmap = [], []
    for j, r in enumerate(rows):
        for i, s_ in enumerate(b0["batch_idx_map"]):
            if s_ == r:
                dsel.append(i)
                bmap.append(j)
    b = {"input_ids": b0["input_ids"][rows], "attention_mask": b0["attention_mask"][rows],
         "dna_tokenized": {k: v[dsel] for k, v in b0["dna_tokenized"].items()}, "batch_idx_map": bmap}
    seen = {}

    def reward(ids, mask):
        r = torch.stack([(ids[:, 0] % 5).float() + 0.5 * rank, (ids[:, 1] % 3).float()], dim=1)
        seen["local"] = r.clone()
        return r

    G = 4
    runner = GRPOStepRunner(m, GRPOConfig(num_generations=G, max_completion_length=3, eos_token_id=None, seed=7, learning_rate=1e-3), reward)
    out = runner.step(b)
    gathered = [torch.empty_like(seen["local"]) for _ in range(world)]
    dist.all_gather(gathered, seen["local"])
    allr = torch.cat(gathered, 0).sum(1)                             # [12]
    grp = allr.view(-1, G)
    want_adv = ((allr - grp.mean(1).repeat_interleave(G)) / (grp.std(1).repeat_interleave(G) + 1e-4))[rank * 3:(rank + 1) * 3]
    params = [torch.empty_like(m.arena.params) for _ in range(world)]
    dist.all_gather(params, m.arena.params)
    mets = [torch.empty_like(out["metrics_t"]) for _ in range(world)]
    dist.all_gather(mets, out["metrics_t"])
    got_adv = runner._buffered_inputs[0]["advantages"]
    q.put((rank, all(bool(torch.equal(params[0], p_)) for p_ in params), all(bool(torch.equal(mets[0], m_)) for m_ in mets),
           want_adv.tolist(), got_adv.tolist(), float(out["metrics_t"][1]), float(allr.mean())))
    dist.destroy_process_group()


def test_grpo_step_four_ranks_groups_span_ranks_unevenly(emu_lib_path):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 37500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker_uneven, args=(r, 4, port, emu_lib_path, q)) for r in range(4)]
    for p in procs:
        p.start()
    res = [q.get(timeout=1500) for _ in range(4)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, same_p, same_m, want_adv, got_adv, reward_mean, want_reward in res:
        assert same_p, "replicas diverged after the optimiser step"
        assert same_m, "metric record differs across ranks"
        assert torch.allclose(torch.tensor(got_adv), torch.tensor(want_adv), atol=1e-5), (rank, got_adv, want_adv)
        assert abs(reward_mean - want_reward) < 1e-5


def _single_rank_worker(port, emu_path, q, dp):
    """one process; dp: a ONE-rank gloo group with BRA_DP_SINGLE_RANK=1 (every collective of the step issued, each the identity) — or no
    process group at all"""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), BRA_EMU_THREADS="2")
    torch.set_num_threads(1)
    if dp:
        os.environ["BRA_DP_SINGLE_RANK"] = "1"
        dist.init_process_group("gloo", rank=0, world_size=1)
    from bioreason_amd import _lib
    _lib.use_library_for_tests(emu_path)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_model_parity import build, to_dev
    from bioreason_amd.trainer import GRPOConfig, GRPOStepRunner
    fix = torch.load(os.path.join(GOLD, "tiny_a.pt"), weights_only=False)
    m = build(fix, torch.device("cpu"), True)
    b = to_dev(fix["batch"], torch.device("cpu"))
    b.pop("labels")
    runner = GRPOStepRunner(m, GRPOConfig(num_generations=2, max_completion_length=4, eos_token_id=None, seed=7, learning_rate=1e-3),
                            lambda ids, mask: torch.stack([(ids[:, 0] % 5).float(), (ids[:, 1] % 3).float()], dim=1))
    out = runner.step(b)
    q.put((dp, bool(runner.dp), len(runner._cuts), m.arena.params.tolist(), out["metrics_t"].tolist()))
    if dp:
        dist.destroy_process_group()


def test_single_rank_group_issues_the_collectives_and_changes_nothing(emu_lib_path):
    """BRA_DP_SINGLE_RANK (the hardware rehearsal of the RCCL path on a 1-GPU box, tools/rccl_single_rank.sh): the data-parallel code
    path — bucketed all-reduce from inside the backward, packed all-gather, metric slot — in a one-rank group gives the parameters
    and the metric record of the plain single-process step (the SGD-free part is deterministic on the emulator)"""
    ctx = mp.get_context("spawn")
    res = {}
    for dp in (True, False):
        q = ctx.Queue()
        p = ctx.Process(target=_single_rank_worker, args=(35500 + (os.getpid() % 2000), emu_lib_path, q, dp))
        p.start()
        r = q.get(timeout=900)
        p.join(timeout=60)
        assert p.exitcode == 0
        res[dp] = r
    assert res[True][1] and res[True][2] >= 1 and not res[False][1]
    plain = torch.tensor(res[False][3])
    assert torch.allclose(torch.tensor(res[True][3])[: plain.numel()], plain, rtol=0, atol=1e-6)       # (the DP arena carries a 64-float metric slot at its end)
    assert torch.allclose(torch.tensor(res[True][4]), torch.tensor(res[False][4]), atol=1e-5)


def _sft_worker(rank, world, port, emu_path, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["BRA_EMU_THREADS"] = "2"
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from bioreason_amd import _lib
    _lib.use_library_for_tests(emu_path)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_model_parity import build, to_dev
    from bioreason_amd.trainer import SFTStepRunner
    fix = torch.load(os.path.join(GOLD, "tiny_a.pt"), weights_only=False)
    m = build(fix, torch.device("cpu"), True)
    m.train()
    b = to_dev(fix["batch"], torch.device("cpu"))
    # each rank trains on its own half of the batch (rows are whole samples with their DNA sequences): DDP semantics
    nb = b["input_ids"].shape[0]
    half = list(range(rank * nb // world, (rank + 1) * nb // world))
    bmap = b["batch_idx_map"]
    dsel = [i for i, s in enumerate(bmap) if s in half]
    local = {"input_ids": b["input_ids"][half], "attention_mask": b["attention_mask"][half], "labels": b["labels"][half],
             "dna_tokenized": {k: v[dsel] for k, v in b["dna_tokenized"].items()}, "batch_idx_map": [bmap[i] - half[0] for i in dsel]}
    runner = SFTStepRunner(m, learning_rate=1e-3, weight_decay=0.0)
    p0 = m.arena.params.clone()
    out = runner.step(local)
    params = [torch.empty_like(m.arena.params) for _ in range(world)]
    dist.all_gather(params, m.arena.params)
    grads = [torch.empty_like(m.arena.grads) for _ in range(world)]
    dist.all_gather(grads, m.arena.grads)
    q.put((rank, bool(torch.equal(params[0], params[1])), bool(torch.equal(grads[0], grads[1])),
           float((m.arena.params - p0).abs().max()), float(out["loss_t"])))
    dist.destroy_process_group()


def test_sft_step_two_ranks(emu_lib_path):
    """train_dna_qwen.py's DDP step on 2 ranks (gloo, emulator): different local batches, identical replicas after the bucketed
    gradient mean + AdamW — the path `bench.py --mode sft --gpus N` runs"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_sft_worker, args=(r, 2, port, emu_lib_path, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=900) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    losses = []
    for rank, same_p, same_g, moved, loss in res:
        assert same_p, "replicas diverged after the optimiser step"
        assert same_g, "gradient buckets differ across ranks after the all-reduce"
        assert moved > 0 and loss == loss
        losses.append(loss)
    assert losses[0] != losses[1], "the two ranks should have seen different samples"
