"""2-process RCCL smoke of the data-parallel GRPO step on real GPUs (`-m gpu`; skipped with fewer than 2 devices): the same
assertions as the gloo / emulator test (tests/test_distributed.py) — replicas identical after the bucketed, overlapped
gradient all-reduce, advantages from the gathered rewards — on backend "nccl" (= RCCL) with libbioreason_hip.so."""
import os
import socket
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist
    torch.cuda.set_device(rank)
    dev = torch.device("cuda", rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    if rank == 0:        # the first multi-GPU execution is self-diagnosing: what RCCL, how many ranks, which devices
        try:
            ver = ".".join(str(v) for v in torch.cuda.nccl.version())
        except Exception as e:
            ver = "unknown (%s)" % type(e).__name__
        print("[nccl smoke] RCCL %s, world_size %d, visible devices %d (%s), HSA_ENABLE_IPC_MODE_LEGACY=%s"
              % (ver, dist.get_world_size(), torch.cuda.device_count(), torch.cuda.get_device_name(0),
                 os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY")), flush=True)
    from test_model_parity import GOLD, build, to_dev
    from bioreason_amd.trainer import GRPOConfig, GRPOStepRunner
    fix = torch.load(os.path.join(GOLD, "tiny_a.pt"), weights_only=False)
    m = build(fix, dev, True)
    b = to_dev(fix["batch"], dev)
    b.pop("labels")
    seen = {}

    def reward(ids, mask):
        r = torch.stack([(ids[:, 0] % 5).float() + rank, (ids[:, 1] % 3).float()], dim=1)
        seen["local"] = r.clone()
        return r

    runner = GRPOStepRunner(m, GRPOConfig(num_generations=2 * world, max_completion_length=4, eos_token_id=None, seed=7, learning_rate=1e-3), reward)
    out = runner.step(b)
    gathered = [torch.empty_like(seen["local"]) for _ in range(world)]
    dist.all_gather(gathered, seen["local"])
    allr = torch.cat(gathered, 0).sum(1)
    want_adv = ((allr - allr.mean()) / (allr.std() + 1e-4))[rank * 2:(rank + 1) * 2]
    got_adv = runner._buffered_inputs[0]["advantages"]
    params = [torch.empty_like(m.arena.params) for _ in range(world)]
    dist.all_gather(params, m.arena.params)
    q.put((rank, bool(torch.equal(params[0], params[1])), bool(torch.allclose(got_adv, want_adv, atol=1e-5)),
           bool(torch.isfinite(out["loss_t"]).item()), len(runner._cuts)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
def test_grpo_step_two_gpus_rccl():
    if not torch.cuda.is_available() or torch.cuda.device_count() < 2:
        pytest.skip("needs 2 GPUs")
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in range(2)]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for rank, same_p, adv_ok, finite, ncuts in res:
        assert same_p and adv_ok and finite and ncuts >= 1


# ---------------------------------------------------------------------------------------------------------------------------
# Stream ordering of the bucketed gradient reduction, checkable with ONE rank: with the bf16 transport a bucket is
# cast into its own image when the hook fires, the image is reduced, and the arena range is rewritten from the image at the end of the
# step.  So (a) whatever is written into the range AFTER the hook must not appear in the result (a poison fill proves the reduction
# consumed the hook-time values), and (b) everything written BEFORE the hook — also by the completion chain on its side stream, here
# delayed by a spin kernel in front of every one of its layers — must: the parameters after the step equal those of the undisturbed run
# bit for bit, and the per-bucket stamps are ordered (issued inside the backward, joined no earlier than issued).
def _ordering_worker(backend, port, q, emu_path, disturb):
    try:
        _ordering_worker_body(backend, port, q, emu_path, disturb)
    except BaseException as e:                   # the parent must not sit in q.get() until its timeout
        q.put(("error", "%s: %s" % (type(e).__name__, str(e)[:500])))
        raise


def _ordering_worker_body(backend, port, q, emu_path, disturb):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0", BRA_DP_SINGLE_RANK="1")
    import torch.distributed as dist
    if backend == "nccl":
        torch.cuda.set_device(0)
        dev = torch.device("cuda", 0)
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    else:
        os.environ["BRA_EMU_THREADS"] = "2"
        torch.set_num_threads(1)
        dev = torch.device("cpu")
        dist.init_process_group("gloo", rank=0, world_size=1)
        from bioreason_amd import _lib
        _lib.use_library_for_tests(emu_path)
    from test_model_parity import GOLD, build, to_dev
    from bioreason_amd.trainer import GRPOConfig, GRPOStepRunner
    from test_shared_policy import _group_batch
    fix = torch.load(os.path.join(GOLD, "tiny_a.pt"), weights_only=False)
    m = build(fix, dev, True)
    # G copies of every prompt, consecutive (RepeatRandomSampler's order): the shared-prompt policy pass with its two chains
    G = 2
    ids, mask, mm, alias = _group_batch(fix, dev, G)
    b = {"input_ids": ids, "attention_mask": mask, "dna_tokenized": mm["dna_tokenized"], "batch_idx_map": mm["batch_idx_map"],
         "prompt_alias": alias}
    cfg = GRPOConfig(num_generations=G, max_completion_length=4, eos_token_id=None, seed=7, learning_rate=1e-3, share_policy_prompt=True,
                     overlap_policy_chains=True, grad_allreduce_dtype="bf16", grad_buckets=2)
    runner = GRPOStepRunner(m, cfg, lambda ids, mask: torch.stack([(ids[:, 0] % 5).float(), (ids[:, 1] % 3).float()], dim=1))
    runner.trace_buckets = True
    n_poison = [0]
    if disturb:
        orig_issue = runner._issue

        def issue_then_poison(lo, hi):
            before = len(runner._bf_images)
            orig_issue(lo, hi)
            for r_lo, r_hi, _img in runner._bf_images[before:]:
                m.arena.grads[r_lo:r_hi].fill_(float("nan"))          # enqueued AFTER the hook's cast on the same stream
                n_poison[0] += 1
        runner._issue = issue_then_poison
        if dev.type == "cuda":
            eng = m.text_model.ensure_packed()
            orig_bwd = eng.layer_bwd
            main = torch.cuda.default_stream(dev)

            def slow_side_bwd(*a, **kw):
                if torch.cuda.current_stream(dev) != main:
                    torch.cuda._sleep(3_000_000)                          # ~1.5 ms in front of every layer of the completion chain
                return orig_bwd(*a, **kw)
            eng.layer_bwd = slow_side_bwd
    out = runner.step(b)
    rep = runner.bucket_report() if dev.type == "cuda" else []
    q.put((bool(runner.dp), len(runner._cuts), n_poison[0], m.arena.params.float().cpu().tolist(), float(out["loss_t"]), rep))
    dist.destroy_process_group()


def _run_ordering(backend, emu_path):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    res = {}
    for disturb in (False, True):
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        q = ctx.Queue()
        p = ctx.Process(target=_ordering_worker, args=(backend, port, q, emu_path, disturb))
        p.start()
        res[disturb] = q.get(timeout=900)
        assert res[disturb][0] != "error", res[disturb]
        p.join(timeout=120)
        assert p.exitcode == 0
    dp, ncuts, npoison, params, loss, rep = res[True]
    assert dp and ncuts >= 1 and npoison >= 2                               # at least one in-backward bucket and the tail were poisoned
    clean = torch.tensor(res[False][3])
    got = torch.tensor(params)
    assert torch.isfinite(got).all(), "a value written after the hook reached the reduced gradients"
    assert torch.equal(got, clean), "the reduction did not see the gradients of the undisturbed step"
    assert loss == res[False][4]
    return rep


def test_bucket_reduction_consumes_hook_time_gradients_gloo(emu_lib_path):
    _run_ordering("gloo", emu_lib_path)


@pytest.mark.gpu
def test_bucket_reduction_stream_ordering_one_rank_rccl():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    rep = _run_ordering("nccl", None)
    assert len(rep) >= 2
    for t in rep:
        assert t["joined_at_ms"] is not None and t["joined_at_ms"] >= t["issued_at_ms"] >= 0.0
        assert t["joined_at_ms"] >= t["backward_end_ms"] - 1e-3              # joins happen after the backward's last launch
    assert rep[0]["issued_at_ms"] < rep[0]["backward_end_ms"]                # the first bucket left from INSIDE the backward
    print("[dp buckets]", rep)
