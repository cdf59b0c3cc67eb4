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
