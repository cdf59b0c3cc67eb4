"""Is the host the limit of the MFMA phases of the GRPO step?  For the reference pass, the policy forward and the policy backward of the
cfg-3 step (one prompt x 8 rollouts, full-size models): wall time the host needs to ISSUE the phase (no synchronisation) against the time
until the device has finished it.  issue ~ total means the launches of the phase are paced by Python, not by the GPU."""
import os, sys, time, argparse
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from bioreason_amd import grpo

args = argparse.Namespace(no_graph=False, no_shared_decode=False, no_shared_policy=False, no_overlap_ref=False, lora_dropout=0.05)
dev = torch.device("cuda:0")
dims = bench.Dims(False)
model = bench.build_model(dims, dev, 0.05)
runner, step, B = bench.make_grpo_leg(model, dims, 1, dims.c, 0, dev, args, None, 8)
batch = None
for c in step.__closure__:
    try:
        v = c.cell_contents
    except ValueError:
        continue
    if isinstance(v, dict) and "input_ids" in v:
        batch = v
for _ in range(2):
    step(0)
torch.cuda.synchronize()
pc = time.perf_counter
for it in range(3):
    inputs = runner.generate_and_score(batch)            # (joined: everything complete on the main stream's queue)
    torch.cuda.synchronize()
    mm = inputs["multimodal_inputs"]
    t0 = pc()
    with torch.no_grad(), model.text_model.disable_adapter():
        grpo.per_token_logps_shared_prefix(model, inputs["prompt_ids"], inputs["prompt_mask"], inputs["completion_ids"], inputs["completion_mask"],
                                           inputs["prompt_alias"], **mm)
    t1 = pc(); torch.cuda.synchronize(); t2 = pc()
    model.arena.zero_grad()
    t3 = pc()
    loss, stats = runner.compute_loss(inputs)
    t4 = pc(); torch.cuda.synchronize(); t5 = pc()
    loss.backward()
    t6 = pc(); torch.cuda.synchronize(); t7 = pc()
    print(f"iter {it}: reference pass issue {1e3*(t1-t0):.1f} ms / done {1e3*(t2-t0):.1f} | policy forward issue {1e3*(t4-t3):.1f} / done {1e3*(t5-t3):.1f} | "
          f"backward issue {1e3*(t6-t5):.1f} / done {1e3*(t7-t5):.1f}", flush=True)
