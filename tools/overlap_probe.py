import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import argparse, bench
from bioreason_amd import grpo
args = argparse.Namespace(no_graph=False, no_shared_decode=False, no_shared_policy=False, no_overlap_ref=False, lora_dropout=0.05)
dev = torch.device("cuda:0")
dims = bench.Dims(False)
model = bench.build_model(dims, dev, 0.05)
runner, step, B = bench.make_grpo_leg(model, dims, 1, dims.c, 0, dev, args, None, 4)
batch = None
# full-size: ref pass on the side stream while the policy forward runs on the main stream, 6 times; compare with a quiet recompute
import types
for it in range(4):
    bt = step.__closure__
    inputs = None
    # reach into the leg: rebuild the batch through make_grpo_leg's closure
    for c in step.__closure__:
        try:
            v = c.cell_contents
        except ValueError:
            continue
        if isinstance(v, dict) and "input_ids" in v:
            batch = v
    inputs = runner.generate_and_score(batch, defer_ref_join=True)
    lp = grpo.per_token_logps_shared_policy(model, inputs["prompt_ids"], inputs["prompt_mask"], inputs["completion_ids"],
                                            inputs["completion_mask"], inputs["prompt_alias"], **inputs["multimodal_inputs"])
    torch.cuda.current_stream(dev).wait_stream(inputs["ref_join"])
    torch.cuda.synchronize()
    ref_side = inputs["ref_per_token_logps"].clone()
    with torch.no_grad(), model.text_model.disable_adapter():
        ref_quiet = grpo.per_token_logps_shared_prefix(model, inputs["prompt_ids"], inputs["prompt_mask"], inputs["completion_ids"],
                                                       inputs["completion_mask"], inputs["prompt_alias"], **inputs["multimodal_inputs"])
    with torch.no_grad():
        lp_quiet = grpo.per_token_logps_shared_policy(model, inputs["prompt_ids"], inputs["prompt_mask"], inputs["completion_ids"],
                                                      inputs["completion_mask"], inputs["prompt_alias"], **inputs["multimodal_inputs"])
    torch.cuda.synchronize()
    print("iter", it, "ref side == quiet:", bool(torch.equal(ref_side, ref_quiet)), float((ref_side - ref_quiet).abs().max()),
          "| policy beside ref vs quiet (dropout masks differ per call, so only finite check):", bool(torch.isfinite(lp).all()), flush=True)

# the two-stream chains of the policy pass at full size: log-probs bit-identical, gradients equal up to atomics order (dropout off)
model.text_model.lora_dropout_p = 0.0
side = torch.cuda.Stream(device=dev)
res = []
for s_ in (None, side, side):
    model.arena.zero_grad()
    lp = grpo.per_token_logps_shared_policy(model, inputs["prompt_ids"], inputs["prompt_mask"], inputs["completion_ids"],
                                            inputs["completion_mask"], inputs["prompt_alias"], side=s_, **inputs["multimodal_inputs"])
    lp.sum().backward()
    torch.cuda.synchronize()
    res.append((lp.detach().clone(), model.arena.grads.clone()))
for lp, gr in res[1:]:
    print("two-stream chains: logps equal", bool(torch.equal(lp, res[0][0])), "grads rel", float((gr - res[0][1]).norm() / res[0][1].norm()), flush=True)
