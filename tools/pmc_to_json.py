"""PMC summaries (tools/pmc_summarize.py CSVs) + the bench line of a tag -> <tag>_pmc_gemm.json: HBM-side traffic per API call of the
MFMA GEMM family and per token step of the rollout, matrix-pipe busy fractions, and the source hashes bench.py checks.
   python tools/pmc_to_json.py <dir> <tag> <repo root> [GRPO steps in the trace]"""
import csv, hashlib, json, os
import sys
out, tag, R = sys.argv[1], sys.argv[2], sys.argv[3]
NSTEPS = float(sys.argv[4]) if len(sys.argv) > 4 else 3.0   # GRPO steps in the trace
def rows(name):
    p = os.path.join(out, f"{tag}_pmc_{name}.csv")
    return list(csv.DictReader(open(p))) if os.path.exists(p) else []
def pick(rs, counter, key):
    return [r for r in rs if r["Counter_Name"] == counter and key in r["Kernel_Name"]]
h = hashlib.sha256()
for f in ("k_gemm.hip", "bra_device.h"):
    h.update(open(os.path.join(R, "bioreason_amd", "csrc", f), "rb").read())
res = {"kernel_source_sha": h.hexdigest()[:16], "command": "bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-secondary (the timed configuration; GRPO steps traced: see steps_in_trace)",
       "fetch_correction": 2.0, "steps_in_trace": NSTEPS, "kernels": {}}
tot_b, tot_n = 0.0, 0
for key in ("gemm_ring_kernel<0", "gemm_glds_kernel<0", "gemm_w4_kernel<0"):
    # (round 4: the LDS-DMA kernel has one instantiation per tile height — every kernel name that matches the key is summed)
    f = {r["Kernel_Name"]: r for r in pick(rows("FETCH_SIZE"), "FETCH_SIZE", key)}
    w = {r["Kernel_Name"]: r for r in pick(rows("WRITE_SIZE"), "WRITE_SIZE", key)}
    for nm in sorted(set(f) & set(w)):
        n = int(f[nm]["Dispatches"]); fk, wk = float(f[nm]["Mean"]), float(w[nm]["Mean"])
        b = (2.0 * fk + wk) * 1024.0            # FETCH_SIZE / WRITE_SIZE are in KiB; FETCH_SIZE reads 1/2 of a wide coalesced stream on gfx950
        res["kernels"][nm.split("(")[0][-48:]] = {"dispatches": n, "fetch_kb_raw": fk, "write_kb_raw": wk, "traffic_bytes_per_launch": b}
        tot_b += b * n; tot_n += n
if tot_n:
    # ONE denominator: bench.py's `algorithmic_bytes_per_launch` is per API call (bra_gemm_bf16_nt; a row-split call = one ring
    # dispatch + one 256x128 dispatch), so the traffic is divided by API calls too: calls per step from the bench line of this tag
    # (roofline.launches / steps) x the 3 steps in the trace.  The per-dispatch figure is kept beside it, labelled.
    res["traffic_bytes_total_in_trace"] = tot_b
    res["dispatches"] = tot_n
    res["traffic_bytes_per_dispatch"] = tot_b / tot_n
    try:
        bl = json.load(open(os.path.join(out, f"{tag}_bench.json")))
        calls_per_step = bl["roofline_mfma"]["launches"] / bl["steps"]
        res["api_calls_per_step"] = calls_per_step
        res["traffic_bytes_per_call"] = tot_b / (NSTEPS * calls_per_step)
        res["algorithmic_bytes_per_call"] = bl["roofline_mfma"]["algorithmic_bytes_per_launch"]
        res["traffic_over_algorithmic"] = res["traffic_bytes_per_call"] / max(res["algorithmic_bytes_per_call"], 1.0)
    except Exception as e:
        res["traffic_bytes_per_call_error"] = repr(e)
# the token loop (dominant by time): HBM-side bytes per token step over its kernels (255 token steps per GRPO step, 3 steps traced)
dh = hashlib.sha256()
for f in ("k_decgemm.hip", "bra_decgemm.h", "k_decattn.hip", "bra_decattn.h", "k_decode.hip", "k_grpo.hip", "bra_device.h"):
    dh.update(open(os.path.join(R, "bioreason_amd", "csrc", f), "rb").read())
res["decode_source_sha"] = dh.hexdigest()[:16]
dec_b, dec_k = 0.0, {}
for key in ("dec_gemm2_kernel", "dec_attn_items_kernel", "dec_attn_merge_kernel", "topk_slices_kernel", "sample_merge_kernel", "sample_tiles_kernel", "tile_max_kernel", "advance_counters_kernel"):
    fb = sum(float(r["Total"]) for r in rows("FETCH_SIZE") if r["Counter_Name"] == "FETCH_SIZE" and key in r["Kernel_Name"])
    wb = sum(float(r["Total"]) for r in rows("WRITE_SIZE") if r["Counter_Name"] == "WRITE_SIZE" and key in r["Kernel_Name"])
    b = (2.0 * fb + wb) * 1024.0
    dec_k[key] = b
    dec_b += b
if dec_b > 0:
    res["decode_traffic_bytes_in_trace"] = dec_b
    res["decode_traffic_by_kernel_in_trace"] = dec_k
    res["decode_traffic_bytes_per_token_step"] = dec_b / (NSTEPS * 255.0)
sq = rows("SQ_BUSY_CYCLES")
for key in ("gemm_ring_kernel<0", "gemm_glds_kernel<0", "gemm_w4_kernel<0", "attn_fwd_kernel<128", "attn_bwd_dq_kernel", "attn_bwd_dkv_kernel<128", "dec_gemm2_kernel<0, 2, 1", "dec_attn_items_kernel", "dec_attn_merge_kernel"):
    busy, mfma = pick(sq, "SQ_BUSY_CYCLES", key), pick(sq, "SQ_VALU_MFMA_BUSY_CYCLES", key)
    if busy and mfma:
        # SQ_BUSY_CYCLES is reported per shader engine (32 SEs), SQ_VALU_MFMA_BUSY_CYCLES summed over the 1024 SIMDs; every
        # instantiation that matches the key is summed
        bt, mt = sum(float(r["Total"]) for r in busy), sum(float(r["Total"]) for r in mfma)
        util = mt / max(bt / 32.0 * 1024.0, 1.0)
        res.setdefault("mfma_busy", {})[key] = {"mfma_busy_cycles": mt, "sq_busy_cycles": bt, "mfma_pipe_busy_frac": util}
json.dump(res, open(os.path.join(out, f"{tag}_pmc_gemm.json"), "w"), indent=1)
print(json.dumps(res, indent=1)[:1500])