#!/bin/bash
# Evidence of the bench configuration, collected on the GPU box:  tools/collect_evidence.sh <tag>   (e.g. r3_a)
#   profiles/<tag>_bench.json                 the bench line (default command, CPU baseline included unless NOCPU=1)
#   profiles/<tag>_bench_kernel_stats.csv     rocprofv3 --kernel-trace --stats of `bench.py --steps 1 --warmup 1` (3 steps in the trace)
#   profiles/<tag>_pmc_*.csv                  separate rocprofv3 --pmc passes on the SAME command (FETCH_SIZE | WRITE_SIZE | SQ busy counters)
#   profiles/<tag>_pmc_gemm.json              dominant-kernel traffic per launch + the kernel-source hash bench.py checks
tag=${1:-r3_x}
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
out=$R/gpurun_out/evidence_$tag; mkdir -p $out
cd /tmp; export TMPDIR=/tmp
BENCH="python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-secondary"
if [ "$PMC_ONLY" = "1" ]; then :      # only the counter passes (the bench line and the kernel stats of this state exist already)
elif [ "$NOCPU" = "1" ]; then timeout 300 python $R/bench.py --no-cpu-baseline --steps 5 2>/dev/null | tail -1 > $out/${tag}_bench.json
else timeout 600 python $R/bench.py --steps 5 2>/dev/null | tail -1 > $out/${tag}_bench.json; fi
[ "$PMC_ONLY" = "1" ] || { rm -rf /tmp/ev_stats; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/ev_stats -o r --output-format csv -- $BENCH > /tmp/ev_stats.log 2>&1
f=$(find /tmp/ev_stats -name "*kernel_stats.csv" | head -1)
{ echo "# rocprofv3 --kernel-trace --stats --output-format csv -- $BENCH   (MI355X, $tag; 3 GRPO steps in the trace: warm-up, timed, instrumented)"; cat $f; } > $out/${tag}_bench_kernel_stats.csv; }
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU"; do
  name=$(echo $pass | awk '{print $1}')
  rm -rf /tmp/ev_pmc; timeout 400 rocprofv3 --pmc $pass --kernel-trace -d /tmp/ev_pmc -o r --output-format csv -- $BENCH > /tmp/ev_pmc_$name.log 2>&1
  python $R/tools/pmc_summarize.py /tmp/ev_pmc $out/${tag}_pmc_$name.csv >> /tmp/ev_pmc_$name.log 2>&1
done
python - <<PY
import csv, hashlib, json, os
out, tag, R = "$out", "$tag", "$R"
def rows(name):
    p = os.path.join(out, f"{tag}_pmc_{name}.csv")
    return list(csv.DictReader(open(p))) if os.path.exists(p) else []
def pick(rs, counter, key):
    return [r for r in rs if r["Counter_Name"] == counter and key in r["Kernel_Name"]]
h = hashlib.sha256()
for f in ("k_gemm.hip", "bra_device.h"):
    h.update(open(os.path.join(R, "bioreason_amd", "csrc", f), "rb").read())
res = {"kernel_source_sha": h.hexdigest()[:16], "command": "bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-secondary (the timed configuration, 3 GRPO steps traced)",
       "fetch_correction": 2.0, "kernels": {}}
tot_b, tot_n = 0.0, 0
for key in ("gemm_ring_kernel<0", "gemm_glds_kernel<0"):
    f, w = pick(rows("FETCH_SIZE"), "FETCH_SIZE", key), pick(rows("WRITE_SIZE"), "WRITE_SIZE", key)
    if f and w:
        n = int(f[0]["Dispatches"]); fk, wk = float(f[0]["Mean"]), float(w[0]["Mean"])
        b = (2.0 * fk + wk) * 1024.0            # FETCH_SIZE / WRITE_SIZE are in KiB; FETCH_SIZE reads 1/2 of a wide coalesced stream on gfx950
        res["kernels"][key] = {"dispatches": n, "fetch_kb_raw": fk, "write_kb_raw": wk, "traffic_bytes_per_launch": b}
        tot_b += b * n; tot_n += n
if tot_n:
    # ONE denominator: bench.py's `algorithmic_bytes_per_launch` is per API call (bra_gemm_bf16_nt; a row-split call = one ring
    # dispatch + one 256x128 dispatch), so the traffic is divided by API calls too: calls per step from the bench line of this tag
    # (roofline.launches / steps) x the 3 steps in the trace.  The per-dispatch figure is kept beside it, labelled.
    res["traffic_bytes_total_3_steps"] = tot_b
    res["dispatches"] = tot_n
    res["traffic_bytes_per_dispatch"] = tot_b / tot_n
    try:
        bl = json.load(open(os.path.join(out, f"{tag}_bench.json")))
        calls_per_step = bl["roofline_mfma"]["launches"] / bl["steps"]
        res["api_calls_per_step"] = calls_per_step
        res["traffic_bytes_per_call"] = tot_b / (3.0 * calls_per_step)
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
for key in ("dec_gemm2_kernel", "dec_attn_items_kernel", "dec_attn_merge_kernel", "topk_slices_kernel", "sample_merge_kernel", "advance_counters_kernel"):
    fb = sum(float(r["Total"]) for r in rows("FETCH_SIZE") if r["Counter_Name"] == "FETCH_SIZE" and key in r["Kernel_Name"])
    wb = sum(float(r["Total"]) for r in rows("WRITE_SIZE") if r["Counter_Name"] == "WRITE_SIZE" and key in r["Kernel_Name"])
    b = (2.0 * fb + wb) * 1024.0
    dec_k[key] = b
    dec_b += b
if dec_b > 0:
    res["decode_traffic_bytes_3_steps"] = dec_b
    res["decode_traffic_by_kernel_3_steps"] = dec_k
    res["decode_traffic_bytes_per_token_step"] = dec_b / (3.0 * 255.0)
sq = rows("SQ_BUSY_CYCLES")
for key in ("gemm_ring_kernel<0", "gemm_glds_kernel<0", "attn_fwd_kernel<128", "attn_bwd_dq_kernel", "attn_bwd_dkv_kernel<128, 2", "dec_gemm2_kernel<0, 2, 1", "dec_attn_items_kernel", "dec_attn_merge_kernel"):
    busy, mfma = pick(sq, "SQ_BUSY_CYCLES", key), pick(sq, "SQ_VALU_MFMA_BUSY_CYCLES", key)
    if busy and mfma:
        # SQ_BUSY_CYCLES is reported per shader engine (32 SEs), SQ_VALU_MFMA_BUSY_CYCLES summed over the 1024 SIMDs
        util = float(mfma[0]["Total"]) / max(float(busy[0]["Total"]) / 32.0 * 1024.0, 1.0)
        res.setdefault("mfma_busy", {})[key] = {"mfma_busy_cycles": float(mfma[0]["Total"]), "sq_busy_cycles": float(busy[0]["Total"]), "mfma_pipe_busy_frac": util}
json.dump(res, open(os.path.join(out, f"{tag}_pmc_gemm.json"), "w"), indent=1)
print(json.dumps(res, indent=1)[:1500])
PY
ls -la $out
