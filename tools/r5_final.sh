cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 720 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 ) > gpurun_out/r5_f_pytest_gpu.log 2>&1
tail -4 gpurun_out/r5_f_pytest_gpu.log
( T=150 STEPS=3 bash tools/rccl_single_rank.sh --grad-bf16 2>&1 | grep -v amdgpu.ids | tail -3 ) > gpurun_out/r5_f_rccl_single_rank.log 2>&1
python - <<'PY'
import json
try:
    d=json.loads(open("gpurun_out/r5_f_rccl_single_rank.log").read().strip().splitlines()[-1])
    print("rccl 1-rank bf16 buckets:", d["value"], d["ms_per_step"], d["loss"])
except Exception as e:
    print("rccl rehearsal:", e, open("gpurun_out/r5_f_rccl_single_rank.log").read()[-600:])
PY
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/fp8_stats
timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/fp8_stats -o r --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --rollout-fp8 --no-cpu-baseline --no-secondary --no-one-stream-profile --no-gpu-baseline-hf > /tmp/fp8_stats.log 2>&1
f=$(find /tmp/fp8_stats -name "*kernel_stats.csv" | head -1)
{ echo "# rocprofv3 --kernel-trace --stats -- bench.py --steps 1 --warmup 1 --rollout-fp8 --no-secondary (MI355X, r5_f: 3 GRPO steps with fp8 rollout weights)"; cat $f; } > $GRAFT_REPO_ROOT/gpurun_out/r5_f_fp8_kernel_stats.csv
cd $GRAFT_REPO_ROOT
( timeout 900 python bench.py --steps 5 2>gpurun_out/r5_f_bench.err | tail -1 ) > gpurun_out/r5_f_bench.json
python - <<'PY'
import json
d=json.load(open("gpurun_out/r5_f_bench.json"))
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"].get("traffic"), d["roofline_mfma"]["frac"], d["roofline_mfma"].get("traffic"))
print({k:(v.get("value") if isinstance(v,dict) else v) for k,v in d.items() if k in ("sft","straggler","unshared_policy","rollout_fp8","qwen3_4b","gpu_baseline_hf","cpu_baseline","value_reference_semantics")})
PY
