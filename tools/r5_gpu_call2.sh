cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( BRA_FP8_REPORT=1 timeout 600 python -m pytest tests/test_fp8_rollout.py tests/test_kernels.py tests/test_abi.py -m gpu -x -q -s 2>&1 | tail -25 ) > gpurun_out/r5_b_pytest_fp8.log 2>&1
tail -8 gpurun_out/r5_b_pytest_fp8.log
( BENCH_HF_STEPS=2 timeout 900 python bench.py --steps 5 --no-cpu-baseline --no-qwen3-4b 2>gpurun_out/r5_b_bench.err | tail -1 ) > gpurun_out/r5_b_bench.json
python - <<'PY'
import json
d=json.load(open("gpurun_out/r5_b_bench.json"))
print({k:d.get(k) for k in ("value","ms_per_step","value_reference_semantics")})
print(json.dumps(d.get("gpu_baseline_hf"),indent=0)[:2500])
print(json.dumps(d.get("rollout_fp8"),indent=0)[:1200])
print({k:(v.get("value") if isinstance(v,dict) else v) for k,v in d.items() if k in ("sft","straggler","unshared_policy")})
print(d["roofline"]["frac"], d["roofline"].get("ms_per_token_step"), d["roofline_mfma"]["frac"], d["roofline_mfma"].get("one_stream",{}).get("frac"))
PY
tail -5 gpurun_out/r5_b_bench.err
