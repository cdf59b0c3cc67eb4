cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 ) > gpurun_out/r5_a_pytest_gpu.log 2>&1
tail -5 gpurun_out/r5_a_pytest_gpu.log
( timeout 700 python bench.py --steps 5 --no-cpu-baseline --no-qwen3-4b 2>gpurun_out/r5_a_bench.err | tail -1 ) > gpurun_out/r5_a_bench.json
python - <<'PY'
import json
d=json.load(open("gpurun_out/r5_a_bench.json"))
print({k:d.get(k) for k in ("value","ms_per_step","value_reference_semantics")})
print(json.dumps(d.get("gpu_baseline_hf"),indent=0)[:1500])
print({k:(v.get("value") if isinstance(v,dict) else v) for k,v in d.items() if k in ("sft","straggler","unshared_policy")})
print(d["roofline"]["frac"], d["roofline_mfma"]["frac"], d["roofline_mfma"].get("one_stream",{}).get("frac"))
PY
