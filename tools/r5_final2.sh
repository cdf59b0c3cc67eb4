cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 900 python bench.py --steps 5 2>gpurun_out/r5_m_bench.err | tail -1 ) > gpurun_out/r5_m_bench.json
python - <<'PY'
import json
d=json.load(open("gpurun_out/r5_m_bench.json"))
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"].get("traffic"), d["roofline_mfma"]["frac"], d["roofline_mfma"].get("traffic"), d["phases_ms"])
print({k:(v.get("value") if isinstance(v,dict) else v) for k,v in d.items() if k in ("sft","straggler","unshared_policy","rollout_fp8","qwen3_4b","gpu_baseline_hf","cpu_baseline","value_reference_semantics","prompts_per_gpu_2")})
PY
( timeout 720 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 ) > gpurun_out/r5_m_pytest_gpu.log 2>&1
tail -4 gpurun_out/r5_m_pytest_gpu.log
