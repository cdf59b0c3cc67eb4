#!/bin/bash
# round 6: the full GPU suite, then a short bench (no baselines) and the attention probe, on one box
export TMPDIR=/tmp
TAG=${1:-r6}
timeout 1500 python -m pytest tests -q -m gpu -x -p no:cacheprovider > gpurun_out/${TAG}_pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${TAG}_pytest_gpu.log
tail -4 gpurun_out/${TAG}_pytest_gpu.log
timeout 600 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-gpu-baseline-hf > gpurun_out/${TAG}_bench_short.json 2> gpurun_out/${TAG}_bench_short.err; tail -c 1500 gpurun_out/${TAG}_bench_short.json
AP_SHAPES=full,sft,prompt,compl,enc AP_N=20 timeout 300 python tools/attn_probe.py > gpurun_out/${TAG}_attn_probe.txt 2>&1; cat gpurun_out/${TAG}_attn_probe.txt | grep -v amdgpu
