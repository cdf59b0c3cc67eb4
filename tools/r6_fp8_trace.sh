#!/bin/bash
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
timeout 300 python -m pytest tests/test_fp8_gemm.py -q -x -p no:cacheprovider -k "decoded_products and hip" 2>&1 | grep -E "^E  |passed|failed" | head -12
timeout 300 python tools/prefill_fp8_probe.py 2>&1 | grep -v amdgpu
cd /tmp; rm -rf /tmp/prof_p
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_p -o p --output-format csv -- python $ROOT/tools/prefill_fp8_probe.py > /dev/null 2>&1
python $ROOT/tools/kernel_breakdown.py "$(find /tmp/prof_p -name '*kernel_stats.csv' | head -1)" 1 30
