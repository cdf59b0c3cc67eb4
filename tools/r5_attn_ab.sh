cd $GRAFT_REPO_ROOT
for rep in 1 2; do
  echo "== b64 (rounds 1-4)"; AP_LIB=libbioreason_hip_b64.so AP_N=20 timeout 120 python tools/attn_probe.py 2>&1 | grep -v amdgpu.ids
  echo "== b128 (round 5)";   AP_N=20 timeout 120 python tools/attn_probe.py 2>&1 | grep -v amdgpu.ids
done
timeout 300 python -m pytest tests/test_kernels.py -m gpu -q -k "attn" 2>&1 | tail -2
