cd $GRAFT_REPO_ROOT
for rep in 1 2; do
  echo "== prev"; AP_LIB=libbioreason_hip_prev.so AP_N=20 timeout 120 python tools/attn_probe.py 2>&1 | grep -v "amdgpu.ids\|checksums"
  echo "== now (statistics addressed from a per-iteration base: no spills in the one-pass dK + dV kernel)";   AP_N=20 timeout 120 python tools/attn_probe.py 2>&1 | grep -v "amdgpu.ids"
done
timeout 300 python -m pytest tests/test_kernels.py tests/test_long_text.py -m gpu -q -k "attn or long" 2>&1 | tail -2
