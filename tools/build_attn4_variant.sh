#!/bin/bash
# builds bioreason_amd/libbioreason_hip_<name>.so = the debug library with k_attn4.hip recompiled with extra flags (A/B probes of
# the 4-wave attention forward; select with AP_LIB / AS_LIB).  usage: tools/build_attn4_variant.sh <name> "<flags>" [source file stem, default k_attn4]
set -e
cd "$(dirname "$0")/../bioreason_amd/csrc"
make -s -j8 debug > /dev/null
mkdir -p build_var
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffast-math -fno-finite-math-only -Wno-unused-result -I. -mllvm -amdgpu-kernarg-preload-count=16 -mllvm -amdgpu-mfma-vgpr-form -DBRA_DEBUG"
SRC=${3:-k_attn4}
/opt/rocm/bin/hipcc $F $2 -c $SRC.hip -o build_var/${SRC}_$1.o 2>&1 | grep -v "warning\|^$" || true
OBJS=$(ls build_dbg/*.o | grep -v "/$SRC.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libbioreason_hip_$1.so $OBJS build_var/${SRC}_$1.o 2>&1 | grep -v "warning\|^$" || true
ls -la ../libbioreason_hip_$1.so
