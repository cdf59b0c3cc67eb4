#!/bin/bash
# Hardware rehearsal of the data-parallel path on a ONE-GPU box: the driver's multi-GPU command line with --nproc-per-node 1 and
# BRA_DP_SINGLE_RANK=1, so that the RCCL group is created (device_id init, dmabuf IPC mode) and every collective of the step is
# issued on it — the bucketed asynchronous gradient all-reduces from inside the two-stream backward, the packed reward all-gather,
# the metric slot, the barrier and the MAX-reduce of the elapsed time.  In a one-rank group each collective is the identity, so the
# line must equal the plain single-process run (same loss, same throughput within noise).  Multi-rank semantics: tests/test_distributed.py
# (gloo, 2 ranks) and tests/test_nccl_smoke.py (2 GPUs).
R=${GRAFT_REPO_ROOT:-/root/repo}
export BRA_DP_SINGLE_RANK=1 HSA_ENABLE_IPC_MODE_LEGACY=0
timeout ${T:-280} python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port ${PORT:-29517} \
    $R/bench.py --gpus 1 --steps ${STEPS:-3} --warmup 2 --no-cpu-baseline --no-secondary --no-gpu-baseline-hf "$@"
