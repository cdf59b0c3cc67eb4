"""`python bench.py --gpus N` rehearsed without GPUs (BENCH_DRYRUN=1: gloo instead of RCCL, the kernel-source emulator instead of
libbioreason_hip.so, toy dimensions): the self-launch under torch.distributed.run (the reference starts its trainers from one
command the same way, sh_reason.sh:38-44), argument forwarding, the WORLD_SIZE / --gpus check, both collectives of the step, the
MAX-reduce of the elapsed time and the single rank-0 JSON line.  The numbers of a dry run mean nothing; the line says so."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")
REQUIRED = ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
            "data", "config", "roofline"]


def _run(argv, extra_env=None, timeout=600):
    env = dict(os.environ, BENCH_DRYRUN="1", BRA_EMU_THREADS="2", OMP_NUM_THREADS="1")
    env.pop("RANK", None), env.pop("WORLD_SIZE", None), env.pop("LOCAL_RANK", None)
    env.update(extra_env or {})
    return subprocess.run([sys.executable, BENCH] + argv, capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)


def _json_lines(stdout):
    out = []
    for ln in stdout.splitlines():
        ln = ln.strip()
        if ln.startswith("{") and ln.endswith("}"):
            try:
                out.append(json.loads(ln))
            except ValueError:
                pass
    return out


def test_two_rank_self_launch_prints_one_line(emu_lib_path):
    r = _run(["--gpus", "2", "--steps", "2", "--warmup", "1", "--completion-len", "3"])
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    lines = _json_lines(r.stdout)
    assert len(lines) == 1, "exactly one JSON line (rank 0) expected, got %d" % len(lines)
    d = lines[0]
    for k in REQUIRED:
        assert k in d, k
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["warmup"] == 1 and d["scaling"] == "weak" and d["dryrun"] is True
    assert d["config"]["parallelism"] == "dp2" and d["config"]["completion_len"] == 3          # forwarded through the launcher
    assert d["config"]["global_batch"] == 2 * 2                                                 # G = 2 rows per rank in the dry run
    # value = whole-job samples / MAX-over-ranks elapsed: consistent with ms_per_step
    assert abs(d["value"] - d["config"]["global_batch"] / (d["ms_per_step"] / 1000.0)) < 1e-6 * d["value"]
    assert "sft" not in d and "straggler" not in d       # secondary legs are N = 1 only
    assert d["metrics"]["completion_length"] == 3.0
    # the line explains a slow run by itself (round 5): per-step host time and the host's load around the timed region; allocator fields on a GPU
    diag = d["timed_region_diagnostics"]
    assert len(diag["host_ms_inside_step_calls"]) == d["steps"] and "host_loadavg_before" in diag


def test_eight_rank_self_launch_prints_one_line(emu_lib_path):
    """the command the driver's SCALE record runs — `python bench.py --gpus 8` (the reference: `deepspeed --num_gpus=8 reason.py`,
    sh_reason.sh:38-44) — with eight processes: rendezvous, argument forwarding, both collectives at world size 8, MAX-reduce of
    the elapsed time, ONE line from rank 0"""
    r = _run(["--gpus", "8", "--steps", "1", "--warmup", "1", "--completion-len", "2"], extra_env={"BRA_EMU_THREADS": "1"}, timeout=1500)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    lines = _json_lines(r.stdout)
    assert len(lines) == 1, "exactly one JSON line (rank 0) expected, got %d" % len(lines)
    d = lines[0]
    for k in REQUIRED:
        assert k in d, k
    assert d["n_gpus"] == 8 and d["steps"] == 1 and d["warmup"] == 1 and d["scaling"] == "weak" and d["dryrun"] is True
    assert d["config"]["parallelism"] == "dp8" and d["config"]["completion_len"] == 2
    assert d["config"]["global_batch"] == 8 * 2                                                 # G = 2 rows per rank in the dry run
    assert abs(d["value"] - d["config"]["global_batch"] / (d["ms_per_step"] / 1000.0)) < 1e-6 * d["value"]
    assert "sft" not in d and "straggler" not in d


def test_single_process_line_carries_the_secondary_legs(emu_lib_path):
    r = _run(["--steps", "1", "--warmup", "1", "--secondary-steps", "1", "--completion-len", "4"])
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    (d,) = _json_lines(r.stdout)
    assert d["n_gpus"] == 1 and "cpu_baseline" not in d                  # the CPU leg is skipped in a dry run
    for leg in ("sft", "straggler"):
        assert d[leg]["value"] > 0 and d[leg]["steps"] == 1 and d[leg]["unit"] == "samples/s", leg
    assert d["roofline_decode"]["bound"] == "hbm" and d["roofline_mfma"]["bound"] == "mfma"
    assert "gemm_ring_kernel" in d["roofline_mfma"]["kernel"]
    assert d["roofline"]["bound"] in ("hbm", "mfma") and "frac" in d["roofline"]       # = whichever family is dominant by time
    assert d["unshared_policy"]["value"] > 0


def test_world_size_mismatch_is_refused():
    """a launcher that starts N processes for `--gpus M` (M != N) must fail loudly, not report an N-GPU number as M"""
    r = _run(["--gpus", "4"], extra_env={"RANK": "0", "WORLD_SIZE": "2", "LOCAL_RANK": "0", "MASTER_ADDR": "127.0.0.1",
                                         "MASTER_PORT": "29999"}, timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE=2" in (r.stderr + r.stdout)
