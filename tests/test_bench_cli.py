"""bench.py's launch contract on CPU ranks (gloo): it starts its own ranks for ``--gpus N`` outside a launcher,
refuses a rank count that differs from ``--gpus``, and a hang inside the measured loop ends with stacks, an error
record and a non-zero exit code instead of silence."""
import json
import os
import subprocess
import sys
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")
SMALL = ["--cpu", "--model", "resnet18", "--image-size", "32", "--batch-size", "4", "--steps", "3", "--warmup", "1"]

pytestmark = pytest.mark.slow


def _env(**extra):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(OMP_NUM_THREADS="2", **extra)
    return env


def test_gpus_2_without_a_launcher_starts_two_ranks_and_prints_one_line():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2"] + SMALL, env=_env(), capture_output=True, text=True, timeout=420)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["config"]["global_batch"] == 8 and rec["value"] > 0 and "error" not in rec
    ex = rec["extras"]
    assert ex["params_in_sync"] is True and len(ex["elapsed_s_per_rank"]) == 2 and "error" not in ex
    assert rec["config"]["bucket_policy"] == "auto" and ex["bucket_autotune"]["bucket_mb"] > 0


def test_rank_count_that_differs_from_gpus_is_an_error_not_a_warning():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2"] + SMALL,
                       env=_env(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999"),
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 2
    rec = json.loads(r.stdout.strip().splitlines()[-1])
    assert rec["value"] is None and "WORLD_SIZE=1" in rec["error"]


def test_a_hang_in_the_step_loop_ends_loudly_within_the_timeout():
    t0 = time.time()
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--step-timeout-s", "6"] + SMALL,
                       env=_env(PDT_FAULT_INJECT="1:3:hang"), capture_output=True, text=True, timeout=300)
    took = time.time() - t0
    assert r.returncode != 0
    assert "[pdt fault] rank 1 hanging at step 3" in r.stderr
    assert "[pdt watchdog] rank" in r.stderr and "dumping stacks" in r.stderr and "Thread 0x" in r.stderr
    recs = [json.loads(ln) for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(recs) == 1 and recs[0]["value"] is None and "no step finished" in recs[0]["error"], r.stdout
    assert took < 120
