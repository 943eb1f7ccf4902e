"""Failure handling end to end on two CPU ranks (gloo): a rank that dies is restarted by the launcher and the job
continues from its last checkpoint; a stop signal that reaches one rank only stops every rank at the same step."""
import os
import signal
import subprocess
import sys
import time

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MAIN = os.path.join(ROOT, "src", "main.py")

pytestmark = pytest.mark.slow


def _launch(port, extra_env, args, max_restarts=0):
    env = dict(os.environ, OMP_NUM_THREADS="2", PYTHONPATH=ROOT, **extra_env)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--max-restarts",
           str(max_restarts), "--master-addr", "127.0.0.1", "--master-port", str(port), MAIN, "--distributed", "--use-cpu",
           "--dataset", "synthetic", "--dataset-length", "96", "--batch-size", "4", "--num-workers", "0",
           "--learning-rate", "0.001", "--no-progress"] + args
    return subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)


def _port():
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_killed_rank_is_restarted_and_training_resumes_from_the_checkpoint(tmp_path):
    """Rank 1 exits hard after step 3.  `torch.distributed.run --max-restarts=1` starts fresh processes (nothing is
    re-executed in place), `--resume auto` picks up the checkpoint of step 2, and the job ends at step 6 with the
    parameters of a run that was never interrupted."""
    ref = str(tmp_path / "ref.pt")
    p = _launch(_port(), {}, ["--steps", "6", "--checkpoint-path", ref])
    out, err = p.communicate(timeout=300)
    assert p.returncode == 0, err[-3000:]

    path = str(tmp_path / "job.pt")
    marker = str(tmp_path / "fault_fired")
    p = _launch(_port(), {"PDT_FAULT_INJECT": "1:3:exit", "PDT_FAULT_INJECT_ONCE_FILE": marker},
                ["--steps", "6", "--save-every", "2", "--checkpoint-path", path, "--resume", "auto"], max_restarts=1)
    out, err = p.communicate(timeout=420)
    assert p.returncode == 0, err[-3000:]
    assert os.path.exists(marker), "the fault was never injected"
    assert "[pdt fault] rank 1 exiting at step 3" in err
    assert "Resumed from" in out and "step 2" in out, out[-2000:]
    side = torch.load(path + ".resume", weights_only=True)
    assert side["step"] == 6  # continuity: 2 steps before the crash survive, 4 are (re)done after the restart
    a = torch.load(ref, weights_only=True)
    b = torch.load(path, weights_only=True)
    assert a.keys() == b.keys()
    for k in a:
        if a[k].is_floating_point():
            assert torch.allclose(a[k], b[k], atol=1e-6), k
        else:
            assert torch.equal(a[k], b[k]), k


def test_stop_signal_to_one_rank_stops_all_ranks_at_the_same_step(tmp_path):
    """SIGUSR1 reaches rank 1 only.  The ranks agree on the stop, leave the loop together and the graceful
    checkpoint is written (a rank leaving alone would block the other in its next bucket collective)."""
    path = str(tmp_path / "stop.pt")
    pidfile = str(tmp_path / "rank1.pid")
    p = _launch(_port(), {"PDT_TEST_PIDFILE_RANK1": pidfile},
                ["--epochs", "400", "--checkpoint-path", path, "--log-every", "1"])
    try:
        deadline = time.time() + 240
        while not os.path.exists(pidfile) and time.time() < deadline:
            time.sleep(0.2)
            assert p.poll() is None, p.stderr.read()[-3000:]
        assert os.path.exists(pidfile), "rank 1 never reached its training loop"
        time.sleep(2.0)
        os.kill(int(open(pidfile).read()), signal.SIGUSR1)
        out, err = p.communicate(timeout=240)
    finally:
        if p.poll() is None:
            p.kill()
    assert p.returncode == 0, err[-3000:]
    assert out.count("Training complete") == 2
    side = torch.load(path + ".resume", weights_only=True)
    assert 0 < side["step"] < 400 * 12  # stopped early, on a step every rank agreed on


@pytest.mark.parametrize("stage,resumed_step", [("link_model", 4), ("model", 2)])
def test_writer_killed_inside_a_save_leaves_a_pair_the_restarted_job_resumes_from(tmp_path, stage, resumed_step):
    """Rank 0 dies inside its second save (step 4): after the main model file has been re-pointed but before its
    sidecar (`link_model`: the main pair is torn, generation 4 is complete), or after the generation's model file and
    before its sidecar (`model`: generation 4 is unusable, the main pair still is generation 2).  The launcher starts
    fresh workers, `--resume auto` finds the newest intact pair, and the job ends with the parameters of a run that was
    never interrupted."""
    ref = str(tmp_path / "ref.pt")
    p = _launch(_port(), {}, ["--steps", "6", "--checkpoint-path", ref])
    out, err = p.communicate(timeout=300)
    assert p.returncode == 0, err[-3000:]

    path = str(tmp_path / "job.pt")
    marker = str(tmp_path / "fault_fired")
    p = _launch(_port(), {"PDT_FAULT_INJECT_SAVE": f"2:{stage}", "PDT_FAULT_INJECT_ONCE_FILE": marker},
                ["--steps", "6", "--save-every", "2", "--checkpoint-path", path, "--resume", "auto"], max_restarts=1)
    out, err = p.communicate(timeout=420)
    assert p.returncode == 0, err[-3000:]
    assert os.path.exists(marker), "the fault was never injected"
    assert f"dying in save #2 after stage '{stage}'" in err
    assert "Resumed from" in out and f"step {resumed_step}" in out, out[-2000:]
    if stage == "link_model":
        assert "is torn or missing; using the intact generation" in out, out[-2000:]
    side = torch.load(path + ".resume", weights_only=True)
    assert side["step"] == 6
    a = torch.load(ref, weights_only=True)
    b = torch.load(path, weights_only=True)
    assert a.keys() == b.keys()
    for k in a:
        if a[k].is_floating_point():
            assert torch.allclose(a[k], b[k], atol=1e-6), k
        else:
            assert torch.equal(a[k], b[k]), k
