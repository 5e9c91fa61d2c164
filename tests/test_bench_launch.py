"""bench.py --gpus N without a launcher around it (the form the driver uses): it must start its own
N ranks as fresh child processes before anything touches the GPU and relay ONE JSON line."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env(**extra):
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env.update(extra)
    return env


def test_self_launch_without_a_gpu_fails_in_the_ranks_not_in_the_launcher():
    """No GPU here: the two child ranks must be started (and refuse, loudly: no CPU fallback); the
    parent relays their failure instead of demanding an external torchrun."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    p = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--workload", "bb", "--cpu-pivots", "0"],
                       env=_env(), capture_output=True, text=True, timeout=300)
    assert p.returncode != 0
    assert "needs an MI355X" in p.stderr
    assert "must be launched with" not in p.stderr
    assert not [ln for ln in p.stdout.splitlines() if ln.startswith("{")]


@pytest.mark.gpu
def test_self_launch_two_ranks_rehearsal_on_one_gpu():
    """`python bench.py --gpus 2 --workload bb` un-wrapped, in the one-GPU rehearsal mode (both
    ranks share GPU 0, the library's collectives carried over gloo): one JSON line, two ranks."""
    p = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--workload", "bb", "--bb-vars", "48",
                        "--bb-cons", "6", "--bb-levels", "5", "--cpu-pivots", "0"],
                       env=_env(LPR_BENCH_SHARED_GPU="1"), capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and "rehearsal" in out
    assert out["config"]["nodes_processed"] > 1 and out["value"] > 0
