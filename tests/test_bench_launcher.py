"""bench.py --gpus N starts N ranks itself (CPU checks: no GPU is touched)."""
import json
import os
import subprocess
import sys

from conftest import ROOT

BENCH = os.path.join(ROOT, "bench.py")


def _run(args, env=None, timeout=300):
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, BENCH, *args], capture_output=True, text=True, timeout=timeout, env=e)


def test_gpus_2_starts_two_ranks_and_reduces_over_them():
    """Two child ranks with disjoint seeds, one gloo reduction, one JSON line from rank 0."""
    r = _run(["--gpus", "2", "--dry-run", "--workload", "C1", "--batch", "2", "--no-pipeline"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["global_batch"] == 4
    assert out["pixels"] == 4 * 512 * 512
    assert out["elapsed_max"] == 2.0          # rank 1 reports the longer time: MAX over ranks


def test_gpus_2_without_a_gpu_fails_in_both_ranks():
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("a GPU is present")
    r = _run(["--gpus", "2", "--workload", "C1", "--batch", "2", "--no-pipeline", "--parity", "none", "--no-cpu-baseline"])
    assert r.returncode != 0
    assert r.stderr.count("AssertionError: bench.py needs a HIP device") == 2, r.stderr[-3000:]
    assert "rank 0 exited" in r.stderr and "rank 1 exited" in r.stderr
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


def test_world_size_that_contradicts_gpus_is_refused():
    r = _run(["--gpus", "2", "--dry-run", "--workload", "C1", "--batch", "2"], env={"WORLD_SIZE": "1", "RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr
