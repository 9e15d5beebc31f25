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


def test_gpus_8_dry_run_places_every_rank_on_its_gpus_node_and_gathers_per_rank_figures(tmp_path):
    """Eight ranks over gloo on a made-up host: two NUMA nodes, four GPUs each.  Every rank
    binds itself to CPUs of its own GPU's node, and rank 0 prints the per-rank arrays in rank order."""
    from test_topology import make_sysfs
    cpus = sorted(os.sched_getaffinity(0))
    if len(cpus) < 2:
        import pytest
        pytest.skip("one CPU")
    half = len(cpus) // 2
    lists = {0: ",".join(map(str, cpus[:half])), 1: ",".join(map(str, cpus[half:]))}
    make_sysfs(str(tmp_path), [(0x10 + i, 0 if i < 4 else 1) for i in range(8)], lists)
    r = _run(["--gpus", "8", "--dry-run", "--workload", "C1", "--batch", "2"], env={"MJ_SYSFS_ROOT": str(tmp_path)})
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 8 and out["global_batch"] == 16 and out["elapsed_max"] == 8.0
    assert out["distinct_batches_per_gpu"] == 2            # fewer distinct batches per rank when there are several ranks
    assert out["numa_node_per_rank"] == [0, 0, 0, 0, 1, 1, 1, 1]
    for rk in range(8):
        node_cpus = set(cpus[:half]) if rk < 4 else set(cpus[half:])
        bound = set(out["cpus_bound_per_rank"][rk])
        assert bound and bound <= node_cpus, (rk, bound)
        assert out["n_cpus_per_rank"][rk] == len(bound)
    assert out["pinned_MPps_per_rank"] == [2000.0 + rk for rk in range(8)]
    assert out["resident_MPps_per_rank"] == [1000.0 + rk for rk in range(8)]
    assert out["h2d_GBps_per_rank"] == [50.0 + rk for rk in range(8)]
    assert out["sustained_MPps_per_rank"] == [3000.0 + rk for rk in range(8)]
    assert out["images_per_rank"] == [2] * 8
    assert out["placement_rank0"]["source"] == "kfd" and out["placement_rank0"]["applied"]
