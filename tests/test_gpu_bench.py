"""GPU: the benchmark's contract on one GPU, at a small size: the headline value is the
pinned-host figure, the rank bound itself to CPUs before it allocated page-locked memory, the
per-rank arrays, the sustained leg, and where the traffic figure comes from."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_bench_line_on_one_gpu():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "C1", "--batch", "16", "--steps", "4",
                        "--warmup", "2", "--sustain-seconds", "0.5", "--no-cpu-baseline", "--no-singles", "--gen-workers", "2"],
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    # the headline is the page-locked-host -> HBM rate; the resident rate has its own object
    assert out["value"] == out["pinned_host_to_hbm"]["value"] and out["ms_per_step"] == out["pinned_host_to_hbm"]["ms_per_step"]
    assert out["resident"]["value"] > 0 and out["roofline"]["measured_in"] == "resident region"
    assert out["parity_mismatches_all_ranks"] == 0 and out["parity"]["images_checked"] == 16
    # placement: from the PCI address the library reports, applied before the first page-locked buffer
    p0 = out["host"]["placement_rank0"]
    assert p0["source"].startswith("pci") and p0["applied"] and p0["bdf"] and p0["n_cpus"] >= 1
    assert out["affinity_before_pinned_alloc_per_rank"] == [1]
    assert len(out["numa_node_per_rank"]) == 1 and out["images_per_rank"] == [16]
    for k in ("pinned_MPps_per_rank", "resident_MPps_per_rank", "h2d_GBps_per_rank", "sustained_MPps_per_rank"):
        assert len(out[k]) == 1 and out[k][0] > 0, k
    assert abs(out["pinned_MPps_per_rank"][0] - out["value"]) <= 0.02 * out["value"]
    s = out["sustained"]
    assert s["seconds"] >= 0.5 and s["submits"] >= 3 and s["MPps"] > 0 and 0.2 < s["ratio_to_value"] < 5
    # a traffic figure only with its provenance; C1 x 16 has no counter pass on file
    assert out["roofline"]["traffic"] is None and "why_null" in out["roofline"]["traffic_source"]


def test_two_ranks_rehearsed_on_one_gpu():
    """The N-rank flow end to end on the one GPU of the box: two ranks (started by bench.py
    itself) generate disjoint inputs, place themselves on disjoint CPUs, decode on GPU 0 and
    reduce over gloo; rank 0 prints the job's line with the per-rank arrays in rank order."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rehearse-on-one-gpu", "--workload", "C1",
                        "--batch", "8", "--steps", "3", "--warmup", "1", "--sustain-seconds", "0.3", "--gen-workers", "2"],
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["global_batch"] == 16 and "REHEARSAL" in out["config"]["parallelism"]
    assert out["images_per_rank"] == [8, 8] and out["affinity_before_pinned_alloc_per_rank"] == [1, 1]
    assert out["parity_mismatches_all_ranks"] == 0
    assert all(v > 0 for k in ("pinned_MPps_per_rank", "resident_MPps_per_rank", "sustained_MPps_per_rank") for v in out[k])
    # the job's rate is all ranks' pixels over the slowest rank's time: no more than the sum of the ranks' own rates
    assert out["value"] <= 1.02 * sum(out["pinned_MPps_per_rank"])
    assert "cpu_baseline" not in out              # rank 0 of one GPU only


def test_rccl_reduction_with_one_rank_on_the_gpu():
    """The end-of-run reduction and the per-rank gather over RCCL (backend "nccl") on device
    tensors: a world of one rank on the box's GPU (more ranks than GPUs is not something RCCL
    accepts); what N ranks run is the same two collectives."""
    code = r'''
import os, sys, json
sys.path.insert(0, %r)
import torch, torch.distributed as dist
from jpeg_decoder_amd.shard import reduce_run, gather_ranks
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29731", RANK="0", WORLD_SIZE="1")
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
t = reduce_run(100.0, 2.0, 30.0, 1.0, 0.5, [1.0, 2.0], device="cuda")
rows = gather_ranks([7.0, 8.0, 9.0], device="cuda")
# world of one: force the collectives themselves as well
x = torch.tensor([1.0, 2.0], dtype=torch.float64, device="cuda")
dist.all_reduce(x, op=dist.ReduceOp.SUM)
out = [torch.empty_like(x)]
dist.all_gather(out, x)
dist.barrier()
dist.destroy_process_group()
print(json.dumps({"pixels": t.pixels, "elapsed": t.elapsed_max, "kernels": t.kernel_ms_max, "rows": rows, "x": x.tolist(), "g": out[0].tolist()}))
''' % ROOT
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert out == {"pixels": 100.0, "elapsed": 0.5, "kernels": [1.0, 2.0], "rows": [[7.0, 8.0, 9.0]], "x": [1.0, 2.0], "g": [1.0, 2.0]}
