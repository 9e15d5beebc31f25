"""N > 1 path on CPU: two processes over gloo exercise the same sharding and end-of-run
reduction the benchmark runs over RCCL (one process per GPU, no data-path collective)."""
import os
import socket

import pytest
import torch.multiprocessing as mp

import jpeg_decoder_amd as J
from jpeg_decoder_amd.shard import reduce_run, shard_indices, shard_seeds


def test_shard_indices_partition():
    for n in (0, 1, 7, 64):
        for world in (1, 2, 3, 8):
            parts = [shard_indices(n, r, world) for r in range(world)]
            flat = sorted(i for p in parts for i in p)
            assert flat == list(range(n))
            assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1
    with pytest.raises(ValueError):
        shard_indices(4, 2, 2)


def test_shard_seeds_disjoint_and_world_independent():
    for world in (1, 2, 4, 8):
        allseeds = [s for r in range(world) for s in shard_seeds(64, r, world)]
        assert allseeds == list(range(64 * world))
    with pytest.raises(ValueError):
        shard_seeds(4, -1, 2)


def test_reduce_run_single_process_is_identity():
    t = reduce_run(100.0, 2.0, 30.0, 0.0, 0.5, [1.0, 2.0])
    assert (t.pixels, t.images, t.comp_bytes, t.mismatches, t.elapsed_max) == (100.0, 2.0, 30.0, 0.0, 0.5)
    assert t.kernel_ms_max == [1.0, 2.0]
    assert t.megapixels_per_s(4) == pytest.approx(100.0 * 4 / 0.5 / 1e6)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _rank_main(rank, world, port, batch, out):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # every rank builds its own inputs from its own seeds and counts what it holds
        seeds = shard_seeds(batch, rank, world)
        files = [J.synth.encode_jpeg(J.synth.synth_image(64, 48, seed=s), "4:2:0", 85, 4) for s in seeds]
        pixels = float(batch * 64 * 48)
        comp = float(sum(len(f) for f in files))
        # rank 1 is "slower" and reports one mismatch: MAX and SUM must both show it
        elapsed = 0.25 * (rank + 1)
        tot = reduce_run(pixels, float(batch), comp, float(rank), elapsed, [1.0 + rank, 3.0 - rank])
        out.put((rank, seeds, comp, tot))
    except Exception as e:   # the parent must not wait for a result that will not come
        out.put((rank, None, 0.0, repr(e)))
        raise
    finally:
        dist.barrier()
        dist.destroy_process_group()


def test_two_ranks_gloo_reduction():
    world, batch = 2, 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rank_main, args=(r, world, port, batch, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted((q.get(timeout=120) for _ in range(world)), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ss is not None for _, ss, _, _ in got), got
    seeds = [s for _, ss, _, _ in got for s in ss]
    assert seeds == list(range(world * batch))            # disjoint inputs, nothing shared
    comp_sum = sum(c for _, _, c, _ in got)
    for _, _, _, tot in got:                               # every rank sees the same totals
        assert tot.pixels == world * batch * 64 * 48
        assert tot.images == world * batch
        assert tot.comp_bytes == comp_sum
        assert tot.mismatches == 1.0
        assert tot.elapsed_max == 0.5                      # the slowest rank sets the job's time
        assert tot.kernel_ms_max == [2.0, 3.0]
        assert tot.megapixels_per_s(10) == pytest.approx(world * batch * 64 * 48 * 10 / 0.5 / 1e6)
