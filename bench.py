#!/usr/bin/env python3
"""Benchmark of the MI355X baseline-JPEG decode path.

    python bench.py --gpus N --steps K --warmup W

One step = one pass of the whole hot path (unstuff/restart scan -> Huffman decode ->
dequant+IDCT+upsample+colour) over one batch of synthetic 4K 4:2:0 baseline JPEGs with
one restart interval per MCU row (BASELINE config 3), compressed bytes already resident
in HBM, RGB24 left in HBM.  N > 1: one process per GPU (torch.distributed, backend
"nccl" = RCCL), each rank decodes its own batch (images are independent: no data-path
collective); RCCL only reduces the end-of-run counters and the max time.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import multiprocessing as mp
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (config key for synth.config_jpeg, width, height, sampling, samples/pixel)
    "C3": ("C3", 3840, 2160, "4:2:0", 1.5),
    "C2": ("C2", 1920, 1080, "4:4:4", 3.0),
    "C1": ("C1", 512, 512, "4:2:0", 1.5),
    "C4": ("C4", 7680, 4320, "4:2:0", 1.5),
}
HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak
N_CUS = 256


def _gen_one(args):
    config, seed, cache = args
    path = os.path.join(cache, f"{config}_{seed}.jpg") if cache else None
    if path and os.path.exists(path):
        with open(path, "rb") as f:
            return f.read()
    import jpeg_decoder_amd as J
    data = J.synth.config_jpeg(config, seed)
    if path:
        tmp = path + f".{os.getpid()}.tmp"
        with open(tmp, "wb") as f:
            f.write(data)
        os.replace(tmp, path)
    return data


def generate(config, seeds, workers):
    cache = os.environ.get("MJ_BENCH_CACHE", "/tmp/mj_bench_cache")
    try:
        os.makedirs(cache, exist_ok=True)
    except OSError:
        cache = None
    jobs = [(config, s, cache) for s in seeds]
    if workers <= 1 or len(jobs) == 1:
        return [_gen_one(j) for j in jobs]
    with mp.get_context("fork").Pool(min(workers, len(jobs))) as pool:
        return pool.map(_gen_one, jobs)


def _cpu_worker(args):
    kind, data, iters = args
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    from refbind import Oracle, Ref
    lib = Ref() if kind == "reference" else Oracle()
    t = lib.time_decode(data, iters)
    return t


def cpu_baseline(files, width, height, max_cores=16, iters=64):
    """The reference (oracle/_ref, compiled from its unmodified sources) -- or, where
    that library is absent, the CPU restatement -- timed headless on the host cores:
    one process per core, each decoding its own image `iters` times (decode + colour,
    file already in memory).  64 decodes of a 4K image per core are about ten seconds."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    from refbind import Oracle, Ref
    kind = "reference" if Ref.available() else ("port" if Oracle.available() else None)
    if kind is None:
        return None
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, max_cores, len(files)))
    jobs = [(kind, files[i % len(files)], iters) for i in range(cores)]
    t0 = time.perf_counter()
    with mp.get_context("fork").Pool(cores) as pool:
        per = pool.map(_cpu_worker, jobs)
    wall = time.perf_counter() - t0
    if min(per) <= 0:
        return None
    mp_total = cores * iters * width * height / 1e6
    # aggregate rate of `cores` concurrent single-threaded decoders
    value = sum(iters * width * height / 1e6 / t for t in per)
    return {"value": round(value, 2), "unit": "MP/s", "cores": cores, "kind": kind,
            "per_core": round(value / cores, 2),
            "sample": f"{cores} processes x {iters} decodes of one {width}x{height} image each "
                      f"({mp_total:.0f} MP, pool wall {wall:.1f} s incl. start-up)"}


def measured_traffic(kernel, workload, batch):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 counter passes
    (profiles/traffic.json: FETCH_SIZE and WRITE_SIZE collected in separate runs of this
    benchmark; reads doubled as gfx950's FETCH_SIZE counts 128-byte requests as 64).  None
    when no measurement for this workload and batch size is on file."""
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
            t = json.load(f)
    except (OSError, ValueError):
        return None
    for e in t.get("entries", []):
        if e.get("kernel") == kernel and e.get("workload") == workload and e.get("batch") == batch:
            return int((2 * e["fetch_size_kb"] + e["write_size_kb"]) * 1024)
    return None


def parity_check(batch, files, idx=0):
    """RGB and coefficients of one image of the timed batch against the checker."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import numpy as np
    from refbind import Oracle, Ref
    lib = Ref() if Ref.available() else (Oracle() if Oracle.available() else None)
    if lib is None:
        return None
    want = lib.decode(files[idx])
    rgb = batch.rgb(idx)
    coefs = batch.coefs(idx)
    return {"image": idx, "checker": "reference" if isinstance(lib, Ref) else "port",
            "coef_blocks_mismatched": int((coefs != want.coefs).any(axis=1).sum()),
            "rgb_max_abs_err": int(np.abs(rgb.astype(int) - want.rgb.astype(int)).max())}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=64, help="images per GPU")
    ap.add_argument("--workload", default="C3", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--gen-workers", type=int, default=0,
                    help="processes for input generation (0 = auto; use 1 under rocprofv3: no forks)")
    ap.add_argument("--serial-entropy", action="store_true", help="diagnostic: lane-per-interval entropy kernel")
    ap.add_argument("--streams", type=int, default=1,
                    help="issue the batch as this many sub-batches on separate HIP streams (kernels of different "
                         "sub-batches overlap; per-kernel times then include the overlap)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    config, width, height, sampling, spp = WORKLOADS[args.workload]
    if args.workload == "C4":
        args.batch = min(args.batch, 4)

    # inputs first: worker processes are forked before anything touches the GPU
    try:
        host_cores = len(os.sched_getaffinity(0))
    except AttributeError:
        host_cores = os.cpu_count() or 1
    workers = args.gen_workers if args.gen_workers > 0 else max(1, min(16, host_cores // max(1, world)))
    from jpeg_decoder_amd.shard import reduce_run, shard_seeds
    seeds = shard_seeds(args.batch, rank, world)
    t0 = time.perf_counter()
    files = generate(config, seeds, workers)
    t_gen = time.perf_counter() - t0

    # the CPU leg also runs before the GPU is initialised (its workers are plain forks)
    cpu_base = None
    if world == 1 and not args.no_cpu_baseline:
        cpu_base = cpu_baseline(files, width, height)

    import torch
    import torch.distributed as dist
    import jpeg_decoder_amd as J
    from jpeg_decoder_amd import binding

    assert torch.cuda.is_available(), "bench.py needs a HIP device"
    torch.cuda.set_device(local_rank)
    if world > 1:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    flags = binding.MJ_TIMING | (binding.MJ_ENTROPY_SERIAL if args.serial_entropy else 0)
    n_streams = max(1, min(args.streams, len(files)))
    ctxs = [J.Context(local_rank) for _ in range(n_streams)]       # one HIP stream each
    t0 = time.perf_counter()
    batches = [ctxs[i].batch(files[i::n_streams], flags) for i in range(n_streams)]
    t_prepare_first = time.perf_counter() - t0      # host parse + table flattening + allocation + H2D of the compressed bytes
    # once more for the steady state: the context keeps its page-locked staging buffer from batch to batch
    for bt in batches:
        bt.close()
    t0 = time.perf_counter()
    batches = [ctxs[i].batch(files[i::n_streams], flags) for i in range(n_streams)]
    t_prepare = time.perf_counter() - t0

    class _Group:
        """The sub-batches of one step, driven together."""

        def decode(self):
            for bt in batches:
                bt.decode()           # asynchronous on the sub-batch's own stream

        def reset_timing(self):
            for bt in batches:
                bt.reset_timing()

        def kernel_ms(self):
            tot = {}
            for bt in batches:
                for k, (ms, n) in bt.kernel_ms().items():
                    a = tot.get(k, (0.0, 0))
                    tot[k] = (a[0] + ms, a[1] + n)
            return tot

        def entropy_stats(self):
            tot = {}
            for bt in batches:
                for k, v in bt.entropy_stats().items():
                    tot[k] = max(tot.get(k, 0), v) if k.endswith("_max") else tot.get(k, 0) + v
            return tot

        def close(self):
            for bt in batches:
                bt.close()
            for c in ctxs:
                c.close()

    batch = _Group()

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        batch.decode()
    barrier()
    batch.reset_timing()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        batch.decode()
    barrier()
    elapsed = time.perf_counter() - t0

    kms = batch.kernel_ms()
    stats = batch.entropy_stats()
    pixels_rank = args.batch * width * height
    comp_bytes_rank = sum(len(f) for f in files)
    knames = sorted(kms)
    tot = reduce_run(pixels_rank, args.batch, comp_bytes_rank, 0.0, elapsed,
                     [kms[k][0] / max(kms[k][1], 1) for k in knames], device="cuda")
    total_pixels, total_images, total_bytes = tot.pixels, tot.images, tot.comp_bytes
    elapsed_max = tot.elapsed_max
    kernel_ms_max = dict(zip(knames, tot.kernel_ms_max))

    if rank == 0:
        value = total_pixels * args.steps / elapsed_max / 1e6
        ms_step = 1e3 * elapsed_max / args.steps
        # dominant HBM kernel: dequant + IDCT + upsample + colour.  Algorithmic bytes =
        # 2*S B/px of int16 coefficients read + 3 B/px of RGB written (6 B/px for 4:2:0).
        k_ms, k_n = kms["idct_colour"]
        k_avg_ms = k_ms / max(k_n, 1)
        alg_bytes = (2.0 * spp + 3.0) * pixels_rank / n_streams     # one launch covers one sub-batch
        achieved = alg_bytes / (k_avg_ms * 1e-3) / 1e9 if k_avg_ms > 0 else 0.0
        e_ms, e_n = kms["entropy"]
        f_ms, f_n = kms["entropy_fix"]
        e_avg_ms = (e_ms + f_ms) / max(e_n, 1)
        kernels = {name: {"avg_ms": round(ms / max(n, 1), 4), "launches": n} for name, (ms, n) in kms.items()}
        out = {
            "metric": "decoded megapixels/sec, 4K 4:2:0 baseline JPEG" if args.workload == "C3"
                      else f"decoded megapixels/sec, {args.workload}",
            "value": round(value, 1),
            "unit": "MP/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_step, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "int16 coefficients; f32 storage with f64 intermediates in the IDCT and colour stages; u8 out",
            "data": "synthetic",
            "config": {"workload": f"{args.workload}: batch of {args.batch} {width}x{height} {sampling} q85 baseline "
                                   f"JPEGs per GPU, DRI = one MCU row" if args.workload in ("C2", "C3") else
                                   f"{args.workload}: {args.batch} x {width}x{height} {sampling} q85, no restart markers",
                       "images_per_gpu": args.batch, "global_batch": int(total_images),
                       "compressed_MB_per_gpu": round(comp_bytes_rank / 1e6, 1),
                       "bits_per_pixel": round(8.0 * comp_bytes_rank / pixels_rank, 3),
                       "parallelism": f"image-sharded x{world}, no data-path collective",
                       "entropy_mode": "serial (diagnostic)" if args.serial_entropy else "subsequence-sync",
                       "streams": n_streams},
            "roofline": {"kernel": "mj_k_idct_colour", "bound": "hbm", "achieved": round(achieved, 1),
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                         "traffic": measured_traffic("mj_k_idct_colour", args.workload, args.batch // n_streams),
                         "algorithmic_bytes_per_launch": int(alg_bytes), "avg_launch_ms": round(k_avg_ms, 4)},
            "entropy": {"kernel": "mj_k_entropy_sync (+fix/write)", "avg_launch_ms": round(e_avg_ms, 4),
                        "decoded_Gbit_per_s": round(8.0 * comp_bytes_rank / n_streams / (e_avg_ms * 1e-3) / 1e9, 2) if e_avg_ms else None,
                        "Gbit_per_s_per_CU": round(8.0 * comp_bytes_rank / n_streams / (e_avg_ms * 1e-3) / 1e9 / N_CUS, 4) if e_avg_ms else None,
                        "sync_rounds_avg": round(stats["rounds_sum"] / max(stats["workgroups"], 1), 2),
                        "sync_rounds_max": stats["rounds_max"], "sequences": stats["sequences"],
                        "subsequences": stats["subsequences"],
                        "redecoded_fraction": round(stats["round_tasks"] / max(stats["subsequences"], 1), 3)},
            "end_to_end_hbm": {"algorithmic_bytes_per_px": round(8.0 * comp_bytes_rank / pixels_rank / 8 + 4 * spp + 3, 3),
                               "frac_of_peak": round((comp_bytes_rank / pixels_rank + 4 * spp + 3) * value * 1e6 / world
                                                     / (HBM_PEAK_GBS * 1e9), 4)},
            "kernels": kernels,
            "kernels_max_over_ranks_ms": {k: round(v, 4) for k, v in kernel_ms_max.items()},
            "host": {"generate_s": round(t_gen, 2), "prepare_h2d_s": round(t_prepare, 4), "prepare_first_batch_s": round(t_prepare_first, 4),
                     "MPps_incl_prepare": round(pixels_rank / 1e6 / (t_prepare + ms_step * 1e-3), 1)},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["parity"] = parity_check(batches[0], files, 0)
        batch.close()
        if cpu_base is not None:
            out["cpu_baseline"] = cpu_base
            out["gpu_over_cpu"] = round(value / cpu_base["value"], 1)
        print(json.dumps(out), flush=True)
    else:
        batch.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
