#!/usr/bin/env python3
"""Benchmark of the MI355X baseline-JPEG decode path.

    python bench.py --gpus N --steps K --warmup W

One step = one pass of the whole hot path (unstuff/restart scan -> Huffman decode ->
dequant+IDCT+upsample+colour) over one batch of synthetic 4K 4:2:0 baseline JPEGs with
one restart interval per MCU row (BASELINE config 3), RGB24 left in HBM.  Two timed
regions, both K steps after W warm-up steps:

  value                 compressed bytes already resident in HBM when the clock starts;
  pinned_host_to_hbm    the clock starts with the files in page-locked host memory: every
                        step parses its batch's headers on the host, copies the
                        entropy-coded bytes over PCIe and decodes, pipelined over a few
                        device slots (mj_pipe_*), cycling through D distinct batches.

N > 1: one process per GPU (torch.distributed, backend "nccl" = RCCL), each rank decodes
its own batches (images are independent: no data-path collective); RCCL only reduces the
end-of-run counters and the max time.  When WORLD_SIZE is not set and N > 1 this script
starts the N ranks itself, as child processes, before anything touches a GPU.

Prints ONE JSON line on rank 0.
"""
import argparse
import hashlib
import json
import multiprocessing as mp
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (width, height, sampling, samples/pixel, description)
    "C3": (3840, 2160, "4:2:0", 1.5, "3840x2160 4:2:0 q85 baseline JPEGs, DRI = one MCU row"),
    "C2": (1920, 1080, "4:4:4", 3.0, "1920x1080 4:4:4 q85 baseline JPEGs, DRI = one MCU row"),
    "C1": (512, 512, "4:2:0", 1.5, "512x512 4:2:0 q85 baseline JPEGs, no restart markers"),
    "C4": (7680, 4320, "4:2:0", 1.5, "7680x4320 4:2:0 q85 baseline JPEGs, no restart markers"),
    "C5": (None, None, "mixed", None, "640x480 and 1920x1080, grey / 4:2:2 / 4:2:0, q 50/75/95, standard and optimised "
                                      "Huffman tables, with and without DRI"),
}
HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak
N_CUS = 256


def c5_spec(seed):
    """BASELINE config 5, image `seed` of the mix: (width, height, sampling, quality, restart
    interval, grey, optimised tables)."""
    w, h = ((640, 480), (1920, 1080))[seed % 2]
    mode = ("grey", "4:2:2", "4:2:0")[(seed // 2) % 3]
    q = (50, 75, 95)[(seed // 6) % 3]
    optimise = (seed // 18) % 2 == 1
    grey = mode == "grey"
    samp = "4:4:4" if grey else mode
    hs = 1 if grey else 2
    ri = 0 if (seed // 36) % 2 == 0 else (w + 8 * hs - 1) // (8 * hs)
    return w, h, samp, q, ri, grey, optimise


def make_jpeg(workload, seed):
    import jpeg_decoder_amd as J
    if workload != "C5":
        return J.synth.config_jpeg(workload, seed)
    w, h, samp, q, ri, grey, optimise = c5_spec(seed)
    return J.synth.encode_jpeg(J.synth.synth_image(w, h, seed, grey=grey), samp, q, ri, optimise=optimise)


def pixels_of(workload, seed):
    if workload != "C5":
        return WORKLOADS[workload][0] * WORKLOADS[workload][1]
    w, h = c5_spec(seed)[:2]
    return w * h


def samples_of(workload, seed):
    """Samples (coefficients) per image: what the IDCT kernel reads, two bytes each."""
    if workload != "C5":
        return WORKLOADS[workload][3] * pixels_of(workload, seed)
    w, h, samp, _, _, grey, _ = c5_spec(seed)
    return w * h * (1.0 if grey else {"4:2:2": 2.0, "4:2:0": 1.5}[samp])


def _gen_one(args):
    workload, seed, cache = args
    path = os.path.join(cache, f"{workload}_{seed}.jpg") if cache else None
    if path and os.path.exists(path):
        with open(path, "rb") as f:
            return f.read()
    data = make_jpeg(workload, seed)
    if path:
        tmp = path + f".{os.getpid()}.tmp"
        with open(tmp, "wb") as f:
            f.write(data)
        os.replace(tmp, path)
    return data


def generate(workload, seeds, workers):
    # generated files are kept only where the caller names a directory (the profiling passes of
    # profiles/collect.sh run the benchmark five times): a default batch is 2 GB of JPEGs per rank
    cache = os.environ.get("MJ_BENCH_CACHE") or None
    if cache:
        try:
            os.makedirs(cache, exist_ok=True)
        except OSError:
            cache = None
    jobs = [(workload, s, cache) for s in seeds]
    if workers <= 1 or len(jobs) == 1:
        return [_gen_one(j) for j in jobs]
    with mp.get_context("fork").Pool(min(workers, len(jobs))) as pool:
        return pool.map(_gen_one, jobs, chunksize=1)


def _in_child(fn, *args):
    """fn(*args) in a forked child; None if the child died (the compiled reference walks off
    its buffers on some streams, see REFERENCE_UNUSABLE and reference_survives)."""
    import pickle
    r, w = os.pipe()
    pid = os.fork()
    if pid == 0:
        code = 1
        try:
            os.close(r)
            with os.fdopen(w, "wb") as f:
                pickle.dump(fn(*args), f)
            code = 0
        finally:
            os._exit(code)
    os.close(w)
    with os.fdopen(r, "rb") as f:
        blob = f.read()
    _, status = os.waitpid(pid, 0)
    if status != 0 or not blob:
        return None
    return pickle.loads(blob)


def _cpu_worker(args):
    kind, data, iters = args
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    from refbind import Oracle, Ref
    if kind == "reference":
        t = _in_child(lambda: Ref().time_decode(data, iters))
        return t if t is not None else -1.0
    return Oracle().time_decode(data, iters)


# Workloads the compiled reference cannot be the checker or the CPU baseline for: C4 (its
# 16-bit MCU counter misplaces blocks beyond 65,535 MCUs, decode.h:15) and C5 (at q95 with
# restart markers it walks into the marker bytes and crashes: its restart check looks at
# a partly consumed FF byte, bitstream.c:84-176).  The CPU restatement stands in.
REFERENCE_UNUSABLE = ("C4", "C5")


def _cpu_pool_init(cpus):
    try:
        os.sched_setaffinity(0, cpus)       # the rank's own placement does not bind the baseline's workers
    except (AttributeError, OSError):
        pass


def cpu_baseline(files, pixels, workload, cpus, seconds=10.0):
    """The reference (oracle/_ref, compiled from its unmodified sources) -- or, where
    that library is absent or cannot decode the workload, the CPU restatement -- timed
    headless on the host cores: one process per core the job may use (`cpus`), each decoding
    its own image repeatedly (decode + colour, file already in memory), about `seconds` of
    CPU work per core."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    from refbind import Oracle, Ref
    kind = "reference" if Ref.available() and workload not in REFERENCE_UNUSABLE[1:] else ("port" if Oracle.available() else None)
    if kind is None:
        return None
    cores = max(1, min(len(cpus), len(files)))
    if kind == "reference":
        # only files the compiled reference gets through (it dies on some streams with restart
        # markers, bitstream.c:84-176): the first `cores` such files
        ok = []
        for i, f in enumerate(files):
            if _in_child(lambda: Ref().decode(f, want_tap=False) is not None):
                ok.append(i)
                if len(ok) == cores:
                    break
        if not ok:
            kind, ok = ("port", list(range(len(files)))) if Oracle.available() else (None, [])
        if kind is None:
            return None
        files, pixels = [files[i] for i in ok], [pixels[i] for i in ok]
    # about 50 MP/s per core: size every core's sample to `seconds`
    iters = [min(4096, max(1, int(seconds * 50e6 / max(pixels[i % len(files)], 1)))) for i in range(cores)]
    jobs = [(kind, files[i % len(files)], iters[i]) for i in range(cores)]
    t0 = time.perf_counter()
    with mp.get_context("fork").Pool(cores, initializer=_cpu_pool_init, initargs=(set(cpus),)) as pool:
        per = pool.map(_cpu_worker, jobs, chunksize=1)
    wall = time.perf_counter() - t0
    if min(per) <= 0:
        return None
    mp_total = sum(iters[i] * pixels[i % len(files)] for i in range(cores)) / 1e6
    # aggregate rate of `cores` concurrent single-threaded decoders
    value = sum(iters[i] * pixels[i % len(files)] / 1e6 / per[i] for i in range(cores))
    return {"value": round(value, 2), "unit": "MP/s", "cores": cores, "kind": kind,
            "per_core": round(value / cores, 2), "host_cores": os.cpu_count(), "cores_allowed": len(cpus),
            "sample": f"{cores} processes, each decoding one image of the workload {min(iters)}-{max(iters)} times "
                      f"({mp_total:.0f} MP in all, pool wall {wall:.1f} s incl. start-up)"}


def _digest_worker(args):
    """(sha256 of coefficients, sha256 of RGB, checker that produced them).  The compiled
    reference decodes in a child of its own: where a restart marker follows a partly consumed
    data FF (FF 00 FF Dn) it takes the FF for the marker, lands on the marker bytes and dies
    (bitstream.c:84-151).  The CPU restatement then stands in for that file -- as
    "reference_prefix" when every block the reference's coefficient tap recorded before it
    died (in memory shared with the child) equals the restatement's, else as "port"."""
    kind, data = args
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    from refbind import Oracle, Ref

    def digests(lib):
        want = lib.decode(data)
        return hashlib.sha256(want.coefs.tobytes()).hexdigest(), hashlib.sha256(want.rgb.tobytes()).hexdigest()
    if kind == "reference":
        got = _in_child(lambda: digests(Ref()))
        if got is not None:
            return got[0], got[1], "reference"
    want = Oracle().decode(data)
    by = "port"
    if kind == "reference":
        t = Ref().decode_traced(data, want.coefs.shape[0], 1)
        n = t.coefs.shape[0]
        if t.signal and 0 < n <= want.coefs.shape[0] and (t.coefs == want.coefs[:n]).all():
            by = "reference_prefix"
    return hashlib.sha256(want.coefs.tobytes()).hexdigest(), hashlib.sha256(want.rgb.tobytes()).hexdigest(), by


def checker_digests(files, workers, workload):
    """SHA-256 of the checker's coefficients and RGB for every file (forked workers: call
    before the GPU is initialised).  The compiled reference where it is present and can
    decode the workload (see REFERENCE_UNUSABLE), else the CPU restatement."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    from refbind import Oracle, Ref
    if Ref.available() and workload not in REFERENCE_UNUSABLE:
        kind = "reference"
    elif Oracle.available():
        kind = "port"
    else:
        return None, None
    jobs = [(kind, f) for f in files]
    if workers <= 1:
        return kind, [_digest_worker(j) for j in jobs]
    with mp.get_context("fork").Pool(min(workers, len(jobs))) as pool:
        return kind, pool.map(_digest_worker, jobs, chunksize=1)


def checked_by(digests):
    """how many files each checker vouched for"""
    out = {}
    for d in digests:
        out[d[2]] = out.get(d[2], 0) + 1
    return out


def single_image_ms(ctx, data, flags, reps=30):
    """Steady-state milliseconds of one image on its own: batch set-up on the host (parse,
    flattening, layout, upload), and the decode from launch to the RGB being in HBM."""
    import torch
    t_create = t_decode = 0.0
    for i in range(reps + 3):
        t0 = time.perf_counter()
        b = ctx.batch([data], flags)
        t1 = time.perf_counter()
        b.decode()
        b.sync()
        t2 = time.perf_counter()
        b.close()
        if i >= 3:
            t_create += t1 - t0
            t_decode += t2 - t1
    torch.cuda.synchronize()
    return {"set_up_ms": round(1e3 * t_create / reps, 3), "decode_ms": round(1e3 * t_decode / reps, 3)}


def parity_of(batch, digests):
    """(images with differing coefficients, images with differing RGB) of a decoded batch."""
    bad_c = bad_r = 0
    for i, (dc, dr, _) in enumerate(digests):
        bad_c += hashlib.sha256(batch.coefs(i).tobytes()).hexdigest() != dc
        bad_r += hashlib.sha256(batch.rgb(i).tobytes()).hexdigest() != dr
    return bad_c, bad_r


def git_blob_sha(path):
    """The object name git gives the file's present contents (sha1 of "blob <size>\0" + bytes)."""
    try:
        with open(path, "rb") as f:
            data = f.read()
    except OSError:
        return None
    return hashlib.sha1(b"blob %d\0" % len(data) + data).hexdigest()


IDCT_KERNEL_SOURCE = os.path.join("jpeg-decoder_amd", "csrc", "mj_k_idct_colour.hip")


def measured_traffic(kernel, workload, batch):
    """(HBM bytes per launch of `kernel`, where the figure comes from) out of the committed
    rocprofv3 counter passes (profiles/traffic.json: FETCH_SIZE and WRITE_SIZE collected in
    separate runs of this benchmark; reads doubled as gfx950's FETCH_SIZE counts 128-byte
    requests as 64).  An entry counts only if it was measured on the kernel source as it is
    now (git blob name of mj_k_idct_colour.hip); otherwise (None, why)."""
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
            t = json.load(f)
    except (OSError, ValueError):
        return None, "profiles/traffic.json missing"
    blob = git_blob_sha(os.path.join(ROOT, IDCT_KERNEL_SOURCE))
    stale = None
    for e in t.get("entries", []):
        if e.get("kernel") == kernel and e.get("workload") == workload and e.get("batch") == batch:
            if e.get("kernel_blob") != blob:
                stale = e
                continue
            return (int((2 * e["fetch_size_kb"] + e["write_size_kb"]) * 1024),
                    {"file": "profiles/traffic.json", "summary": e.get("summary"), "measured_at_commit": e.get("measured_at_commit"),
                     "kernel_source": IDCT_KERNEL_SOURCE, "kernel_blob": blob})
    if stale is not None:
        return None, {"file": "profiles/traffic.json", "why_null": "the entry on file was measured on another version of the kernel source",
                      "entry_kernel_blob": stale.get("kernel_blob"), "kernel_blob": blob}
    return None, {"file": "profiles/traffic.json", "why_null": "no entry for this workload and batch size", "kernel_blob": blob}


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch(n, argv):
    """Start n ranks of this script as child processes (one per GPU) and wait for them.
    Nothing in this process has touched a GPU.  Returns the exit status for the job."""
    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), *argv], env=env))
    rc = 0
    pending = set(range(n))
    failed_at = None
    while pending:
        for r in sorted(pending):
            code = procs[r].poll()
            if code is None:
                continue
            pending.discard(r)
            if code != 0:
                print(f"bench.py: rank {r} exited with status {code}", file=sys.stderr, flush=True)
                rc = rc or (code if code > 0 else 1)
                failed_at = failed_at or time.monotonic()
        # a rank that died leaves the others waiting at a rendezvous: give them a moment, then end them
        if failed_at and pending and time.monotonic() - failed_at > 20:
            for r in pending:
                procs[r].terminate()
            for r in pending:
                try:
                    procs[r].wait(timeout=10)
                except subprocess.TimeoutExpired:
                    procs[r].kill()
            pending.clear()
        time.sleep(0.05)
    return rc


class SensorLog:
    """Shader clock and board power of the rank's GPU, sampled from sysfs on a thread of its
    own while a leg runs (jpeg-decoder_amd/topology.py: GpuSensors)."""

    def __init__(self, bdf, period=0.05):
        import threading
        from jpeg_decoder_amd.topology import GpuSensors
        self.sensors = GpuSensors(bdf) if bdf else None
        self.period = period
        self.sclk, self.power = [], []
        self._stop = threading.Event()
        self._thread = threading.Thread(target=self._run, daemon=True)

    def _run(self):
        while not self._stop.is_set():
            r = self.sensors.read()
            if "sclk_MHz" in r:
                self.sclk.append(r["sclk_MHz"])
            if "power_W" in r:
                self.power.append(r["power_W"])
            self._stop.wait(self.period)

    def __enter__(self):
        if self.sensors and (self.sensors.sclk_path or self.sensors.power_path):
            self._thread.start()
        return self

    def __exit__(self, *exc):
        self._stop.set()
        if self._thread.is_alive():
            self._thread.join(timeout=2)

    def summary(self):
        out = {}
        if self.sclk:
            out.update(sclk_MHz_min=min(self.sclk), sclk_MHz_max=max(self.sclk), sclk_samples=len(self.sclk))
        if self.power:
            out.update(power_W_max=round(max(self.power), 1), power_W_mean=round(sum(self.power) / len(self.power), 1))
        return out


# what every rank contributes to the per-rank arrays of the output line, in this order
RANK_FIELDS = ["resident_MPps", "pinned_MPps", "h2d_GBps", "numa_node", "n_cpus", "sustained_MPps", "affinity_before_pinned_alloc",
               "images"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=0,
                    help="images per GPU and step (default: 256 for C3 -- 14 GB of the 288 GB -- 4 for C4, 64 otherwise)")
    ap.add_argument("--workload", default="C3", choices=sorted(WORKLOADS))
    ap.add_argument("--distinct", type=int, default=0,
                    help="distinct batches per GPU the pipelined region cycles through (every step re-parses and re-uploads); "
                         "default 4 on one GPU, 2 per rank on several (input generation is what a rank spends its start-up on)")
    ap.add_argument("--depth", type=int, default=3, help="batches in flight in the pipelined region")
    ap.add_argument("--sustain-seconds", type=float, default=5.0,
                    help="after the timed regions, keep the pinned-host pipeline running for this long (0 = skip)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-singles", action="store_true", help="skip the single-image latency figures (single_image in the output)")
    ap.add_argument("--no-pipeline", action="store_true", help="skip the pinned-host -> HBM region (value is then the resident rate)")
    ap.add_argument("--no-affinity", action="store_true", help="leave the process's CPU affinity alone")
    ap.add_argument("--parity", default="all", choices=["all", "one", "none"],
                    help="check every image of the timed batch against the checker, one image, or nothing")
    ap.add_argument("--gen-workers", type=int, default=0,
                    help="processes for input generation (0 = auto; use 1 under rocprofv3: no forks)")
    ap.add_argument("--serial-entropy", action="store_true", help="diagnostic: lane-per-interval entropy kernel")
    ap.add_argument("--streams", type=int, default=1,
                    help="issue the resident batch as this many sub-batches on separate HIP streams (kernels of different "
                         "sub-batches overlap; per-kernel times then include the overlap)")
    ap.add_argument("--dry-run", action="store_true",
                    help="launcher, placement and reduction wiring only: no GPU, no decode, gloo instead of RCCL")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="every rank decodes on GPU 0 and the reductions go over gloo: the whole N-rank flow on a box with one GPU "
                         "(RCCL refuses two ranks on one device); the rates it prints are those of ranks sharing a GPU")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch(args.gpus, sys.argv[1:]))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: start one rank per GPU "
                 f"(torch.distributed.run --nproc-per-node {args.gpus}) or leave WORLD_SIZE unset")
    desc = WORKLOADS[args.workload][4]
    if args.batch <= 0:
        # a larger batch amortises the ends of the kernels, above all the serial tail of the entropy
        # synchronisation: C3 on one GPU 181 GP/s at 64 images, 196 at 128, 212 at 256
        args.batch = {"C3": 256, "C4": 4}.get(args.workload, 64)
    if args.workload == "C4":
        args.batch = min(args.batch, 4)
    if args.distinct <= 0:
        args.distinct = 4 if world == 1 else 2
    n_distinct = 1 if args.no_pipeline else max(1, args.distinct)

    # ---- placement: this rank onto the CPUs next to its GPU, before it forks workers and before
    # any page-locked allocation (first guess from the KFD topology: nothing touches the GPU yet)
    from jpeg_decoder_amd import topology
    from jpeg_decoder_amd.shard import gather_ranks, reduce_run, shard_seeds
    try:
        allowed0 = set(os.sched_getaffinity(0))
    except AttributeError:
        allowed0 = set(range(os.cpu_count() or 1))
    place = topology.plan(local_rank, local_world, allowed0)
    if not args.no_affinity:
        topology.apply(place)
    t_affinity = time.monotonic()

    # inputs: worker processes are forked before anything touches the GPU and inherit the placement
    workers = args.gen_workers if args.gen_workers > 0 else max(1, min(16, len(place.cpus) or 1))
    # batch d of rank r: seeds d * world * batch + (r * batch .. r * batch + batch - 1)
    seed_sets = [[d * world * args.batch + s for s in shard_seeds(args.batch, rank, world)] for d in range(n_distinct)]
    t0 = time.perf_counter()
    flat = generate(args.workload, [s for ss in seed_sets for s in ss], workers)
    file_sets = [flat[d * args.batch:(d + 1) * args.batch] for d in range(n_distinct)]
    t_gen = time.perf_counter() - t0
    files, seeds = file_sets[0], seed_sets[0]
    px_img = [pixels_of(args.workload, s) for s in seeds]
    pixels_rank = float(sum(px_img))
    samples_rank = float(sum(samples_of(args.workload, s) for s in seeds))
    comp_bytes_rank = sum(len(f) for f in files)

    def rank_arrays(rows):
        return {f"{name}_per_rank": [round(r[k], 1) if name.endswith(("MPps", "GBps")) else int(r[k]) for r in rows]
                for k, name in enumerate(RANK_FIELDS)}

    if args.dry_run:
        import torch.distributed as dist
        if world > 1:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        tot = reduce_run(pixels_rank, args.batch, comp_bytes_rank, 0.0, 1.0 + rank, [1.0])
        # stand-in rates that name their rank, so that the order of the gathered arrays is checked
        rows = gather_ranks([1000.0 + rank, 2000.0 + rank, 50.0 + rank, place.numa_node, len(place.cpus), 3000.0 + rank,
                             1.0, args.batch])
        placements = gather_ranks([float(c in place.cpus) for c in range(max(allowed0) + 1)])
        if rank == 0:
            out = {"dry_run": True, "n_gpus": world, "global_batch": int(tot.images), "pixels": tot.pixels,
                   "elapsed_max": tot.elapsed_max, "seeds_rank0": seeds[:4], "distinct_batches_per_gpu": n_distinct,
                   "gen_workers": workers, "placement_rank0": place.as_dict(),
                   "cpus_bound_per_rank": [[c for c, on in enumerate(row) if on] for row in placements]}
            out.update(rank_arrays(rows))
            print(json.dumps(out), flush=True)
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return

    # the CPU legs also run before the GPU is initialised (their workers are plain forks)
    cpu_base = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu_base = cpu_baseline(files, px_img, args.workload, allowed0)
    checker, digests = (None, None)
    if args.parity != "none":
        n_check = len(files) if args.parity == "all" else 1
        checker, digests = checker_digests(files[:n_check], workers, args.workload)

    import torch
    import torch.distributed as dist
    import jpeg_decoder_amd as J
    from jpeg_decoder_amd import binding

    assert torch.cuda.is_available(), "bench.py needs a HIP device"
    rehearsal = args.rehearse_on_one_gpu and world > 1
    dev_index = 0 if rehearsal else local_rank
    red_dev = "cpu" if rehearsal else "cuda"          # where the reductions' few numbers live: gloo reduces on the host
    torch.cuda.set_device(dev_index)
    # the placement once more, now from the PCI addresses the runtime reports for the local GPUs
    try:
        n_dev = torch.cuda.device_count()
        bdfs = [binding.device_pci_bus_id(0 if rehearsal else r) if (rehearsal or r < n_dev) else None for r in range(local_world)]
        sure = topology.plan(local_rank, local_world, allowed0, bdfs=bdfs, source="pci")
        if sure.source == "split":
            sure.source = "pci, split"
        if sorted(sure.cpus) != sorted(place.cpus) or not place.applied:
            if not args.no_affinity:
                topology.apply(sure)
        else:
            sure.applied = place.applied
        place = sure
        t_affinity = time.monotonic() if place.applied else t_affinity
    except RuntimeError as e:
        place.note = (place.note + "; " if place.note else "") + f"PCI address not available: {e}"
    if world > 1:
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    flags = binding.MJ_TIMING | (binding.MJ_ENTROPY_SERIAL if args.serial_entropy else 0)

    # ---- region 1: compressed bytes resident in HBM ---------------------------------
    n_streams = max(1, min(args.streams, len(files)))
    ctxs = [J.Context(dev_index) for _ in range(n_streams)]       # one HIP stream each
    t0 = time.perf_counter()
    batches = [ctxs[i].batch(files[i::n_streams], flags) for i in range(n_streams)]
    t_prepare_first = time.perf_counter() - t0      # host parse + table flattening + allocation + H2D of the compressed bytes
    for bt in batches:
        bt.close()
    t0 = time.perf_counter()
    batches = [ctxs[i].batch(files[i::n_streams], flags) for i in range(n_streams)]
    t_prepare = time.perf_counter() - t0            # steady state: the context keeps its buffers from batch to batch

    def decode_all():
        for bt in batches:
            bt.decode()           # asynchronous on the sub-batch's own stream

    for _ in range(args.warmup):
        decode_all()
    barrier()
    for bt in batches:
        bt.reset_timing()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        decode_all()
    barrier()
    elapsed = time.perf_counter() - t0

    kms = {}
    for bt in batches:
        for k, (ms, n) in bt.kernel_ms().items():
            a = kms.get(k, (0.0, 0))
            kms[k] = (a[0] + ms, a[1] + n)
    stats = {}
    for bt in batches:
        for k, v in bt.entropy_stats().items():
            stats[k] = max(stats.get(k, 0), v) if k.endswith("_max") else stats.get(k, 0) + v
    speculative = stats.get("workgroups", 0) == 0 and stats.get("subsequences", 0) > 0     # no workgroup ran the rounds
    mismatched = 0
    parity = None
    if digests is not None and n_streams == 1:
        bad_c, bad_r = parity_of(batches[0], digests)
        mismatched += bad_c + bad_r
        parity = {"checker": checker, "images_checked": len(digests), "checked_by": checked_by(digests),
                  "images_with_coefficient_mismatch": bad_c, "images_with_rgb_mismatch": bad_r,
                  "how": "SHA-256 of every image's coefficient blocks and RGB; the CPU restatement stands in for files on "
                         "which the compiled reference dies (reference_prefix: the blocks it tapped before it died "
                         "equal the restatement's)"}
    singles = None
    if rank == 0 and not args.no_singles:
        # latency of small work: one image alone (nothing to overlap its kernels with)
        singles = {"C1 (512x512 4:2:0, no restart markers)": single_image_ms(ctxs[0], J.synth.config_jpeg("C1", 0), 0),
                   "C4 (7680x4320 4:2:0, no restart markers)": single_image_ms(ctxs[0], J.synth.config_jpeg("C4", 0), 0, reps=10)}
    small = None
    if rank == 0 and not args.no_singles and n_streams == 1 and args.batch > 64:
        # the same decode on the first 64 images only (the batch size the configuration names)
        sb = ctxs[0].batch(files[:64], flags)
        for _ in range(3):
            sb.decode()
        torch.cuda.synchronize()
        sb.reset_timing()
        ts = time.perf_counter()
        for _ in range(10):
            sb.decode()
        torch.cuda.synchronize()
        ts = (time.perf_counter() - ts) / 10
        small = {"images": 64, "ms_per_step": round(1e3 * ts, 4), "MPps": round(sum(px_img[:64]) / ts / 1e6, 1),
                 "kernels_avg_ms": {k: round(ms / max(n, 1), 4) for k, (ms, n) in sb.kernel_ms().items()}}
        ik = small["kernels_avg_ms"].get("idct_colour")
        if ik:
            # the roofline kernel at this size: 2 bytes per coefficient sample read + 3 bytes per pixel written
            sbytes = 2.0 * sum(samples_of(args.workload, s) for s in seeds[:64]) + 3.0 * sum(px_img[:64])
            small["idct_colour_GBps"] = round(sbytes / (ik * 1e-3) / 1e9, 1)
            small["idct_colour_frac_of_hbm_peak"] = round(sbytes / (ik * 1e-3) / 1e9 / 8000.0, 4)
        sb.close()
    for bt in batches:
        bt.close()
    for c in ctxs:
        c.close()

    # ---- region 2: page-locked host memory -> RGB in HBM, pipelined ------------------
    pipe_res = None
    sustained = None
    t_pinned = None
    if not args.no_pipeline:
        ctx = J.Context(dev_index)
        t_pinned = time.monotonic()                  # the first page-locked input buffer is made here, after the placement
        pinned = [binding.PinnedFiles(fs) for fs in file_sets]
        depth = max(2, min(args.depth, 8))
        pipe = ctx.pipe(depth, flags)
        pix_sets = [float(sum(pixels_of(args.workload, s) for s in ss)) for ss in seed_sets]
        last = None
        for k in range(max(args.warmup, depth)):               # fills the slots: allocations happen here
            last = pipe.submit(pinned[k % n_distinct])
        last.sync()
        barrier()
        pipe.host_ms(reset=True)
        seen = {}
        t_host = 0.0
        px_done = 0.0
        t0 = time.perf_counter()
        for k in range(args.steps):
            th = time.perf_counter()
            last = pipe.submit(pinned[k % n_distinct])
            t_host += time.perf_counter() - th
            px_done += pix_sets[k % n_distinct]
            seen[last.h.value] = last
        last.sync()
        barrier()
        elapsed_pipe = time.perf_counter() - t0
        host_wait_ms, host_work_ms, _ = pipe.host_ms()
        pkms = {}
        for bt in seen.values():
            for name, (ms, n) in bt.kernel_ms().items():
                a = pkms.get(name, (0.0, 0))
                pkms[name] = (a[0] + ms, a[1] + n)
        # one more, untimed: the checked batch through the pipeline, every image compared
        pbad = None
        if digests is not None:
            chk = pipe.submit(pinned[0])
            pbad = parity_of(chk, digests)
            mismatched += sum(pbad)

        # ---- sustained leg: the same pipeline for seconds, not milliseconds -------------
        if args.sustain_seconds > 0:
            for bt in seen.values():
                bt.reset_timing()
            barrier()
            s_px, s_n = 0.0, 0
            with SensorLog(place.bdf) as sensors:
                t0 = time.perf_counter()
                while True:
                    last = pipe.submit(pinned[s_n % n_distinct])
                    s_px += pix_sets[s_n % n_distinct]
                    s_n += 1
                    if time.perf_counter() - t0 >= args.sustain_seconds and s_n >= depth:
                        break
                last.sync()
                barrier()
                s_elapsed = time.perf_counter() - t0
            sustained = {"elapsed": s_elapsed, "pixels": s_px, "submits": s_n, "sensors": sensors.summary()}

        # the copy alone, for scale: one batch's compressed bytes, page-locked -> HBM
        nbytes = sum(len(f) for f in file_sets[0])
        src = torch.empty(nbytes, dtype=torch.uint8).pin_memory()
        dst = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        dst.copy_(src, non_blocking=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(5):
            dst.copy_(src, non_blocking=True)
        e1.record()
        torch.cuda.synchronize()
        h2d_ms = e0.elapsed_time(e1) / 5
        pipe.close()
        for p in pinned:
            p.close()
        ctx.close()
        pipe_res = {"elapsed": elapsed_pipe, "pixels": px_done, "host_submit_ms": host_work_ms / args.steps,
                    "host_wait_ms": host_wait_ms / args.steps, "python_submit_ms": 1e3 * t_host / args.steps,
                    "h2d_alone_ms": h2d_ms, "h2d_GBps": nbytes / (h2d_ms * 1e-3) / 1e9, "depth": depth, "kernels": pkms,
                    "parity_bad": pbad}

    # ---- one reduction per region (SUM of the counters, MAX of the times) + the per-rank figures
    knames = sorted(kms)
    tot = reduce_run(pixels_rank, args.batch, comp_bytes_rank, float(mismatched), elapsed,
                     [kms[k][0] / max(kms[k][1], 1) for k in knames], device=red_dev)
    total_pixels, total_images = tot.pixels, tot.images
    elapsed_max = tot.elapsed_max
    kernel_ms_max = dict(zip(knames, tot.kernel_ms_max))
    ptot = stot = None
    if pipe_res is not None:
        ptot = reduce_run(pipe_res["pixels"], 0.0, 0.0, 0.0, pipe_res["elapsed"], [pipe_res["host_submit_ms"]], device=red_dev)
    if sustained is not None:
        stot = reduce_run(sustained["pixels"], float(sustained["submits"]), 0.0, 0.0, sustained["elapsed"], [0.0], device=red_dev)
    rows = gather_ranks([
        pixels_rank * args.steps / elapsed / 1e6,
        pipe_res["pixels"] / pipe_res["elapsed"] / 1e6 if pipe_res else 0.0,
        pipe_res["h2d_GBps"] if pipe_res else 0.0,
        place.numa_node, len(place.cpus),
        sustained["pixels"] / sustained["elapsed"] / 1e6 if sustained else 0.0,
        1.0 if (place.applied and t_pinned is not None and t_affinity < t_pinned) else 0.0,
        args.batch], device=red_dev)

    if rank == 0:
        resident_value = total_pixels * args.steps / elapsed_max / 1e6
        resident_ms = 1e3 * elapsed_max / args.steps
        if ptot is not None:
            value = ptot.pixels / ptot.elapsed_max / 1e6
            ms_step = 1e3 * ptot.elapsed_max / args.steps
        else:
            value, ms_step = resident_value, resident_ms
        # dominant HBM kernel: dequant + IDCT + upsample + colour.  Algorithmic bytes =
        # 2 B per coefficient read + 3 B/px of RGB written (6 B/px for 4:2:0).
        k_ms, k_n = kms["idct_colour"]
        k_avg_ms = k_ms / max(k_n, 1)
        alg_bytes = (2.0 * samples_rank + 3.0 * pixels_rank) / n_streams     # one launch covers one sub-batch
        achieved = alg_bytes / (k_avg_ms * 1e-3) / 1e9 if k_avg_ms > 0 else 0.0
        e_ms, e_n = kms["entropy"]
        f_ms, f_n = kms["entropy_fix"]
        e_avg_ms = (e_ms + f_ms) / max(e_n, 1)
        kernels = {name: {"avg_ms": round(ms / max(n, 1), 4), "launches": n} for name, (ms, n) in kms.items()}
        e2e_bytes_px = (comp_bytes_rank + 4.0 * samples_rank + 3.0 * pixels_rank) / pixels_rank
        traffic, traffic_source = measured_traffic("mj_k_idct_colour", args.workload, args.batch // n_streams)
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
            "value_measures": ("JPEG files in page-locked host memory -> RGB24 in HBM: host header parse, H2D copy of the "
                               "entropy-coded bytes and all kernels, pipelined over distinct batches (pinned_host_to_hbm has the "
                               "details); the rate with the compressed bytes already in HBM is `resident`, and the per-kernel "
                               "times and the roofline are measured there, one batch at a time on one stream")
                              if ptot is not None else
                              "hot path with the compressed bytes already resident in HBM (--no-pipeline: the pinned-host region was skipped)",
            "config": {"workload": f"{args.workload}: batch of {args.batch} per GPU, {desc}",
                       "images_per_gpu": args.batch, "global_batch": int(total_images),
                       "compressed_MB_per_gpu": round(comp_bytes_rank / 1e6, 1),
                       "bits_per_pixel": round(8.0 * comp_bytes_rank / pixels_rank, 3),
                       "parallelism": f"image-sharded x{world}, no data-path collective" + (" (REHEARSAL: all ranks on GPU 0, gloo)" if rehearsal else ""),
                       "entropy_mode": "serial (diagnostic)" if args.serial_entropy else "subsequence-sync",
                       "streams": n_streams},
            "resident": {"value": round(resident_value, 1), "unit": "MP/s", "ms_per_step": round(resident_ms, 4),
                         "measures": "the same batch decoded every step, compressed bytes already in HBM, one stream"},
            "roofline": {"kernel": "mj_k_idct_colour", "bound": "hbm", "achieved": round(achieved, 1),
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                         "traffic": traffic, "traffic_source": traffic_source,
                         "algorithmic_bytes_per_launch": int(alg_bytes), "avg_launch_ms": round(k_avg_ms, 4),
                         "measured_in": "resident region"},
            "entropy": dict({"kernel": ("mj_k_spec_* (+fix/write)" if speculative else "mj_k_entropy_sync (+fix/write)"),
                             "avg_launch_ms": round(e_avg_ms, 4),
                             "decoded_Gbit_per_s": round(8.0 * comp_bytes_rank / n_streams / (e_avg_ms * 1e-3) / 1e9, 2) if e_avg_ms else None,
                             "Gbit_per_s_per_CU": round(8.0 * comp_bytes_rank / n_streams / (e_avg_ms * 1e-3) / 1e9 / N_CUS, 4) if e_avg_ms else None,
                             "synchronisation": "speculative" if speculative else "rounds",
                             "sequences": stats["sequences"], "subsequences": stats["subsequences"]},
                            **({"guessed_hops": stats["rounds_sum"], "subsequences_decoded_again": stats["round_tasks"]} if speculative else
                               {"sync_rounds_avg": round(stats["rounds_sum"] / max(stats["workgroups"], 1), 2),
                                "sync_rounds_max": stats["rounds_max"],
                                "redecoded_fraction": round(stats["round_tasks"] / max(stats["subsequences"], 1), 3)})),
            "end_to_end_hbm": {"algorithmic_bytes_per_px": round(e2e_bytes_px, 3),
                               "frac_of_peak": round(e2e_bytes_px * resident_value * 1e6 / world / (HBM_PEAK_GBS * 1e9), 4)},
            "kernels": kernels,
            "single_image": singles,
            "resident_batch_of_64": small,
            "kernels_max_over_ranks_ms": {k: round(v, 4) for k, v in kernel_ms_max.items()},
            "host": {"generate_s": round(t_gen, 2), "gen_workers": workers, "one_shot_batch_create_s": round(t_prepare, 4),
                     "first_batch_create_s": round(t_prepare_first, 4), "host_cpus": os.cpu_count(),
                     "cpus_allowed": len(allowed0), "placement_rank0": place.as_dict()},
        }
        out.update(rank_arrays(rows))
        if ptot is not None:
            pk = {name: round(ms / max(n, 1), 4) for name, (ms, n) in pipe_res["kernels"].items()}
            # consecutive batches decode on two streams in turn: the per-kernel durations of this
            # region include the time a kernel shares the chip with the neighbouring batch's, so the
            # resident region's kernel times (above) are the yardstick for decode time alone
            decode_ms = sum(k["avg_ms"] for k in kernels.values())
            out["pinned_host_to_hbm"] = {
                "value": round(value, 1), "unit": "MP/s", "ms_per_step": round(ms_step, 4),
                "starts": "JPEG files in page-locked host memory", "ends": "RGB24 in HBM",
                "includes": "host header parse + table flattening, H2D copy of the entropy-coded bytes, all kernels",
                "distinct_batches_per_gpu": n_distinct, "batches_in_flight": pipe_res["depth"],
                "host_work_ms_per_submit_max_over_ranks": round(ptot.kernel_ms_max[0], 4),
                "host_wait_for_slot_ms_per_submit": round(pipe_res["host_wait_ms"], 4),
                "submit_call_ms_incl_binding": round(pipe_res["python_submit_ms"], 4),
                "h2d_copy_alone_ms": round(pipe_res["h2d_alone_ms"], 4),
                "h2d_copy_alone_GBps": round(pipe_res["h2d_GBps"], 2),
                "kernels_avg_ms_while_overlapping_the_next_batch": pk,
                "decode_streams": 2,
                "ms_per_step_over_max_of_decode_and_copy": round(ms_step / max(decode_ms, pipe_res["h2d_alone_ms"], 1e-9), 3),
                "fraction_of_resident_rate": round(value / resident_value, 3)}
            if pipe_res["parity_bad"] is not None:
                out["pinned_host_to_hbm"]["parity"] = {"images_checked": len(digests),
                                                       "images_with_coefficient_mismatch": pipe_res["parity_bad"][0],
                                                       "images_with_rgb_mismatch": pipe_res["parity_bad"][1]}
        if stot is not None:
            s_value = stot.pixels / stot.elapsed_max / 1e6
            out["sustained"] = {"seconds": round(stot.elapsed_max, 3), "submits": int(stot.images), "MPps": round(s_value, 1),
                                "ratio_to_value": round(s_value / value, 4),
                                "measures": "the pinned-host pipeline of `value`, kept running over the distinct batches",
                                **sustained["sensors"]}
        if parity is not None:
            out["parity"] = parity
        out["parity_mismatches_all_ranks"] = int(tot.mismatches)
        if cpu_base is not None:
            out["cpu_baseline"] = cpu_base
            out["gpu_over_cpu"] = round(value / cpu_base["value"], 1)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
