"""A/B timing of library builds on one GPU box: python tools/ab.py libA.so libB.so [libC.so ...] [workload] [reps]
Each build decodes the same batch (MJ_AB_BATCH images, 64 if unset) in its own process, alternating, and the per-stage times are printed."""
import os, subprocess, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, json
os.environ.setdefault("MJ_BENCH_CACHE", "/tmp/mj_bench_cache")      # the builds decode the same generated files
sys.path.insert(0, %r)
from jpeg_decoder_amd import binding
binding.LIB_PATH = sys.argv[1]
import bench, torch
import jpeg_decoder_amd as J
from jpeg_decoder_amd.shard import shard_seeds
files = bench.generate(sys.argv[2], shard_seeds(int(os.environ.get("MJ_AB_BATCH", "4" if sys.argv[2] == "C4" else "64")), 0, 1), 16)
ctx = J.Context(0)
bt = ctx.batch(files, binding.MJ_TIMING)
for i in range(3):
    bt.decode()
torch.cuda.synchronize()
bt.reset_timing() if hasattr(bt, "reset_timing") else None
for i in range(20):
    bt.decode()
torch.cuda.synchronize()
print(json.dumps({k: round(v[0] / max(v[1], 1), 4) for k, v in bt.kernel_ms().items()}))
''' % ROOT
libs = [a for a in sys.argv[1:] if a.endswith(".so")]        # two builds or more
rest = [a for a in sys.argv[1:] if not a.endswith(".so")]
wl = rest[0] if rest else "C3"
reps = int(rest[1]) if len(rest) > 1 else 3
for r in range(reps):
    for lib in libs:
        out = subprocess.run([sys.executable, "-c", CHILD, os.path.abspath(lib), wl], capture_output=True, text=True, timeout=300)
        line = out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-300:]
        print(os.path.basename(lib), line, flush=True)
