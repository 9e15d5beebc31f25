"""Where the synchronisation kernel's time goes (library built with -DMJ_SYNC_TRACE): python tools/trace_sync.py lib.so [workload]"""
import os, sys
os.environ.setdefault("MJ_BENCH_CACHE", "/tmp/mj_bench_cache")
sys.path.insert(0, os.getcwd())
from jpeg_decoder_amd import binding
binding.LIB_PATH = os.path.abspath(sys.argv[1])
import bench
import jpeg_decoder_amd as J
from jpeg_decoder_amd.shard import shard_seeds
import torch
wl = sys.argv[2] if len(sys.argv) > 2 else "C3"
files = bench.generate(wl, shard_seeds(64, 0, 1), 8)
ctx = J.Context(0)
bt = ctx.batch(files, binding.MJ_TIMING)
for i in range(3):
    bt.decode(); torch.cuda.synchronize()
st = list(bt.entropy_stats().values())
wgs = 1088 if wl == "C3" else 1
print(f"cold pass: sum over workgroups {st[1] / 100 / 1000:9.2f} ms  (/1088 = {st[1] / 100 / 1088:7.1f} us)")
print(f"longest-lived workgroup: {st[2] / 100:9.1f} us")
for name, i in (("rounds with 9..64", 4), ("rounds with <= 8", 6)):
    t, c = st[i], st[i + 1]
    print(f"{name:28s} {c:8d} x {t / max(c, 1) / 100:8.2f} us   (sum over workgroups {t / 100 / 1000:9.2f} ms)")
print({k: round(v[0] / max(v[1], 1), 4) for k, v in bt.kernel_ms().items()})
