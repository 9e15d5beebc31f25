"""Phase times of the IDCT + colour kernel (library built with -DMJ_IDCT_TRACE)."""
import os, sys
sys.path.insert(0, os.getcwd())
import bench, numpy as np
import jpeg_decoder_amd as J
from jpeg_decoder_amd import binding
from jpeg_decoder_amd.shard import shard_seeds
import torch
wl = sys.argv[1] if len(sys.argv) > 1 else "C3"
files = bench.generate(wl, shard_seeds(64, 0, 1), 8)
ctx = J.Context(0)
dbg = int(sys.argv[2], 0) if len(sys.argv) > 2 else 0
bt = ctx.batch(files, binding.MJ_TIMING | (dbg << 16))
for i in range(4):
    bt.decode(); torch.cuda.synchronize()
st = bt.entropy_stats()
print(st)
print(bt.kernel_ms())
