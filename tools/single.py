"""Single-image decode time with the current tuning environment: python tools/single.py C1|C4|C2|C3"""
import os, sys
sys.path.insert(0, os.getcwd())
import torch
torch.cuda.init()
import bench
import jpeg_decoder_amd as J
ctx = J.Context(0)
for wl in sys.argv[1:]:
    data = J.synth.config_jpeg(wl, 1)
    print(wl, bench.single_image_ms(ctx, data, 0, reps=20), flush=True)
