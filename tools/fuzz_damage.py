"""Truncated, bit-flipped, marker-injected, overwritten and shortened entropy segments in one batch with intact files:
nothing faults, and the intact neighbours still decode exactly: python tools/fuzz_damage.py seed rounds  (GPU box)."""
import os, sys, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
import jpeg_decoder_amd as J
from jpeg_decoder_amd import binding
from conftest import load_fixture
import torch
names = ["c420_q85_100x75_ri3", "c444_q85_80x56_rirow", "g_q75_72x40_ri4", "c422_q95_opt_136x40_ri5", "c420_q50_opt_96x80",
         "c420_q95_160x120_rirow", "C1_c420_q85_512x512", "c444_q75_opt_57x43"]
fx = [load_fixture(n) for n in names]
big = J.synth.encode_jpeg(J.synth.synth_image(1600, 1200, 9), "4:2:0", 90, 0)   # multi-sequence interval
ctx = J.Context(0)
rng = np.random.default_rng(int(sys.argv[1]))
def damage(data):
    i = data.index(b"\xff\xda"); h = i + 2 + ((data[i + 2] << 8) | data[i + 3]); n = len(data) - h
    d = bytearray(data); k = int(rng.integers(0, 6))
    if k == 0: d = d[:h + int(rng.integers(0, n))]
    elif k == 1:
        for _ in range(int(rng.integers(1, 6))): d[h + int(rng.integers(0, n - 2))] ^= 1 << int(rng.integers(0, 8))
    elif k == 2:
        for _ in range(3): d[h + int(rng.integers(0, n - 2))] = 0xFF
    elif k == 3:
        p = h + int(rng.integers(0, n - 4)); d[p] = 0xFF; d[p + 1] = 0xD0 + int(rng.integers(0, 8))
    elif k == 4:
        a = h + int(rng.integers(0, n - 40)); d[a:a + 32] = bytes(rng.integers(0, 256, 32, dtype=np.uint8))
    else:
        a = h + int(rng.integers(0, n - 8)); del d[a:a + int(rng.integers(1, 8))]
    return bytes(d)
bad = 0; total = 0
for rnd in range(int(sys.argv[2])):
    files = []; good = []
    for j in range(40):
        src = big if j % 13 == 5 else fx[int(rng.integers(0, len(fx)))][0]
        files.append(damage(src)); good.append(None)
    for (d, g) in fx[:3]:
        files.append(d); good.append(g)
    for flags in (0, binding.MJ_ENTROPY_SERIAL):
        b = ctx.batch(files, flags)
        b.decode(); b.sync()
        for i, g in enumerate(good):
            if g is not None:
                total += 1
                if b.status(i) != 0 or not np.array_equal(b.coefs(i), g["coefs"]) or not np.array_equal(b.rgb(i), g["rgb"]):
                    bad += 1; print("NEIGHBOUR DAMAGED", rnd, i, flush=True)
        b.close()
    if rnd % 10 == 9: print("round", rnd, "intact checked", total, "bad", bad, flush=True)
print("DONE", total, bad)
sys.exit(1 if bad else 0)
