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
# frames coded in one scan per component: any of the three scans is damaged
per_comp = [J.synth.encode_jpeg(J.synth.synth_image(w, h, 11 + w), s_, q, ri, optimise=o, scans_per_component=True)
            for (w, h, s_, q, ri, o) in ((200, 120, "4:2:0", 85, 0, False), (97, 61, "4:2:2", 75, 5, True), (640, 480, "4:4:4", 90, 80, False))]
if os.environ.get("MJ_SYNC_POLICY"):          # "mode,sub_bits,lane_subs,extra" (mj_set_sync_policy)
    binding.set_sync_policy(*[int(x) for x in os.environ["MJ_SYNC_POLICY"].split(",")])
ctx = J.Context(0)
rng = np.random.default_rng(int(sys.argv[1]))
def scan_starts(data):
    """offsets of the SOS markers (a marker walk over the segments, entropy-coded bytes skipped by their stuffing rule)"""
    out = []; i = 2
    while i + 4 <= len(data):
        if data[i] != 0xFF: i += 1; continue
        m = data[i + 1]
        if m in (0x00, 0xFF) or 0xD0 <= m <= 0xD8: i += 2 if m != 0xFF else 1; continue
        if m == 0xD9: break
        if m == 0xDA: out.append(i)
        i += 2 + ((data[i + 2] << 8) | data[i + 3])
    return out
def damage(data):
    sos = scan_starts(data); which = int(rng.integers(0, len(sos)))
    i = sos[which]; h = i + 2 + ((data[i + 2] << 8) | data[i + 3])
    n = (sos[which + 1] if which + 1 < len(sos) else len(data)) - h
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
bad = 0; total = 0; refused = 0
for rnd in range(int(sys.argv[2])):
    files = []; good = []
    for j in range(40):
        src = big if j % 13 == 5 else per_comp[j % 3] if j % 5 == 2 else fx[int(rng.integers(0, len(fx)))][0]
        d = damage(src)
        if src in per_comp:
            # damage that reads as a marker ends a scan early: such a file is refused when the batch is built
            try:
                ctx.batch([d], 0).close()
            except Exception:
                refused += 1
                continue
        files.append(d); good.append(None)
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
    if rnd % 10 == 9: print("round", rnd, "intact checked", total, "bad", bad, "refused", refused, flush=True)
print("DONE", total, bad, "refused at creation", refused)
sys.exit(1 if bad else 0)
