"""Random images, shapes, samplings, qualities, restart intervals and table kinds in mixed batches against the CPU
restatement, both entropy modes: python tools/fuzz_gpu.py [seed] [rounds]  (GPU box; needs oracle/ built)."""
import os, sys, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "oracle"))
import numpy as np
import jpeg_decoder_amd as J
from jpeg_decoder_amd import binding
from refbind import Oracle
import torch
orc = Oracle()
if os.environ.get("MJ_SYNC_POLICY"):          # "mode,sub_bits,lane_subs,extra" (mj_set_sync_policy)
    binding.set_sync_policy(*[int(x) for x in os.environ["MJ_SYNC_POLICY"].split(",")])
ctx = J.Context(0)
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
samplings = ["4:4:4", "4:2:2", "4:2:0", "4:4:0", "4:1:1"]
bad = 0; total = 0
t0 = time.time()
for rnd in range(int(sys.argv[2]) if len(sys.argv) > 2 else 12):
    files = []; meta = []
    for i in range(24):
        grey = rng.random() < 0.15
        w = int(rng.integers(1, 700)); h = int(rng.integers(1, 500))
        if rng.random() < 0.08: w = int(rng.integers(700, 3000)); h = int(rng.integers(500, 2000))
        if grey: w = max(8, w // 8 * 8)
        samp = samplings[int(rng.integers(0, len(samplings)))]
        q = int(rng.integers(15, 101))
        ri = int(rng.choice([0, 0, 1, 2, 3, 7, 16, 40, 200]))
        sigma = float(rng.choice([0.0, 2.0, 12.0, 40.0]))
        img = J.synth.synth_image(w, h, seed=int(rng.integers(0, 1 << 30)), sigma=sigma, grey=grey)
        opt = bool(rng.random() < 0.3)
        spc = bool(rng.random() < 0.2) and not grey      # one scan per component
        grp = None if (grey or spc or rng.random() >= 0.15) else str(rng.choice(["y|cbcr", "ycb|cr"]))     # two scans
        data = J.synth.encode_jpeg(img, samp, q, ri, optimise=opt, scans_per_component=spc, scans=grp)
        files.append(data); meta.append((w, h, samp, q, ri, grey, sigma, opt, spc, grp))
    for flags in (0, binding.MJ_ENTROPY_SERIAL):
        b = ctx.batch(files, binding.MJ_WANT_PLANES | flags)
        b.decode(); b.sync()
        for i, data in enumerate(files):
            o = orc.decode(data)
            total += 1
            ok = b.status(i) == 0 and np.array_equal(b.coefs(i), o.coefs) and np.array_equal(b.rgb(i), o.rgb)
            ok = ok and all(np.array_equal(b.plane(i, c), p) for c, p in enumerate(o.planes))
            if not ok:
                bad += 1
                print("MISMATCH", flags, meta[i], "status", b.status(i), flush=True)
        b.close()
    print("round", rnd, "checked", total, "bad", bad, "%.0fs" % (time.time() - t0), flush=True)
print("FUZZ DONE checked", total, "bad", bad)
sys.exit(1 if bad else 0)
