"""One image decoded on its own, `reps` times, for a kernel trace:
    rocprofv3 --kernel-trace --output-format csv -d OUT -o run -- python3 tools/single_trace.py C1 30
    python3 tools/single_trace.py --summarise OUT [kernels per decode]
The summary prints, per kernel in launch order, its average duration and the average gap
to the kernel before it, and the wall time of a decode from first kernel start to last end."""
import csv
import glob
import os
import sys

sys.path.insert(0, os.getcwd())


def summarise(d):
    f = glob.glob(f"{d}/**/*_kernel_trace.csv", recursive=True)[0]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    rows = [r for r in rows if "mj_k" in r["Kernel_Name"]]
    names = [r["Kernel_Name"][r["Kernel_Name"].index("mj_k"):].split("(")[0] for r in rows]
    # a decode starts at every mj_k_scan_count
    starts = [i for i, n in enumerate(names) if n.startswith("mj_k_scan_count")]
    starts.append(len(rows))
    per = {}
    walls = []
    for a, b in zip(starts[len(starts) // 3:-1], starts[len(starts) // 3 + 1:]):      # skip the warm-up third
        seq = rows[a:b]
        walls.append((int(seq[-1]["End_Timestamp"]) - int(seq[0]["Start_Timestamp"])) / 1e3)
        for j, r in enumerate(seq):
            dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
            gap = (int(r["Start_Timestamp"]) - int(seq[j - 1]["End_Timestamp"])) / 1e3 if j else 0.0
            e = per.setdefault((j, names[a + j]), [0.0, 0.0, 0, r["Workgroup_Size"] if "Workgroup_Size" in r else "", r.get("Grid_Size", "")])
            e[0] += dur
            e[1] += gap
            e[2] += 1
    print(f"decodes {len(walls)}  first start -> last end: {sum(walls) / len(walls):.1f} us")
    tk = tg = 0.0
    for (j, n), (dur, gap, cnt, wg, grid) in sorted(per.items()):
        print(f"  {j:2d} {n:34s} dur {dur / cnt:7.1f} us  gap before {gap / cnt:6.1f} us  grid {grid} wg {wg}")
        tk += dur / cnt
        tg += gap / cnt
    print(f"  kernels {tk:.1f} us + gaps {tg:.1f} us")


if __name__ == "__main__":
    if sys.argv[1] == "--summarise":
        summarise(sys.argv[2])
        sys.exit(0)
    import torch
    torch.cuda.init()
    import jpeg_decoder_amd as J
    if os.environ.get("MJ_LIB"):                  # another build of the library
        J.binding.LIB_PATH = os.path.abspath(os.environ["MJ_LIB"])
    wl, reps = sys.argv[1], int(sys.argv[2])
    n = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    if os.environ.get("MJ_SYNC_POLICY"):          # "mode,sub_bits,lane_subs,extra" (mj_set_sync_policy)
        J.binding.set_sync_policy(*[int(x) for x in os.environ["MJ_SYNC_POLICY"].split(",")])
    ctx = J.Context(0)
    files = [J.synth.config_jpeg(wl, 1 + i) for i in range(n)]
    for _ in range(reps):
        b = ctx.batch(files, 0)
        b.decode()
        b.sync()
        b.close()
    ctx.close()
