"""Copy engine and kernels side by side in the pinned-host pipeline, from a rocprofv3 trace:

    rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d OUT -o run -- \
        python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --gen-workers 1 --no-singles --parity none --sustain-seconds 0
    python3 tools/overlap.py OUT [steps [compressed MB per submit]]

The window analysed is the benchmark's timed pipelined region: from the start of the first of
its last `steps` large host-to-device copies (one per submit: the batch's entropy-coded bytes)
to the end of the last kernel.  Prints the time the copy engine is busy, the time any decode
kernel is running, and how much of the copy time has a kernel running beside it."""
import csv
import glob
import sys


def union(iv):
    iv = sorted(iv)
    out = []
    for a, b in iv:
        if out and a <= out[-1][1]:
            out[-1][1] = max(out[-1][1], b)
        else:
            out.append([a, b])
    return out


def total(iv):
    return sum(b - a for a, b in iv)


def intersect(x, y):
    i = j = 0
    out = []
    while i < len(x) and j < len(y):
        a, b = max(x[i][0], y[j][0]), min(x[i][1], y[j][1])
        if a < b:
            out.append([a, b])
        if x[i][1] < y[j][1]:
            i += 1
        else:
            j += 1
    return out


def main():
    d = sys.argv[1]
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    kf = glob.glob(f"{d}/**/*_kernel_trace.csv", recursive=True)
    cf = glob.glob(f"{d}/**/*_memory_copy_trace.csv", recursive=True)
    if not kf or not cf:
        sys.exit(f"no kernel or memory-copy trace under {d}: {kf} {cf}")
    copies = list(csv.DictReader(open(cf[0])))
    kernels = [r for r in csv.DictReader(open(kf[0])) if "mj_k" in r["Kernel_Name"]]
    if not copies:
        sys.exit("empty copy trace")
    # the trace gives no sizes: a submit's copy of the entropy-coded bytes is told by its duration
    dur = lambda r: int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    h2d = [r for r in copies if "HOST_TO_DEVICE" in r.get("Direction", "").upper()] or copies
    longest = max(dur(r) for r in h2d)
    big = sorted((r for r in h2d if dur(r) > longest // 2), key=lambda r: int(r["Start_Timestamp"]))
    # in time order: 2 batch creations of the resident region, warm-up submits, the `steps` timed submits, (the untimed
    # parity submit when parity is on,) and the 6 copies of the "copy alone" measurement, which come last
    tail = 6
    win = big[-(steps + tail):-tail] if len(big) >= steps + tail else big[-steps:]
    t0 = int(win[0]["Start_Timestamp"])
    ks = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in kernels]
    t1 = max(e for s, e in ks if s <= int(win[-1]["End_Timestamp"]) + 50_000_000)
    cu = union([(int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in win])
    ku = union([(max(s, t0), min(e, t1)) for s, e in ks if e > t0 and s < t1])
    both = intersect(cu, ku)
    span = t1 - t0
    mb = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0          # compressed MB per submit (the benchmark's compressed_MB_per_gpu)
    print(f"window: {len(win)} submits, {span / 1e6:.2f} ms ({span / 1e6 / len(win):.3f} ms per submit)")
    rate = f" ({mb * len(win) / 1e3 / (total(cu) / 1e9):.1f} GB/s while busy)" if mb else ""
    print(f"copy engine busy   {total(cu) / 1e6:8.2f} ms = {total(cu) / span:.3f} of the window{rate}")
    print(f"kernels running    {total(ku) / 1e6:8.2f} ms = {total(ku) / span:.3f} of the window")
    print(f"both at once       {total(both) / 1e6:8.2f} ms = {total(both) / max(total(cu), 1):.3f} of the copy time, "
          f"{total(both) / max(total(ku), 1):.3f} of the kernel time")
    print(f"neither            {(span - total(cu) - total(ku) + total(both)) / 1e6:8.2f} ms")


if __name__ == "__main__":
    main()
