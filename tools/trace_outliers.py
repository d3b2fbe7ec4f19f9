#!/usr/bin/env python3
"""Which launches of the dominant kernel are the slow ones (MaxNs 810 us against AverageNs 577 us)?  Reads a
rocprofv3 --kernel-trace CSV (bench.py --steps 400), prints for the stage-1/2 kernel: the distribution of its durations, the
slowest launches with their position in the run and what ran before them, and the mean by phase of the run (clock settling /
warm-up / timed steps / the five fully instrumented steps at the end)."""
import csv
import glob
import statistics
import sys

files = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))
if not files:
    sys.exit("no kernel_trace.csv under " + sys.argv[1])
rows = list(csv.DictReader(open(files[-1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
name = lambda r: r["Kernel_Name"].split("(")[0].replace("void sdrxgpu::", "")
k12 = [(i, r) for i, r in enumerate(rows) if "k_stage12" in r["Kernel_Name"]]
dur = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for _, r in k12]
print(f"{name(k12[0][1])}: {len(dur)} launches, mean {statistics.mean(dur) / 1e3:.1f} us, median {statistics.median(dur) / 1e3:.1f}, "
      f"min {min(dur) / 1e3:.1f}, max {max(dur) / 1e3:.1f}, stdev {statistics.pstdev(dur) / 1e3:.1f}")
n = len(dur)
for lo, hi, what in ((0, 10, "first 10 launches (cold: clock ramping up)"), (10, 130, "launches 10..129 (settling steps)"),
                     (130, n - 5, "the rest but the last 5 (warm-up + timed steps)"), (n - 5, n, "last 5 (every kernel bracketed by events)")):
    seg = dur[lo:hi]
    if seg:
        print(f"  {what}: mean {statistics.mean(seg) / 1e3:.1f} us, max {max(seg) / 1e3:.1f}")
print("slowest ten launches: (index among the kernel's launches, us, gap to the previous kernel's end in us, previous kernel)")
order = sorted(range(n), key=lambda j: -dur[j])[:10]
for j in sorted(order):
    i, r = k12[j]
    prev = rows[i - 1] if i else None
    gap = (int(r["Start_Timestamp"]) - int(prev["End_Timestamp"])) / 1e3 if prev else float("nan")
    print(f"  #{j}: {dur[j] / 1e3:.1f} us, gap {gap:.1f}, after {name(prev) if prev else '-'}")
hist = {}
for d in dur:
    hist[d // 20000 * 20] = hist.get(d // 20000 * 20, 0) + 1
print("histogram (20 us bins):", ", ".join(f"{k}-{k + 20}: {v}" for k, v in sorted(hist.items())))
