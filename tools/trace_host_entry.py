#!/usr/bin/env python3
"""Timeline of the drop-in entry (sdrxgpu_bank_process: host pointers in and out) from a rocprofv3 --kernel-trace
--memory-copy-trace run of tools/lab/host_entry_calls.py: for the last call with IQ + audio and the last with audio only, every
kernel and copy with its start (us after the call's H2D copy began) and duration, the achieved D2H rate, the gap between a
part's last kernel and its first copy, and the time the copy engine sat idle between the first and the last D2H.
  tools/trace_host_entry.py <dir with *kernel_trace.csv and *memory_copy_trace.csv>"""
import csv
import glob
import sys


def load(pattern):
    f = sorted(glob.glob(sys.argv[1] + "/**/" + pattern, recursive=True))
    return list(csv.DictReader(open(f[-1]))) if f else []


def col(row, *names):
    for n in names:
        for k in row:
            if k.lower() == n.lower():
                return row[k]
    return None


kern, cop = load("*kernel_trace.csv"), load("*memory_copy_trace.csv")
if not kern or not cop:
    sys.exit("need a kernel trace and a memory-copy trace under " + sys.argv[1])
# (this rocprofv3's copy records carry no size: the sizes follow from the bank's channel parts, which the traced program wrote down)
import json
side = json.load(open(sys.argv[2])) if len(sys.argv) > 2 else None
plans = [side["parts"], side.get("parts_audio_only", side["parts"])] if side else []
ev = []
for r in kern:
    nm = r["Kernel_Name"].split("(")[0].replace("void sdrxgpu::", "").split("<")[0]
    ev.append([int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "kernel", nm, 0])
for r in cop:
    d = (col(r, "Direction") or "").upper()
    # (a copy into memory page-locked with hipHostRegister is reported as device-to-device: the registered range is mapped into
    # the device's address space; the entry itself makes no copy between device buffers when its input is a host pointer)
    kind = "H2D" if "HOST_TO_DEVICE" in d else "D2H" if ("DEVICE_TO_HOST" in d or "DEVICE_TO_DEVICE" in d) else d
    ev.append([int(r["Start_Timestamp"]), int(r["End_Timestamp"]), kind, "copy", 0])
ev.sort()
# a call = an H2D copy with a stage-1/2 kernel behind it, up to the next H2D copy
starts = [i for i, e in enumerate(ev) if e[2] == "H2D"]
calls = [ev[a:b] for a, b in zip(starts, starts[1:] + [len(ev)])]
calls = [c for c in calls if any(e[2] == "kernel" and e[3].startswith("k_stage12") for e in c)]
kinds = {}
for c in calls:
    c[0][4] = 2560000
    d2h = [e for e in c if e[2] == "D2H"]
    kinds[id(c)] = None
    if plans:
        rows_iq = [b - a for a, b in zip(plans[0][:-1], plans[0][1:])]
        rows_au = [b - a for a, b in zip(plans[1][:-1], plans[1][1:])]
        if len(d2h) == 2 * len(rows_iq):       # IQ then audio, part by part
            kinds[id(c)] = "both"
            for k, e in enumerate(d2h):
                e[4] = rows_iq[k // 2] * 512 * (8 if k % 2 == 0 else 4)
        elif len(d2h) == len(rows_au):         # audio only
            kinds[id(c)] = "audio"
            for k, e in enumerate(d2h):
                e[4] = rows_au[k] * 512 * 4


def describe(call, title):
    t0 = call[0][0]
    end = max(e[1] for e in call)
    print(f"\n== {title}: {len(call)} operations, {(end - t0) / 1e3:.1f} us from the start of the H2D copy to the end of the last operation")
    last_kernel_end = None
    d2h = [e for e in call if e[2] == "D2H"]
    prev_copy_end = None
    for s, e, kind, nm, nbytes in call:
        line = f"  {(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:8.1f} us  {kind:6s} {nm:16s}"
        if nbytes:
            line += f" {nbytes / 1e6:8.2f} MB  {nbytes / max(e - s, 1):6.1f} GB/s"
        if kind == "D2H":
            if last_kernel_end is not None:
                line += f"   starts {(s - last_kernel_end) / 1e3:6.1f} us after the last kernel end before it"
            if prev_copy_end is not None:
                line += f", {(s - prev_copy_end) / 1e3:6.1f} us after the previous copy"
            prev_copy_end = e
        print(line)
        if kind == "kernel":
            last_kernel_end = e if last_kernel_end is None else max(last_kernel_end, e)
    if d2h:
        nb = sum(e[4] for e in d2h)
        busy = sum(e[1] - e[0] for e in d2h)
        span = d2h[-1][1] - d2h[0][0]
        kb = sum(e[1] - e[0] for e in call if e[2] == "kernel")
        k_end = max(e[1] for e in call if e[2] == "kernel")
        print(f"  D2H: {nb / 1e6:.2f} MB in {len(d2h)} copies, {nb / busy:.1f} GB/s while copying, {nb / span:.1f} GB/s from the first copy's start to the "
              f"last one's end ({(span - busy) / 1e3:.1f} us of that span no copy ran)")
        print(f"  kernels: {kb / 1e3:.1f} us in total, the last ends {(k_end - t0) / 1e3:.1f} us into the call; first D2H starts {(d2h[0][0] - t0) / 1e3:.1f} us "
              f"into the call; {(end - k_end) / 1e3:.1f} us of copies remain after the last kernel")


both = [c for c in calls if kinds[id(c)] == "both"]
audio = [c for c in calls if kinds[id(c)] == "audio"]
if plans:
    print("channel parts: calls with IQ", plans[0], " audio-only calls", plans[1])
if both:
    describe(both[-1], "IQ + audio (61.4 MB out at 10 000 channels)")
if audio:
    describe(audio[-1], "audio only (20.5 MB out at 10 000 channels)")
if not both and not audio:
    describe(calls[-1], "last call")
