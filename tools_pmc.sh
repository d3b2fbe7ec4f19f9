#!/bin/bash
# Collect PMC counters for bench.py in separate rocprofv3 passes (never combined with tracing).
# Usage (on the GPU box): ./tools_pmc.sh <outdir-under-gpurun_out> [bench args...]
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for SET in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "FETCH_SIZE GRBM_GUI_ACTIVE" \
           "WRITE_SIZE" \
           "TCC_HIT_sum TCC_MISS_sum" ; do
  i=$((i+1))
  timeout -k 10 240 rocprofv3 --pmc $SET --output-format csv -d $OUT/pass$i -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-extra-configs "$@" > $OUT/pass$i.log 2>&1
  echo "pass $i ($SET): exit $?"
done
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/pass*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0][:40]
        agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open("$OUT/summary.txt", "w") as o:
    for k, d in agg.items():
        if "sdrxgpu" not in k: continue
        o.write(k + "\n")
        for c, v in sorted(d.items()):
            o.write(f"   {c:28s} n={len(v):3d} mean={sum(v)/len(v):.6g}\n")
print(open("$OUT/summary.txt").read())
# HBM traffic per launch of the dominant kernel, corrected as MI355X_MICROARCH.md (HBM) prescribes for gfx950:
# FETCH_SIZE / WRITE_SIZE are in KiB; FETCH_SIZE reports 1/2 of the bytes of wide (16 B/lane) streaming reads.
import json, re, sys
for k, d in agg.items():
    if "k_stage12" in k and "FETCH_SIZE" in d and "WRITE_SIZE" in d:
        f = sum(d["FETCH_SIZE"]) / len(d["FETCH_SIZE"]); w = sum(d["WRITE_SIZE"]) / len(d["WRITE_SIZE"])
        args = " ".join(sys.argv[1:])
        m = re.search(r"--channels (\d+)", "$*")
        json.dump({"kernel": "k_stage12", "channels": int(m.group(1)) if m else 10000,
                   "FETCH_SIZE_KiB_raw": f, "WRITE_SIZE_KiB_raw": w,
                   "k_stage12_hbm_bytes_per_launch": (2.0 * f + w) * 1024.0,
                   "correction": "bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950: FETCH_SIZE counts 128-B requests as 64 B)"},
                  open("$OUT/traffic.json", "w"), indent=1)
PY
