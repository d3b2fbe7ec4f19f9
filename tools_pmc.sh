#!/bin/bash
# Collect PMC counters for bench.py in separate rocprofv3 passes (never combined with tracing).
# Usage (on the GPU box): ./tools_pmc.sh <outdir-under-gpurun_out> [bench args...]
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export GPU_MAX_HW_QUEUES=8      # (what bench.py sets for itself; under rocprofv3 the runtime starts before the program does)
i=0
for SET in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "FETCH_SIZE GRBM_GUI_ACTIVE" \
           "WRITE_SIZE" \
           "TCC_HIT_sum TCC_MISS_sum" ; do
  i=$((i+1))
  timeout -k 10 240 rocprofv3 --pmc $SET --output-format csv -d $OUT/pass$i -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-extra-configs "$@" > $OUT/pass$i.log 2>&1
  rc=$?
  echo "pass $i ($SET): exit $rc"
  # a pass killed at its time limit means a hang: no further GPU process after it
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "pass $i was killed at its time limit: stopping"; exit 99; fi
done
python3 - "$@" <<PY
import csv, glob, collections, json, re, sys
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/pass*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0][:70]
        agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k in ("VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size"):
            if k in r: agg[name][k].append(float(r[k]))
with open("$OUT/summary.txt", "w") as o:
    for k, d in agg.items():
        if "sdrxgpu" not in k: continue
        o.write(k + "\n")
        for c, v in sorted(d.items()):
            o.write(f"   {c:28s} n={len(v):3d} mean={sum(v)/len(v):.6g}\n")
print(open("$OUT/summary.txt").read())
# HBM traffic per launch, corrected as MI355X_MICROARCH.md (HBM) prescribes for gfx950: FETCH_SIZE / WRITE_SIZE are in
# KiB; FETCH_SIZE reports 1/2 of the bytes of wide (16 B/lane) streaming reads.  Sustained clock = GRBM_GUI_ACTIVE
# (summed over the 8 XCDs) / 8 / the kernel's mean duration in the SAME profiled pass (its bench line).
def mean(d, c): return sum(d[c]) / len(d[c]) if c in d else None
def bench_line(p):
    for l in open(p):
        if l.startswith('{"metric"'): return json.loads(l)
    return None
args = " ".join(sys.argv[1:])
b3 = bench_line("$OUT/pass3.log")
m = re.search(r"--channels (\d+)", args)
out = {"channels": (b3["config"]["channels_per_gpu"] if b3 else int(m.group(1)) if m else 10000), "bench_args": args,
       "correction": "bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950: FETCH_SIZE counts 128-B requests as 64 B)"}
for k, d in agg.items():
    short = "k_stage12" if "k_stage12" in k else "k_tail" if "k_tail" in k else "k_agc" if "k_agc" in k else None
    if not short or "FETCH_SIZE" not in d or "WRITE_SIZE" not in d: continue
    f, w = mean(d, "FETCH_SIZE"), mean(d, "WRITE_SIZE")
    out[short] = {"kernel": k, "FETCH_SIZE_KiB_raw": f, "WRITE_SIZE_KiB_raw": w, "hbm_bytes_per_launch": (2.0 * f + w) * 1024.0,
                  "SQ_INSTS_VALU": mean(d, "SQ_INSTS_VALU"), "SQ_WAVE_CYCLES": mean(d, "SQ_WAVE_CYCLES"),
                  "SQ_ACTIVE_INST_VALU": mean(d, "SQ_ACTIVE_INST_VALU"), "GRBM_GUI_ACTIVE": mean(d, "GRBM_GUI_ACTIVE"),
                  "VGPR_Count": mean(d, "VGPR_Count"), "LDS_Block_Size": mean(d, "LDS_Block_Size")}
    if b3 and short in b3["roofline"]["kernels"]:
        ms = b3["roofline"]["kernels"][short]["ms"]
        out[short]["ms_in_the_GRBM_pass"] = ms
        out[short]["clock_ghz"] = mean(d, "GRBM_GUI_ACTIVE") / 8.0 / (ms * 1e-3) / 1e9
if "k_stage12" in out:
    out["k_stage12_hbm_bytes_per_launch"] = out["k_stage12"]["hbm_bytes_per_launch"]
    out["sustained_clock_ghz"] = out["k_stage12"].get("clock_ghz")
if b3:
    out.update(b3["config"]["kernel_flags"])         # fused_tail, pair_lanes, matrix_pipe, fused3, tail_agc: what bank_info said
    out["kernel_path"] = b3["config"]["kernel_path"]
json.dump(out, open("$OUT/traffic.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
