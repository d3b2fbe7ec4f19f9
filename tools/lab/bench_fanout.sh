#!/bin/bash
# What the per-step fan-out of the IQ block costs, on one GPU: no fan-out / forced 1-rank RCCL broadcast
# (pipelined, in line) / host fan-out (shared page-locked ring, the rank's own H2D) -- at configs[3]'s per-GPU size and at the
# headline's; two ranks on the one GPU through the host fan-out; the in-process multibank.   tools/lab/bench_fanout.sh <out prefix>
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-r5_fan}
B="python3 $GRAFT_REPO_ROOT/bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-extra-configs"
show() { python3 - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    k = (d.get("roofline") or {}).get("kernels") or {}
    print("%-34s ms/step %.4f  value %.3e  n_gpus %d  %s %s" % (sys.argv[2], d["ms_per_step"], d["value"] * 1e6, d["n_gpus"],
          {n: round(v["ms"], 4) for n, v in k.items()}, (d.get("input_check") or {}).get("blocks_wrong", "")))
    h = d.get("host_pointer_entry")
    if h and "ms_per_block" in h:
        print("%-34s    host-pointer entry: %.3f ms per block" % ("", h["ms_per_block"]))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
run() { tag=$1; shift; timeout -k 10 300 "$@" > ${O}_$tag.json 2> ${O}_$tag.err; rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "$tag killed: stopping"; exit 99; fi; show ${O}_$tag.json "$tag"; }
for cfg in "c4:--config c4" "headline:"; do
  n=${cfg%%:*}; a=${cfg#*:}
  run ${n}_nofan $B $a || exit $?
  run ${n}_rccl $B $a --force-dist --fanout rccl || exit $?
  run ${n}_inline $B $a --force-dist --fanout inline || exit $?
  run ${n}_host $B $a --force-dist --fanout host || exit $?
done
run c4_2ranks_1gpu_host $B --config c4 --gpus 2 --fanout host --ranks-on-one-gpu || exit $?
run c4_4ranks_1gpu_host $B --config c4 --gpus 4 --fanout host --ranks-on-one-gpu || exit $?
run multibank_1x10000 $B --multibank 1 --channels 10000 || exit $?
run multibank_2x5000 $B --multibank 2 --channels 5000 || exit $?
run multibank_2x1024 $B --multibank 2 --channels 1024 || exit $?
run multibank_4x1024 $B --multibank 4 --channels 1024 || exit $?
