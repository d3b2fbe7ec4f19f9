#!/bin/bash
cd $GRAFT_REPO_ROOT
B="python3 bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-extra-configs"
show() { python3 - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith('{"metric"')][0])
    k = (d.get("roofline") or {}).get("kernels") or {}
    print("%-40s ms/step %.4f  %s" % (sys.argv[2], d["ms_per_step"], {n: round(v["ms"], 4) for n, v in k.items()}))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
run() { tag=$1; shift; timeout -k 10 300 "$@" > gpurun_out/r5_ba_$tag.json 2> gpurun_out/r5_ba_$tag.err; show gpurun_out/r5_ba_$tag.json "$tag"; }
run mb_2x512 $B --multibank 2 --channels 512
run mb_4x256 $B --multibank 4 --channels 256
run mb_1x1024 $B --multibank 1 --channels 1024
run head_pipelined $B --pipelined
run c3 $B --config c4
run c3_pipelined $B --config c4 --pipelined
export SDRXGPU_LIB_PATH=$GRAFT_REPO_ROOT/tmp_exp/lab5/lib_agcprio.so
run prio_c3 $B --config c4
run prio_c3_pipelined $B --config c4 --pipelined
run prio_head_pipelined $B --pipelined
run prio_head $B
run prio_64 $B --channels 64
run prio_64_pipelined $B --channels 64 --pipelined
