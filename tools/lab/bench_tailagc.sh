#!/bin/bash
# k_tail_agc against k_tail_s + k_agc, untraced: tools/lab/bench_tailagc.sh <out prefix> [sizes...]
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-r5_ta}; shift
SIZES=${@:-"c4 64 256 2048 4096"}
for n in $SIZES; do
  for m in "--tail-agc 1" "--tail-agc 2" "--tail-agc 2 --pipelined" "--tail-agc 1 --pipelined"; do
    tag=$(echo "$n$m" | tr -d ' -')
    if [ "$n" = c4 ]; then sel="--config c4"; else sel="--channels $n"; fi
    python3 $GRAFT_REPO_ROOT/bench.py $sel --steps 300 --warmup 20 --no-cpu-baseline --no-extra-configs $m > ${O}_$tag.json 2> ${O}_$tag.err || { echo "bench $n $m failed"; tail -5 ${O}_$tag.err; continue; }
    python3 - ${O}_$tag.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
k = d["roofline"]["kernels"]
print(sys.argv[1].split("/")[-1], "ms/step %.4f" % d["ms_per_step"], {n: round(v["ms"], 4) for n, v in k.items()})
PY
  done
done
