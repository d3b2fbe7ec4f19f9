#!/bin/bash
# serial against pipelined order, small banks, untraced: tools/lab/bench_pipelined.sh <out prefix>
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-r5_pipe}
for n in 1024 64 2048 4096; do
  for m in "" "--pipelined"; do
    python3 $GRAFT_REPO_ROOT/bench.py --channels $n --steps 300 --warmup 20 --no-cpu-baseline --no-extra-configs $m > ${O}_$n$m.json 2> ${O}_$n$m.err || { echo "bench $n $m failed"; tail -5 ${O}_$n$m.err; }
    python3 - ${O}_$n$m.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
k = d["roofline"]["kernels"]
print(sys.argv[1].split("/")[-1], "ms/step %.4f" % d["ms_per_step"], {n: round(v["ms"], 4) for n, v in k.items()}, d["config"]["call_semantics"][:9])
PY
  done
done
