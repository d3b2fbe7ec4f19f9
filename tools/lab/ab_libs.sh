#!/bin/bash
# A/B of lab builds of the library (SDRXGPU_LIB_PATH): tools/lab/ab_libs.sh <out prefix> "<bench args>" lib1.so lib2.so ...
# prints ms/step and the kernel times of each; the in-tree build first
O=$GRAFT_REPO_ROOT/gpurun_out/$1; ARGS=$2; shift 2
for lib in default "$@"; do
  tag=$(basename $lib .so)
  if [ $lib = default ]; then unset SDRXGPU_LIB_PATH; else export SDRXGPU_LIB_PATH=$GRAFT_REPO_ROOT/$lib; fi
  python3 $GRAFT_REPO_ROOT/bench.py $ARGS --steps 300 --warmup 20 --no-cpu-baseline --no-extra-configs > ${O}_$tag.json 2> ${O}_$tag.err || { echo "bench $tag failed"; tail -5 ${O}_$tag.err; continue; }
  python3 - ${O}_$tag.json $tag <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
k = d["roofline"]["kernels"]
print("%-12s ms/step %.4f" % (sys.argv[2], d["ms_per_step"]), {n: round(v["ms"], 4) for n, v in k.items()})
PY
done
