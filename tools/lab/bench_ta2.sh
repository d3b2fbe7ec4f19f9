#!/bin/bash
cd $GRAFT_REPO_ROOT
for n in ${@:-c4 64 256 512 2048}; do if [ $n = c4 ]; then sel="--config c4"; else sel="--channels $n"; fi; for m in "--tail-agc 1" "--tail-agc 2"; do python3 bench.py $sel --steps 300 --warmup 20 --no-cpu-baseline --no-extra-configs $m > gpurun_out/r5_te.json 2>gpurun_out/r5_te.err; python3 -c "
import json,sys
d=json.loads([l for l in open(\"gpurun_out/r5_te.json\") if l.startswith(\"{\")][0]); k=d[\"roofline\"][\"kernels\"]
print(\"$n $m\", \"ms/step %.4f\"%d[\"ms_per_step\"], {n:round(v[\"ms\"],4) for n,v in k.items()})"; done; done
