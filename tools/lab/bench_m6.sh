#!/bin/bash
cd $GRAFT_REPO_ROOT
for n in ${@:-10000 4096 1024 256}; do for m in 1 2; do python3 bench.py --rate 6 --channels $n --matrix $m --steps 200 --no-cpu-baseline --no-extra-configs 2>gpurun_out/r5_rate.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d[\"roofline\"][\"kernels\"]
print(\"6 MS/s $n ch matrix=$m\", \"ms/step %.4f\"%d[\"ms_per_step\"], {n:round(v[\"ms\"],4) for n,v in k.items()})" || tail -3 gpurun_out/r5_rate.err; done; done
