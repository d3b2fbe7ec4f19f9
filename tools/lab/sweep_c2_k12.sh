#!/bin/bash
# Lab: the stage-1/2 kernel of the 64-channel bank (BASELINE configs[1]) against bodies per lane.  -> gpurun_out/r05_sweep_c2_k12.txt
cd $GRAFT_REPO_ROOT
out=gpurun_out/r05_sweep_c2_k12.txt
: > $out
for n in 64 256; do
for h in 0 1 2 3 4 6 12; do
  timeout -k 10 120 python3 bench.py --channels $n --steps 400 --warmup 20 --no-cpu-baseline --no-extra-configs --block-hint $h 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['roofline']['kernels']
print('$n channels, block hint $h: ms/step %.4f'%d['ms_per_step'], {n:round(v['ms'],4) for n,v in k.items()})" >> $out
done
done
cat $out
