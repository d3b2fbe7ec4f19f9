#!/bin/bash
# Lab: how much of k_stage12m's scratch stays ordinary stores (the last SDRX_LAB_Y2_KEEP_MB of a launch; the rest streams, `nt`).
# Needs a library built with -DSDRX_LAB_ENV at tmp_exp/lab6/lib_keep.so.  -> gpurun_out/r05_y2_keep.txt
cd $GRAFT_REPO_ROOT
export SDRXGPU_LIB_PATH=$GRAFT_REPO_ROOT/tmp_exp/lab6/lib_keep.so
out=gpurun_out/r05_y2_keep.txt
: > $out
for n in ${@:-10000 6000 4096}; do
  for keep in 100000 0 64 128 192 256 320 100000 192; do
    SDRX_LAB_Y2_KEEP_MB=$keep timeout -k 10 120 python3 bench.py --channels $n --steps 300 --warmup 20 --no-cpu-baseline --no-extra-configs 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['roofline']['kernels']
print('$n channels, keep $keep MB: ms/step %.4f'%d['ms_per_step'], {n:round(v['ms'],4) for n,v in k.items()})" >> $out
  done
done
cat $out
