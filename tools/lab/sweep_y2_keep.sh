#!/bin/bash
# Lab: how much of k_stage12m's scratch stays ordinary stores (the sets of a launch written last; everything before them streams, `nt`):
# bench.py --y2-keep-mb = tuning.y2_keep_mb (MB + 1; 1 = everything streams, 4294967295 = nothing does, 0 = the library's 192 MB).
# (profiles/r05_y2_keep.txt came from a lab build that read the number from the environment, before the tuning field existed.)
cd $GRAFT_REPO_ROOT
out=gpurun_out/r05_y2_keep.txt
: > $out
for n in ${@:-10000 6000 4096}; do
  for keep in none 0 64 128 192 256 320 none 192; do
    if [ $keep = none ]; then v=4294967295; else v=$((keep + 1)); fi
    timeout -k 10 120 python3 bench.py --channels $n --steps 300 --warmup 20 --no-cpu-baseline --no-extra-configs --y2-keep-mb $v 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['roofline']['kernels']
print('$n channels, ordinary stores for the last $keep MB: ms/step %.4f'%d['ms_per_step'], {n:round(v['ms'],4) for n,v in k.items()})" >> $out
  done
done
cat $out
