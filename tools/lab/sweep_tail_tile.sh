#!/bin/bash
# Lab: k_tail_s's tile (final outputs per workgroup) at the headline size.  -> gpurun_out/r05_sweep_tail_tile.txt
cd $GRAFT_REPO_ROOT
out=gpurun_out/r05_sweep_tail_tile.txt
: > $out
for t in 0 32 64 96 128 192 256 0; do
  timeout -k 10 120 python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extra-configs --tail-tile $t "$@" 2>&1 | python3 -c "
import json,sys
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); k=d['roofline']['kernels']
        print('tail tile $t: ms/step %.4f'%d['ms_per_step'], {n:round(v['ms'],4) for n,v in k.items()})
    elif 'rror' in l: print('tail tile $t:', l[:160])
" >> $out
done
cat $out
