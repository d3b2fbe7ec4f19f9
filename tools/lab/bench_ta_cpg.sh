#!/bin/bash
# Lab: k_tail_agc's channels per workgroup (tuning.tail_agc_cpg) at several bank sizes.  -> gpurun_out/r05_ta_cpg.txt
cd $GRAFT_REPO_ROOT
out=gpurun_out/r05_ta_cpg.txt
: > $out
for n in ${@:-64 128 256 384 512 768 1024 1280}; do
  for c in 1 2 4; do
    timeout -k 10 120 python3 bench.py --channels $n --steps 300 --warmup 20 --no-cpu-baseline --no-extra-configs --tail-agc 2 --tail-agc-cpg $c 2>/dev/null | python3 -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); k=d['roofline']['kernels']
    print('$n channels, $c per workgroup: ms/step %.4f'%d['ms_per_step'], {n:round(v['ms'],4) for n,v in k.items()})
except Exception as e: print('$n channels, $c per workgroup: refused')" >> $out
  done
  timeout -k 10 120 python3 bench.py --channels $n --steps 300 --warmup 20 --no-cpu-baseline --no-extra-configs --tail-agc 1 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['roofline']['kernels']
print('$n channels, k_tail_s + k_agc: ms/step %.4f'%d['ms_per_step'], {n:round(v['ms'],4) for n,v in k.items()})" >> $out
done
cat $out
