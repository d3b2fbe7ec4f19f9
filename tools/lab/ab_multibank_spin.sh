#!/bin/bash
# Lab: sdrxgpu_multibank's job slot with and without the short spin in front of its waits (SDRXGPU_LIB_PATH = a build without it).
# On the GPU box: tools/lab/ab_multibank_spin.sh <lib without the spin>
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
  for lib in "" "$1"; do
    for cfg in "--multibank 8 --channels 1024" "--multibank 2 --channels 1024" "--multibank 2 --channels 5000"; do
      SDRXGPU_LIB_PATH=$lib python bench.py $cfg --steps 200 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-12s %-34s %.4f ms/step (pinned entry)   host entry %.3f ms' % ('$lib' and 'no spin' or 'spin', '$cfg', d['ms_per_step'], d['host_pointer_entry']['ms_per_block']))"
    done
  done
done
