#!/bin/bash
# Lab: k_stage12m of the 1 024-channel bank (--config c4 at N = 1) against the bodies per lane and the workgroup shape.
# One gpurun call; writes gpurun_out/r05_sweep_c4_k12m.txt
out=gpurun_out/r05_sweep_c4_k12m.txt
: > $out
run() {
    echo "== $*" >> $out
    timeout -k 10 120 python bench.py --config c4 --steps 300 --no-extra-configs --no-cpu-baseline "$@" 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    l = l.strip()
    if l.startswith('{'):
        d = json.loads(l)
        k = d['roofline']['kernels']
        print('ms_per_step %.4f' % d['ms_per_step'], {n: round(v['ms'], 4) for n, v in k.items()}, d['config'].get('kernel_path'))
    elif 'rror' in l:
        print(l[:200])
" >> $out
}
run
for h in 1 2 3 4 6 12; do run --block-hint $h; done
for g in "1 2" "2 1" "1 4" "4 1" "1 1" "2 4" "4 2"; do set -- $g; run --mgs $1 --mgt $2; done
cat $out
