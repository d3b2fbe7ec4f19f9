for n in 2048 2560 2816 3072 3328 3584 4096; do for bh in 12 6 4 3; do timeout -k 10 200 python bench.py --channels $n --steps 60 --warmup 10 --no-cpu-baseline --no-extra-configs --matrix 2 --block-hint $bh > gpurun_out/mf.json 2>gpurun_out/mf.err || { tail -5 gpurun_out/mf.err; exit 1; }; python -c "
import json,sys
d=json.loads(open('gpurun_out/mf.json').read().strip().splitlines()[-1]); k=d['roofline']['kernels']; print(sys.argv[1], sys.argv[2], 'step', round(d['ms_per_step'],4), {n:round(v['ms'],4) for n,v in k.items()})" "$n" "nb=$bh"; done; done
