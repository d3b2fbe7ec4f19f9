#!/bin/bash
# kernel traces of (a) one 1 024-channel bank, serial order, (b) the same bank with SDRXGPU_FLAG_PIPELINED,
# (c) two independent 1 024-channel banks (tools/two_streams.py).  Run on the GPU box: tools/lab/trace_pipelined.sh <outdir>
export GPU_MAX_HW_QUEUES=${GPU_MAX_HW_QUEUES:-8}
E=$GRAFT_REPO_ROOT/gpurun_out/${1:-r5_trace}
mkdir -p $E
cd /tmp && export TMPDIR=/tmp
run() {   # name, program args...
  name=$1; shift
  timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $E/$name -- python3 "$@" > $E/$name.log 2>&1
  rc=$?
  echo "trace $name rc=$rc"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "killed at its time limit: stopping"; exit 99; fi
  python3 $GRAFT_REPO_ROOT/tools/trace_overlap.py $E/$name 40 > $E/${name}_overlap.txt 2>&1
  find $E/$name -name "*kernel_trace.csv" -size +20M -delete
}
B="$GRAFT_REPO_ROOT/bench.py --channels 1024 --steps 100 --warmup 10 --no-cpu-baseline --no-extra-configs"
run serial $B || exit $?
run pipelined $B --pipelined || exit $?
run two_streams $GRAFT_REPO_ROOT/tools/two_streams.py 200 || exit $?
tail -n +1 $E/*_overlap.txt
