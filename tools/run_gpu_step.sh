#!/bin/bash
# helper for gpurun calls: run a GPU step under a time limit; stop the whole call if it was killed (never run a
# further GPU step after a hang), but keep going after an ordinary test failure.
#   tools/run_gpu_step.sh <seconds> <logfile> <command...>
lim=$1; log=$2; shift 2
timeout -k 10 "$lim" "$@" > "$log" 2>&1
rc=$?
echo "[$(date +%H:%M:%S)] rc=$rc : $*" | tee -a gpurun_out/steps.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "step killed at its time limit: stopping" ; exit 99; fi
exit 0
