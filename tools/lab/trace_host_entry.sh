#!/bin/bash
# Copy + kernel trace of the drop-in entry.  On the GPU box: tools/lab/trace_host_entry.sh <outdir> [k=v ...]
E=$GRAFT_REPO_ROOT/gpurun_out/${1:-r6_trace_host}
shift
mkdir -p $E
cd /tmp && export TMPDIR=/tmp
export SIDE_JSON=$E/parts.json
timeout -k 10 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $E/trace -- python3 $GRAFT_REPO_ROOT/tools/lab/host_entry_calls.py 6 "$@" > $E/calls.log 2>&1
rc=$?
echo "trace rc=$rc"
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "killed at its time limit: stopping"; exit 99; fi
python3 $GRAFT_REPO_ROOT/tools/trace_host_entry.py $E/trace $E/parts.json > $E/timeline.txt 2>&1
find $E/trace -name "*.csv" -size +20M -delete
cat $E/calls.log $E/timeline.txt
