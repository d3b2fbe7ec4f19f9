#!/bin/bash
# collect a round's evidence on the GPU box: gpurun -- tools/collect_evidence.sh  (outputs under gpurun_out/ev/, copied into profiles/ by hand)
set -u
cd $GRAFT_REPO_ROOT
E=gpurun_out/ev; rm -rf $E; mkdir -p $E
rm -f gpurun_out/parity_margins.jsonl
tools/run_gpu_step.sh 900 $E/gpu_tests.log python -m pytest tests -m gpu -q || exit 1
cp gpurun_out/parity_margins.jsonl $E/ 2>/dev/null
tools/run_gpu_step.sh 400 $E/bench.log python bench.py || exit 1
tools/run_gpu_step.sh 200 $E/bench_separate_tail.log python bench.py --fused 1 --no-cpu-baseline --no-extra-configs || exit 1
tools/run_gpu_step.sh 200 $E/bench_single_lane.log python bench.py --pair 1 --no-cpu-baseline --no-extra-configs || exit 1
tools/run_gpu_step.sh 200 $E/bench_single_lane_separate_tail.log python bench.py --pair 1 --fused 1 --no-cpu-baseline --no-extra-configs || exit 1
tools/run_gpu_step.sh 200 $E/bench_rccl_1rank.log python bench.py --force-dist --no-cpu-baseline --no-extra-configs || exit 1
tools/run_gpu_step.sh 200 $E/bench_rccl_1rank_inline.log python bench.py --force-dist --fanout inline --no-cpu-baseline --no-extra-configs || exit 1
tools/run_gpu_step.sh 60 $E/ubench_valu.txt tools/ubench_valu.bin || exit 1
tools/run_gpu_step.sh 60 $E/ubench_hbm_read.txt tools/ubench_hbm_read.bin || exit 1
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$E/trace -- python3 $GRAFT_REPO_ROOT/bench.py --steps 400 --warmup 10 --no-cpu-baseline --no-extra-configs > $GRAFT_REPO_ROOT/$E/bench_under_rocprof.log 2>&1 ); echo "rocprof stats rc=$?" >> gpurun_out/steps.log
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$E/trace_sep -- python3 $GRAFT_REPO_ROOT/bench.py --pair 1 --fused 1 --steps 400 --warmup 10 --no-cpu-baseline --no-extra-configs > $GRAFT_REPO_ROOT/$E/bench_under_rocprof_separate_tail.log 2>&1 ); echo "rocprof stats (sep) rc=$?" >> gpurun_out/steps.log
./tools_pmc.sh ev/pmc_fused > $E/pmc_fused.log 2>&1
./tools_pmc.sh ev/pmc_separate --pair 1 --fused 1 > $E/pmc_separate.log 2>&1
./tools_pmc.sh ev/pmc_pair_separate --fused 1 > $E/pmc_pair_separate.log 2>&1
find $E -name "*kernel_stats.csv" | head
# keep the merge small: drop the raw traces, keep the stats
find $E/trace $E/trace_sep -type f ! -name "*stats*" -delete 2>/dev/null
find $E -name "*counter_collection.csv" -delete; find $E -name "*agent_info.csv" -delete
du -sh $E
