#!/bin/bash
# copy what tools/collect_evidence.sh left under gpurun_out/ev/ into profiles/ under a tag: tools/publish_evidence.sh r02_c
set -e
cd "$(dirname "$0")/.."
T=$1; E=gpurun_out/ev; P=profiles
j() { grep '^{' "$1" | tail -1 > "$2"; }
j $E/bench.log $P/${T}_bench.json
j $E/bench_separate_tail.log $P/${T}_bench_separate_tail.json
j $E/bench_single_lane.log $P/${T}_bench_single_lane.json
j $E/bench_single_lane_separate_tail.log $P/${T}_bench_single_lane_separate_tail.json
j $E/bench_rccl_1rank.log $P/${T}_bench_rccl_1rank.json
j $E/bench_rccl_1rank_inline.log $P/${T}_bench_rccl_1rank_inline.json
j $E/bench_under_rocprof.log $P/${T}_bench_under_rocprof.json
j $E/bench_under_rocprof_separate_tail.log $P/${T}_bench_under_rocprof_single_lane_separate_tail.json
cp "$(ls -t $E/trace/runc/*_kernel_stats.csv | head -1)" $P/${T}_kernel_stats.csv
cp "$(ls -t $E/trace_sep/runc/*_kernel_stats.csv | head -1)" $P/${T}_kernel_stats_single_lane_separate_tail.csv
cp $E/pmc_fused/summary.txt $P/${T}_pmc_summary.txt;                 cp $E/pmc_fused/traffic.json $P/${T}_traffic.json
cp $E/pmc_separate/summary.txt $P/${T}_single_lane_separate_tail_pmc_summary.txt; cp $E/pmc_separate/traffic.json $P/${T}_single_lane_separate_tail_traffic.json
cp $E/pmc_pair_separate/summary.txt $P/${T}_pair_separate_tail_pmc_summary.txt;   cp $E/pmc_pair_separate/traffic.json $P/${T}_pair_separate_tail_traffic.json
cp $E/ubench_valu.txt $P/r02_ubench_valu.txt; cp $E/ubench_hbm_read.txt $P/r02_ubench_hbm_read.txt
python3 - <<'PY'
import json
old = json.load(open('profiles/r02_parity_margins.json'))
tests = {}
for l in open('gpurun_out/ev/parity_margins.jsonl'):
    r = json.loads(l); name = r.pop('test')
    if name in tests:
        for k, v in r.items():
            if isinstance(v, (int, float)) and isinstance(tests[name].get(k), (int, float)): tests[name][k] = max(tests[name][k], v)
            else: tests[name].setdefault(k, v)
    else:
        tests[name] = r
json.dump({"note": old["note"], "tests": tests}, open('profiles/r02_parity_margins.json', 'w'), indent=1)
PY
echo published $T
