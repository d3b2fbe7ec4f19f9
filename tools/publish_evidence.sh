#!/bin/bash
# copy what tools/collect_evidence.sh left under gpurun_out/ev/ into profiles/ under a tag: tools/publish_evidence.sh r03_a
set -e
cd "$(dirname "$0")/.."
T=$1; E=gpurun_out/ev; P=profiles
j() { grep '^{' "$1" | tail -1 > "$2"; }
j $E/bench.log $P/${T}_bench.json
j $E/bench_vector.log $P/${T}_bench_vector_kernels.json
j $E/bench_vector_separate_tail.log $P/${T}_bench_vector_kernels_separate_tail.json
j $E/bench_rccl_1rank.log $P/${T}_bench_rccl_1rank.json
j $E/bench_host_1rank.log $P/${T}_bench_host_fanout_1rank.json
j $E/bench_c4_host_1rank.log $P/${T}_bench_config_c4_host_fanout_1rank.json
j $E/bench_c4_rccl_1rank.log $P/${T}_bench_config_c4_rccl_1rank.json
j $E/bench_c4_2ranks_1gpu.log $P/${T}_bench_config_c4_2ranks_on_one_gpu.json
j $E/bench_multibank_2x5000.log $P/${T}_bench_multibank_2x5000.json
j $E/bench_fused3.log $P/${T}_bench_fused_stage3.json
j $E/bench_c4.log $P/${T}_bench_config_c4.json
j $E/bench_c5.log $P/${T}_bench_config_c5.json
j $E/bench_under_rocprof_trace_fused3.log $P/${T}_bench_under_rocprof_fused_stage3.json
cp "$(ls -t $E/trace_fused3/runc/*_kernel_stats.csv | head -1)" $P/${T}_kernel_stats_fused_stage3.csv
cp $E/pmc_fused3/summary.txt $P/${T}_fused_stage3_pmc_summary.txt;  cp $E/pmc_fused3/traffic.json $P/${T}_fused_stage3_traffic.json
cp $E/trace_outliers.txt $P/${T}_trace_outliers.txt
cp $E/r05_parity_pairs.jsonl $P/${T}_parity_pairs.jsonl
j $E/bench_under_rocprof_trace.log $P/${T}_bench_under_rocprof.json
j $E/bench_under_rocprof_trace_vector.log $P/${T}_bench_under_rocprof_vector_kernels.json
cp "$(ls -t $E/trace/runc/*_kernel_stats.csv | head -1)" $P/${T}_kernel_stats.csv
cp "$(ls -t $E/trace_vector/runc/*_kernel_stats.csv | head -1)" $P/${T}_kernel_stats_vector_kernels.csv
cp $E/pmc_matrix/summary.txt $P/${T}_pmc_summary.txt;                 cp $E/pmc_matrix/traffic.json $P/${T}_traffic.json
cp $E/pmc_vector/summary.txt $P/${T}_vector_kernels_pmc_summary.txt;  cp $E/pmc_vector/traffic.json $P/${T}_vector_kernels_traffic.json
cp $E/ubench_valu.txt $P/${T}_ubench_valu.txt; cp $E/ubench_hbm_read.txt $P/${T}_ubench_hbm_read.txt; cp $E/ubench_mfma_fir.txt $P/${T}_ubench_mfma_fir.txt; cp $E/ubench_corun.txt $P/${T}_ubench_corun.txt
cp $E/pmc_c4/summary.txt $P/${T}_config_c4_pmc_summary.txt;          cp $E/pmc_c4/traffic.json $P/${T}_config_c4_traffic.json
cp $E/pmc_64/summary.txt $P/${T}_64_channels_pmc_summary.txt;        cp $E/pmc_64/traffic.json $P/${T}_64_channels_traffic.json
cp $E/host_entry/timeline.txt $P/${T}_host_entry_timeline.txt; cp $E/sweep_host_entry.txt $P/${T}_host_entry_sweep.txt; cp $E/pcie_d2h.txt $P/${T}_pcie_d2h.txt
tail -3 $E/gpu_tests.log > $P/${T}_gpu_tests_tail.txt
python3 - "$T" <<'PY'
import json, sys
tests = {}
for l in open('gpurun_out/ev/parity_margins.jsonl'):
    r = json.loads(l); name = r.pop('test')
    if name in tests:
        for k, v in r.items():
            if isinstance(v, (int, float)) and isinstance(tests[name].get(k), (int, float)): tests[name][k] = max(tests[name][k], v)
            else: tests[name].setdefault(k, v)
    else:
        tests[name] = r
note = ("worst relative error each GPU parity test saw against the oracle / the golden vectors (comparator of tests/conftest.py: "
        "|got - ref| / max(|ref|, rms of the channel)); tolerance 1e-5")
json.dump({"note": note, "tests": tests}, open('profiles/%s_parity_margins.json' % sys.argv[1], 'w'), indent=1)
PY
echo published $T
