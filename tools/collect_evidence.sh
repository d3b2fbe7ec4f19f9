#!/bin/bash
# collect a round's evidence on the GPU box: gpurun -- tools/collect_evidence.sh  (outputs under gpurun_out/ev/, copied into
# profiles/ by tools/publish_evidence.sh).  Every GPU step runs under a time limit; a step killed at its limit (a hang)
# ends the script: no further GPU process after it.
# The microbenchmarks it runs are built beforehand, here (no GPU needed):
#   for f in valu hbm_read mfma_fir corun; do hipcc -O3 -std=c++17 --offload-arch=gfx950 -o tools/ubench_$f.bin tools/ubench_$f.hip; done
#   hipcc -O3 --offload-arch=gfx950 -o tools/lab/pcie_d2h.bin tools/lab/pcie_d2h.hip
set -u
cd $GRAFT_REPO_ROOT
export GPU_MAX_HW_QUEUES=8      # (what bench.py sets for itself; under rocprofv3 the runtime starts before the program does)
E=gpurun_out/ev; rm -rf $E; mkdir -p $E
rm -f gpurun_out/parity_margins.jsonl gpurun_out/r05_parity_pairs.jsonl
step() { tools/run_gpu_step.sh "$@" || exit 1; }
step 900 $E/gpu_tests.log python -m pytest tests -m gpu -q
cp gpurun_out/parity_margins.jsonl gpurun_out/r05_parity_pairs.jsonl $E/ 2>/dev/null
step 400 $E/bench.log python bench.py
step 200 $E/bench_vector.log python bench.py --matrix 1 --no-cpu-baseline --no-extra-configs
step 200 $E/bench_vector_separate_tail.log python bench.py --matrix 1 --fused 1 --no-cpu-baseline --no-extra-configs
step 200 $E/bench_rccl_1rank.log python bench.py --force-dist --fanout rccl --no-cpu-baseline --no-extra-configs
step 200 $E/bench_host_1rank.log python bench.py --force-dist --fanout host --no-cpu-baseline --no-extra-configs
step 200 $E/bench_c4_host_1rank.log python bench.py --config c4 --force-dist --fanout host --steps 300 --no-cpu-baseline
step 200 $E/bench_c4_rccl_1rank.log python bench.py --config c4 --force-dist --fanout rccl --steps 300 --no-cpu-baseline
step 300 $E/bench_c4_2ranks_1gpu.log python bench.py --config c4 --gpus 2 --fanout host --ranks-on-one-gpu --steps 300 --no-cpu-baseline
step 300 $E/bench_multibank_2x5000.log python bench.py --multibank 2 --channels 5000 --steps 300
step 200 $E/bench_fused3.log python bench.py --matrix 3 --no-cpu-baseline --no-extra-configs
step 200 $E/bench_c4.log python bench.py --config c4 --steps 300 --no-cpu-baseline
step 200 $E/bench_c5.log python bench.py --config c5 --steps 300 --no-cpu-baseline
step 60 $E/ubench_valu.txt tools/ubench_valu.bin
step 60 $E/ubench_hbm_read.txt tools/ubench_hbm_read.bin
step 120 $E/ubench_mfma_fir.txt tools/ubench_mfma_fir.bin
step 120 $E/ubench_corun.txt tools/ubench_corun.bin
# rocprofv3: the program itself after "--"; a killed run ends the script like any other killed step
prof() {    # prof <name> <bench args...>
  local name=$1; shift
  ( cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$E/$name -- python3 $GRAFT_REPO_ROOT/bench.py "$@" --steps 400 --warmup 10 --no-cpu-baseline --no-extra-configs > $GRAFT_REPO_ROOT/$E/bench_under_rocprof_$name.log 2>&1 )
  local rc=$?
  echo "rocprof stats ($name) rc=$rc" >> gpurun_out/steps.log
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "rocprofv3 run killed at its time limit: stopping"; exit 99; fi
}
prof trace
python3 tools/trace_outliers.py $E/trace > $E/trace_outliers.txt 2>&1
prof trace_fused3 --matrix 3
prof trace_vector --matrix 1
./tools_pmc.sh ev/pmc_matrix > $E/pmc_matrix.log 2>&1 || exit 1
./tools_pmc.sh ev/pmc_fused3 --matrix 3 > $E/pmc_fused3.log 2>&1 || exit 1
./tools_pmc.sh ev/pmc_vector --matrix 1 > $E/pmc_vector.log 2>&1 || exit 1
# round 6: counters for BASELINE's own single-GPU configurations (configs[2]'s 1 024 channels = --config c4 at N = 1; 64 channels)
./tools_pmc.sh ev/pmc_c4 --config c4 > $E/pmc_c4.log 2>&1 || exit 1
./tools_pmc.sh ev/pmc_64 --channels 64 > $E/pmc_64.log 2>&1 || exit 1
# round 6: the drop-in entry's copy + kernel trace, the cut sweep behind its automatic choice, the PCIe link by itself
tools/lab/trace_host_entry.sh ev/host_entry > $E/host_entry.log 2>&1 || exit 1
step 300 $E/sweep_host_entry.txt python tools/lab/sweep_host_entry.py
step 120 $E/pcie_d2h.txt tools/lab/pcie_d2h.bin 10
find $E -name "*kernel_stats.csv" | head
# keep the merge small: drop the raw traces, keep the stats
find $E/trace $E/trace_vector $E/trace_fused3 -type f ! -name "*stats*" -delete 2>/dev/null
find $E -name "*counter_collection.csv" -delete; find $E -name "*agent_info.csv" -delete; find $E/host_entry -name "*.csv" -delete
du -sh $E
