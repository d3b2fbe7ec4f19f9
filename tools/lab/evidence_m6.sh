#!/bin/bash
# Evidence of k_stage12m6 in one gpurun call -> gpurun_out/r05_m6.txt: the kernel's time against the vector kernel's at several bank
# sizes and channel lists (bench.py --rate 6; tools/lab/bench_m6_lists.py), rocprofv3 --kernel-trace --stats of the 10 000-channel
# run, and the stress fixtures' table (tests/test_gpu_matrix6.py writes gpurun_out/parity_margins.jsonl).
cd $GRAFT_REPO_ROOT
out=gpurun_out/r05_m6.txt
{
echo "k_stage12m6 (sdrx_amd/csrc/kernels_mfma6.hpp), round 5.  One MI355X box, one call (tools/lab/evidence_m6.sh)."
echo
echo "(1) bench.py --rate 6 --channels N --matrix 1|2 --steps 200: ms per 32 ms block (192 000 samples), kernel ms from HIP events"
bash tools/lab/bench_m6.sh 10000 4096 2048 1024 512
echo
echo "(2) other channel lists (tools/lab/bench_m6_lists.py): vector / matrix, ms per block"
timeout -k 10 300 python3 tools/lab/bench_m6_lists.py
echo
echo "(3) rocprofv3 --kernel-trace --stats -- python3 bench.py --rate 6 --channels 10000 --matrix 2 --steps 200"
} > $out 2>&1
export TMPDIR=/tmp
rm -rf gpurun_out/prof_m6
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_m6 -o m6 --output-format csv -- python3 bench.py --rate 6 --channels 10000 --matrix 2 --steps 200 --no-cpu-baseline --no-extra-configs > gpurun_out/prof_m6.log 2>&1
f=$(find gpurun_out/prof_m6 -name '*kernel_stats.csv' | head -1)
{
[ -n "$f" ] && head -8 "$f" | cut -c1-260
echo
echo "(4) stress fixtures of the 6 MS/s chain (tests/golden/pairs6, reference compiled here, both builds): relative error against the portable build"
timeout -k 10 300 python3 -m pytest tests/test_gpu_matrix6.py -q -m gpu 2>&1 | tail -2
python3 - <<'PY'
import json
seen = {}
for l in open("gpurun_out/parity_margins.jsonl"):
    d = json.loads(l)
    if d.get("test", "").startswith("pairs6"):
        seen[d["test"]] = d
print("%-28s %-22s %-34s %-34s" % ("case", "reference's two builds", "k_stage12m6 (x spread)", "vector kernels (x spread)"))
for k, d in sorted(seen.items()):
    print("%-28s iq %.2e au %.2e   iq %.2e (%.2f) au %.2e (%.2f)   iq %.2e (%.2f) au %.2e (%.2f)" % (
        k[7:-1], d["ref_avx2_iq"], d["ref_avx2_audio"], d["matrix_iq"], d["matrix_iq"] / d["ref_avx2_iq"], d["matrix_audio"],
        d["matrix_audio"] / d["ref_avx2_audio"], d["vector_iq"], d["vector_iq"] / d["ref_avx2_iq"], d["vector_audio"], d["vector_audio"] / d["ref_avx2_audio"]))
PY
} >> $out 2>&1
rm -rf gpurun_out/prof_m6
cat $out
