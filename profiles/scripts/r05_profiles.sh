# round 5: rocprofv3 evidence (kernel-trace stats + FETCH/WRITE PMC passes of the default bench command), the timeline of one
# step, the default bench line (with every other BASELINE configuration in `other_configs`), the fused-chain A/B and the
# in-kernel stamps of the fused fit.  Run on the GPU box: gpurun --timeout 1200 -- bash profiles/scripts/r05_profiles.sh
set -o pipefail
export TMPDIR=/tmp
OUT=gpurun_out/r5p; mkdir -p $OUT
bash profiles/run_profiles.sh r05 --no-other-configs > $OUT/run_profiles.log 2>&1; echo "run_profiles rc=$?"
tail -n 3 $OUT/run_profiles.log
cp gpurun_out/prof_r05/summary_r05.md gpurun_out/prof_r05/pmc_traffic_r05.json gpurun_out/prof_r05/bench_stats.json $OUT/ 2>/dev/null
python3 profiles/timeline_gaps.py gpurun_out/prof_r05/stats 0.5 > $OUT/timeline_gaps_headline.txt 2>&1
timeout -k 10 500 python bench.py > $OUT/bench_headline.json 2> $OUT/bench_headline.err; echo "headline rc=$?"
timeout -k 10 200 python profiles/fused_chain_ab.py headline 5 > $OUT/fused_chain_ab.txt 2>&1; echo "fused ab rc=$?"
if [ -f cge.jl_amd/csrc/build/libcge_hip_clock.so ]; then
  timeout -k 10 200 python profiles/flow_clock_probe.py headline > $OUT/flow_clock_raw.txt 2>&1; echo "flow clock rc=$?"
  grep "flow clock\|score" $OUT/flow_clock_raw.txt | tail -30 > $OUT/flow_clock.txt
fi
find gpurun_out -name "*_kernel_trace.csv" -size +6M -delete
find gpurun_out -name "*_counter_collection.csv" -size +6M -delete
