# round 3: rocprofv3 evidence (kernel-trace stats + FETCH/WRITE PMC passes of the default bench command), the timeline of one
# step, and one bench line per BASELINE configuration
set -o pipefail
export TMPDIR=/tmp
OUT=gpurun_out/r3p; mkdir -p $OUT
bash profiles/run_profiles.sh r03 > $OUT/run_profiles.log 2>&1; echo "run_profiles rc=$?"
tail -n 5 $OUT/run_profiles.log
cp gpurun_out/prof_r03/summary_r03.md $OUT/ 2>/dev/null
python3 profiles/timeline_gaps.py gpurun_out/prof_r03/stats 0.5 > $OUT/timeline_gaps_headline.txt 2>&1
timeout -k 10 300 python bench.py > $OUT/bench_headline.json 2> $OUT/bench_headline.err; echo "headline rc=$?"
timeout -k 10 300 python bench.py --profile-all --no-cpu-baseline > $OUT/bench_headline_all_timers.json 2> /dev/null; echo "all timers rc=$?"
for w in cfg2 cfg3 cfg4; do
  timeout -k 10 300 python bench.py --workload $w > $OUT/bench_$w.json 2> $OUT/bench_$w.err; echo "$w rc=$?"
done
timeout -k 10 600 python bench.py --workload cfg5 --steps 3 --warmup 1 > $OUT/bench_cfg5.json 2> $OUT/bench_cfg5.err; echo "cfg5 rc=$?"
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace_cfg4 -- python3 bench.py --workload cfg4 --steps 1 --warmup 1 --no-cpu-baseline --no-back-to-back > $OUT/trace_cfg4.log 2>&1; echo "cfg4 trace rc=$?"
python3 profiles/timeline_gaps.py $OUT/trace_cfg4 0.5 > $OUT/timeline_gaps_cfg4.txt 2>&1
find gpurun_out -name "*_kernel_trace.csv" -size +6M -delete
find gpurun_out -name "*_counter_collection.csv" -size +6M -delete
