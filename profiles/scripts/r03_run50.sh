set -o pipefail
export TMPDIR=/tmp
OUT=gpurun_out/r4v; mkdir -p $OUT
timeout -k 10 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/pmc -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-back-to-back > $OUT/pmc.log 2>&1; echo "rc=$?"
python3 profiles/pmc_wait_breakdown.py $OUT/pmc > $OUT/wait_breakdown.txt 2>&1; cat $OUT/wait_breakdown.txt
find $OUT -name "*_counter_collection.csv" -size +6M -delete; find $OUT -name "*_kernel_trace.csv" -size +6M -delete
