set -o pipefail
OUT=gpurun_out/r4n; mkdir -p $OUT
CGE_FIT_FUSED_DEBUG=1 timeout -k 10 200 python -m pytest tests/test_gpu_parity.py -q -x -s -k "randomised_parity_sweep and 16" > $OUT/dbg.log 2>&1; echo "rc=$?"
grep -c "mismatch" $OUT/dbg.log; grep "mismatch" $OUT/dbg.log | head -12; tail -n 3 $OUT/dbg.log
