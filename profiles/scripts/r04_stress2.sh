# offline stress (round 4, after the ordering fixes): new seeds of the randomised parity sweep, and the parity + configuration
# suites (a) with two lanes forced on every runsplit (CGE_LANES=2: the cross-lane wait for the member lists) and (b) under the
# start-skew knob of the persistent fits
set -o pipefail
OUT=gpurun_out/r04_stress2; mkdir -p $OUT
CGE_STRESS_OFFSET=700 timeout -k 10 500 python -m pytest tests/test_gpu_parity.py -q -k "randomised_parity" --timeout 60 > $OUT/stress_700.log 2>&1; rc=$?
echo "offset 700 rc=$rc: $(tail -n 1 $OUT/stress_700.log)"
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT: stopping"; exit 1; fi
CGE_LANES=2 timeout -k 10 700 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -q -m gpu --deselect tests/test_gpu_configs.py::test_config5_full_size_ten_million_vertices > $OUT/lanes2.log 2>&1; rc=$?
echo "suites under CGE_LANES=2 rc=$rc: $(tail -n 1 $OUT/lanes2.log)"
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT: stopping"; exit 1; fi
CGE_FIT_TEST_DELAY=20 timeout -k 10 700 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -q -m gpu --deselect tests/test_gpu_configs.py::test_config5_full_size_ten_million_vertices > $OUT/delay.log 2>&1; rc=$?
echo "suites under CGE_FIT_TEST_DELAY=20 rc=$rc: $(tail -n 1 $OUT/delay.log)"
