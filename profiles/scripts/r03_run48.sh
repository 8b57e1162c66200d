# stress: every persistent fit of the full-size configurations and of the parity suite with a forced start skew of ~60 us
set -o pipefail
OUT=gpurun_out/r4t; mkdir -p $OUT
export CGE_FIT_TEST_DELAY=20
timeout -k 10 500 python -m pytest tests/test_gpu_configs.py -q -x -k "not config5" > $OUT/t_cfg.log 2>&1; echo "cfg rc=$? $(tail -n 1 $OUT/t_cfg.log)"
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -q -x -k "not landmarks_parity" > $OUT/t_par.log 2>&1; echo "parity rc=$? $(tail -n 1 $OUT/t_par.log)"
timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-back-to-back 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench with skew', round(j['ms_per_step'],2), j['result'][:2])"
