# bf16 bound pass: where the time goes (timing-only variants)
OUT=gpurun_out/r3h; mkdir -p $OUT
for v in 0 1 2 3; do
  CGE_PB_DIAG=$v timeout -k 5 200 python bench.py --steps 5 --warmup 2 --profile-all --no-cpu-baseline > $OUT/b_$v.log 2>&1
  python - <<PY
import json
txt=open('$OUT/b_$v.log').read()
b=json.loads([l for l in txt.splitlines() if l.startswith('{')][-1])
print('diag $v pcent ms', b['kernels']['pcent']['avg_launch_ms'], 'pairs', b['diameter']['candidate_landmark_pairs'])
PY
done
