#!/bin/bash
# chunk-count sweep of the per-edge scatter (CGE_EB_CHUNKS), headline workload
set -o pipefail
mkdir -p gpurun_out/r3n
for ch in 0 1024 1536 2048; do
  CGE_EB_CHUNKS=$ch timeout -k 10 300 python bench.py --steps 5 --warmup 2 > gpurun_out/r3n/b_$ch.log 2>&1 || exit 1
  python - <<PY
import json
j=json.loads(open("gpurun_out/r3n/b_$ch.log").read().strip().splitlines()[-1])
k=j["kernels"]["edge_scatter"]; w=j["kernels"]["edge_scatter_wedges"]
print("chunks_env", $ch, "chunks", k.get("chunks"), "edge_scatter_ms", round(k["avg_launch_ms"],4), "frac", round(k["frac"],3), "wedges_ms", round(w["avg_launch_ms"],4), "step", round(j["ms_per_step"],2))
PY
done
