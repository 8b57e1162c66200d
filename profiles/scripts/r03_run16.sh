#!/bin/bash
# chunk-count sweep of the per-edge scatter alone (profiles/scatter_probe.py)
set -o pipefail
mkdir -p gpurun_out/r3n
for ch in 0 850 1000 1024 1100 1280 1536 0; do
  echo "CGE_EB_CHUNKS=$ch" >> gpurun_out/r3n/probe.log
  CGE_EB_CHUNKS=$ch timeout -k 10 200 python profiles/scatter_probe.py 300 >> gpurun_out/r3n/probe.log 2>&1 || exit 1
done
cat gpurun_out/r3n/probe.log
