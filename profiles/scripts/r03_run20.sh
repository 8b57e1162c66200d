set -o pipefail
mkdir -p gpurun_out/r3r
hipcc --offload-arch=gfx950 -O3 profiles/microbench_dpp_fmac.hip -o /tmp/mbdpp || exit 1
timeout -k 10 120 /tmp/mbdpp > gpurun_out/r3r/mbdpp.txt 2>&1; echo rc=$?
cat gpurun_out/r3r/mbdpp.txt
