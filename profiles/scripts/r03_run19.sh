set -o pipefail
mkdir -p gpurun_out/r3r
hipcc --offload-arch=gfx950 -O3 profiles/microbench_hbm_rows.hip -o /tmp/mbrows || exit 1
timeout -k 10 200 /tmp/mbrows > gpurun_out/r3r/mbrows.txt 2>&1; echo rc=$?
cat gpurun_out/r3r/mbrows.txt
