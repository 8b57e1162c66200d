#!/bin/bash
# build/libcge_hip_clock.so: the library with the persistent fit's in-kernel stamps (CGE_FLOW_CLOCK; profiles/flow_clock_probe.py)
# -- diagnostics only, not part of `make`
set -e
cd "$(dirname "$0")"
make -s
F="-O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function -Wno-unused-result --offload-arch=gfx950 -munsafe-fp-atomics -DCGE_FLOW_CLOCK"
/opt/rocm/bin/hipcc $F -c kernels_fitp.hip -o build/kernels_fitp_clock.o
OBJS=$(ls build/*.o | grep -v "_clock.o" | grep -v "kernels_fitp.o")
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o build/libcge_hip_clock.so $OBJS build/kernels_fitp_clock.o -lpthread -ldl
ls -la build/libcge_hip_clock.so
