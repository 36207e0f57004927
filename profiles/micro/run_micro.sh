#!/bin/bash
# Build and run a microbenchmark on the GPU box: bash profiles/micro/run_micro.sh <name> [args]  -> gpurun_out/micro_<name>.txt (appended)
set -e
N=$1; shift
mkdir -p gpurun_out
[ -x gpurun_out/$N.bin ] || /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -Wno-unused-value profiles/micro/$N.hip -o gpurun_out/$N.bin
timeout -k 10 180 gpurun_out/$N.bin "$@" >> gpurun_out/micro_$N.txt 2>&1 || echo "rc=$?" >> gpurun_out/micro_$N.txt
