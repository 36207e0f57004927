#!/bin/bash
# The long GPU-vs-oracle sweeps of a round's final code -> gpurun_out/fuzz.txt (copied to profiles/rNN_fuzz_gpu_vs_oracle.txt).
# Run on the GPU box from the repo root: bash profiles/run_fuzz.sh
mkdir -p gpurun_out
O=gpurun_out/fuzz.txt
: > $O
run() { echo "## python tests/fuzz_gpu_vs_oracle.py $*" >> $O; timeout -k 10 500 python tests/fuzz_gpu_vs_oracle.py "$@" 2>&1 | grep -v amdgpu.ids | grep -E "MISMATCH|^penalty knife-edge|^penalty differs|^fuzz|^  " >> $O; echo "rc=${PIPESTATUS[0]}" >> $O; }
run --cases 6400 --seed 2026
run --cases 800 --seed 7 --max_size 6000
run --search straight --cases 4800 --seed 5
run --search straight --cases 600 --seed 9 --max_size 3000
run --cases 3200 --seed 77 --pipeline
cat $O
