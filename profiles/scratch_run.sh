mkdir -p gpurun_out/r3
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r3/t_final.log 2>&1; echo "rc=$?" >> gpurun_out/r3/t_final.log; tail -4 gpurun_out/r3/t_final.log
O=gpurun_out/fuzz_r2seeds.txt; : > $O
run() { echo "## python tests/fuzz_gpu_vs_oracle.py $*" >> $O; timeout -k 10 500 python tests/fuzz_gpu_vs_oracle.py "$@" 2>&1 | grep -v amdgpu.ids | grep -E "MISMATCH|^penalty knife-edge|^penalty differs|^fuzz|^  " >> $O; echo "rc=${PIPESTATUS[0]}" >> $O; }
run --cases 16000 --seed 31
run --search straight --cases 12000 --seed 77
cat $O | cut -c1-400
