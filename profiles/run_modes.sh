#!/bin/bash
# Mode B / Mode C evidence: bench lines and rocprofv3 kernel-trace statistics of `bench.py --workload c4|dense`.
# Run on the GPU box from the repo root: bash profiles/run_modes.sh  -> gpurun_out/modes/
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/modes
mkdir -p $O
cd $R
python3 bench.py --workload c4 --steps 5 --warmup 2 > $O/bench_c4.json 2> $O/bench_c4.err; echo c4 rc=$?
python3 bench.py --workload dense --steps 3 --warmup 1 > $O/bench_dense.json 2> $O/bench_dense.err; echo dense rc=$?
python3 bench.py --workload c3 --steps 4 --warmup 2 > $O/bench_c3.json 2> $O/bench_c3.err; echo c3 rc=$?
cd /tmp && export TMPDIR=/tmp
ARGS="--cpu_pairs 0 --cpu_procs 0 --e2e_pairs 0 --e2e_files 0"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_c4 -- python3 $R/bench.py --workload c4 --steps 3 --warmup 1 $ARGS > $O/kt_c4.log 2>&1; echo kt_c4 rc=$?
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_dense -- python3 $R/bench.py --workload dense --steps 2 --warmup 1 $ARGS > $O/kt_dense.log 2>&1; echo kt_dense rc=$?
rm -f $O/kt_*/*/*kernel_trace.csv
du -sh $O
