#!/bin/bash
# Collect the rocprofv3 evidence bench.py's roofline object refers to.  Run on the GPU box from the repo root:
#     bash profiles/run_profiles.sh <tag> [extra bench.py arguments]
# -> gpurun_out/prof_<tag>/{kt,fetch,write}: kernel-trace statistics of the default bench.py command (its timed
#    region and warm-up), and the FETCH_SIZE / WRITE_SIZE counters of one warm-up + one timed step at the SAME
#    pairs per step.  Counters are collected in their own passes (never together with a trace), and the program
#    itself follows `--` (no env / bash -c hop: the profiler has initialised the GPU before the program starts).
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; shift
O=$R/gpurun_out/prof_$TAG
mkdir -p $O
ARGS="--cpu_pairs 0 --cpu_procs 0 --e2e_pairs 0 --e2e_files 0 --extra_workloads 0 $*"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 $R/bench.py $ARGS > $O/kt.log 2>&1; echo exit=$? >> $O/kt.log
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 $R/bench.py --steps 1 --warmup 1 --no_profile $ARGS > $O/fetch.log 2>&1; echo exit=$? >> $O/fetch.log
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 $R/bench.py --steps 1 --warmup 1 --no_profile $ARGS > $O/write.log 2>&1; echo exit=$? >> $O/write.log
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/sq -- python3 $R/bench.py --steps 1 --warmup 1 --no_profile $ARGS > $O/sq.log 2>&1; echo exit=$? >> $O/sq.log
rm -f $O/kt/*/*kernel_trace.csv
tail -2 $O/kt.log | cut -c1-300; tail -1 $O/fetch.log | cut -c1-200; tail -1 $O/write.log | cut -c1-200; tail -1 $O/sq.log | cut -c1-200
du -sh $O
