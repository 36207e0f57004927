#!/bin/bash
# Kernel-trace timeline of a few bench steps: bash profiles/run_timeline.sh <tag> [bench.py arguments]
# -> gpurun_out/r3/timeline_<tag>.txt: start, end, duration (ms), queue, kernel, one line per dispatch of this repo's kernels.
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/r3
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r3/kt_$TAG -- python3 $R/bench.py --steps 3 --warmup 1 --no_profile --cpu_pairs 0 --cpu_procs 0 --e2e_pairs 0 --e2e_files 0 --extra_workloads 0 "$@" > $R/gpurun_out/r3/kt_$TAG.log 2>&1
cd $R
python3 - $TAG <<'PY'
import csv, glob, re, sys
tag = sys.argv[1]
f = glob.glob('gpurun_out/r3/kt_%s/*/*kernel_trace.csv' % tag)[0]
rows = [r for r in csv.DictReader(open(f)) if 'k_' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
t0 = int(rows[0]['Start_Timestamp'])
def short(k):
    k = k.replace('(anonymous namespace)::', '').replace('void ', '')
    return re.sub(r'\(.*', '', k)[:44]
with open('gpurun_out/r3/timeline_%s.txt' % tag, 'w') as o:
    for r in rows:
        s = (int(r['Start_Timestamp']) - t0) / 1e6; e = (int(r['End_Timestamp']) - t0) / 1e6
        o.write("%9.3f %9.3f %7.3f q=%s %s\n" % (s, e, e - s, r.get('Queue_Id'), short(r['Kernel_Name'])))
PY
rm -rf gpurun_out/r3/kt_$TAG
