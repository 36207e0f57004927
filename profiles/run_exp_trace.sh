# kernel-trace timeline of the two-context experiment (scratch: output under gpurun_out/r3)
mkdir -p gpurun_out/r3
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r3/kt2 -- python3 $R/profiles/exp_multi_ctx.py --parts 2 --skip_single --stagger_ms 40 --steps 3 --warmup 2 > $R/gpurun_out/r3/kt2.log 2>&1
cd $R
python3 - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/r3/kt2/*/*kernel_trace.csv')[0]
rows = [r for r in csv.DictReader(open(f)) if 'k_' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
t0 = int(rows[0]['Start_Timestamp'])
import re
def short(k):
    k = k.replace('(anonymous namespace)::','').replace('void ','')
    return re.sub(r'\(.*', '', k)[:44]
with open('gpurun_out/r3/kt2_timeline.txt','w') as o:
    for r in rows:
        s=(int(r['Start_Timestamp'])-t0)/1e6; e=(int(r['End_Timestamp'])-t0)/1e6
        o.write("%9.3f %9.3f %7.3f q=%s %s\n" % (s, e, e-s, r.get('Queue_Id'), short(r['Kernel_Name'])))
PY
rm -rf gpurun_out/r3/kt2
