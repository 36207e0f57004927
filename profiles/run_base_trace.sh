set -e
mkdir -p gpurun_out/r3
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r3/kt0 -- python3 $R/bench.py --steps 1 --warmup 1 --no_profile --cpu_pairs 0 --cpu_procs 0 --e2e_pairs 0 --e2e_files 0 > $R/gpurun_out/r3/kt0.log 2>&1
cd $R
python3 - <<'PY'
import csv, glob, collections
f = glob.glob('gpurun_out/r3/kt0/*/*kernel_trace.csv')[0]
rows = list(csv.DictReader(open(f)))
print(rows[0].keys())
agg = collections.OrderedDict()
for r in rows:
    k = r['Kernel_Name']
    if 'k_' not in k: continue
    a = agg.setdefault(k, dict(n=0, ns=0, r=r))
    a['n'] += 1; a['ns'] += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
with open('gpurun_out/r3/kt0_summary.txt','w') as o:
    for k,a in agg.items():
        r = a['r']
        o.write("%-70s n=%d avg_us=%.1f grid=%s wg=%s lds=%s vgpr=%s accum=%s sgpr=%s scratch=%s\n" % (k.replace('void ','')[:70], a['n'], a['ns']/a['n']/1e3, r.get('Grid_Size_X'), r.get('Workgroup_Size_X'), r.get('LDS_Block_Size'), r.get('VGPR_Count'), r.get('Accum_VGPR_Count'), r.get('SGPR_Count'), r.get('Scratch_Size')))
    # timeline of the second step (timed): start offsets
    t0 = int(rows[0]['Start_Timestamp'])
    o.write("\n# timeline (ms since first kernel): start end name stream/queue\n")
    for r in rows:
        k = r['Kernel_Name']
        if 'k_' not in k: continue
        o.write("%.3f %.3f %s q=%s\n" % ((int(r['Start_Timestamp'])-t0)/1e6, (int(r['End_Timestamp'])-t0)/1e6, k.replace('void ','')[:60], r.get('Queue_Id')))
PY
rm -rf gpurun_out/r3/kt0
