mkdir -p gpurun_out/r3
time timeout -k 10 900 python bench.py > gpurun_out/r3/bench_default.json 2> gpurun_out/r3/bench_default.err; echo "rc=$?"
true
python3 - <<'PY'
import json
j=json.load(open('gpurun_out/r3/bench_default.json'))
print(round(j['value']), round(j['ms_per_step'],2), j['roofline']['stage'], round(j['roofline']['frac'],3), j['roofline'].get('traffic'), j.get('parity'))
for k,v in j.get('workloads',{}).items():
    print(k, {kk:(round(vv,3) if isinstance(vv,float) else vv) for kk,vv in v.items() if kk in ('value','ms_per_step','seconds_per_pair','leg_seconds','error')}, (v.get('roofline') or {}).get('frac'))
PY
