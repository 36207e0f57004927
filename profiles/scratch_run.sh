mkdir -p gpurun_out/r3
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r3/t2.log 2>&1; echo "rc=$?" >> gpurun_out/r3/t2.log
tail -6 gpurun_out/r3/t2.log
timeout -k 10 600 python bench.py > gpurun_out/r3/b2.json 2> gpurun_out/r3/b2.err; echo "bench rc=$?"
python3 - <<'PY'
import json
j=json.load(open('gpurun_out/r3/b2.json'))
print(round(j['value']), round(j['ms_per_step'],2), j['roofline']['stage'], round(j['roofline']['frac'],3), j.get('parity'), {k:round(v['value']) for k,v in j.get('end_to_end',{}).items()}, j['cpu_baseline']['value'])
PY
