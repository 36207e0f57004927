mkdir -p gpurun_out/r3
timeout -k 10 300 python -m pytest tests -m gpu -x -q -k "tile or straight or band or dense" 2>&1 | tail -2
timeout -k 10 200 python bench.py --workload c4 --pairs 1 --steps 3 --warmup 1 2>/dev/null | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print('c4x1', round(j['ms_per_step'],1), j['alignments_cover_both_documents'], {k:round(v,1) for k,v in j['stage_ms_per_step'].items() if k in ('tiles','traceback0')})"
timeout -k 10 200 python bench.py --workload c4 --pairs 8 --steps 2 --warmup 1 2>/dev/null | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print('c4x8', round(j['ms_per_step'],1), round(j['value'],1))"
timeout -k 10 200 python bench.py --workload dense --pairs 64 --steps 2 --warmup 1 2>/dev/null | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print('dense', round(j['ms_per_step'],1), round(j['value'],1))"
