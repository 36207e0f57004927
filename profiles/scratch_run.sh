mkdir -p gpurun_out/r3
SVX_BENCH_BACKEND=gloo timeout -k 10 400 python bench.py --gpus 2 --pairs 256 --steps 4 --warmup 2 --cpu_pairs 0 --cpu_procs 0 --e2e_pairs 0 --e2e_files 0 > gpurun_out/r3/bench_2ranks_gloo.json 2> gpurun_out/r3/bench_2ranks_gloo.err; echo "rc=$?"
tail -3 gpurun_out/r3/bench_2ranks_gloo.err | cut -c1-300
python3 -c "
import json
for l in open('gpurun_out/r3/bench_2ranks_gloo.json'):
    l=l.strip()
    if l.startswith('{'):
        j=json.loads(l); print(j['n_gpus'], round(j['value']), round(j['ms_per_step'],1), j['config']['pairs_per_step_per_gpu'], j['config']['parallelism'])
"
SVX_BENCH_BACKEND=gloo timeout -k 10 400 python bench.py --gpus 2 --workload c3 --pairs 128 --steps 2 --warmup 1 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        j=json.loads(l); print('c3', j['n_gpus'], round(j['value']), round(j['ms_per_step'],1))
"
