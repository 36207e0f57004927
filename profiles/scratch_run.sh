mkdir -p gpurun_out/r3
timeout -k 10 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
B="--cpu_pairs 0 --cpu_procs 0 --e2e_pairs 0 --e2e_files 0 --extra_workloads 0 --steps 6 --warmup 2"
run() { tag=$1; shift; env "$@" timeout -k 10 300 python bench.py $B $PIPE > gpurun_out/r3/v_$tag.json 2> gpurun_out/r3/v_$tag.err; python3 - gpurun_out/r3/v_$tag.json $tag <<'PY'
import json,sys
try:
    j=json.load(open(sys.argv[1])); s=j['stage_ms_per_step']
    print(sys.argv[2], round(j['value']), round(j['ms_per_step'],2), ' '.join('%s=%.2f'%(k,s[k]) for k in ('pyr0','pyr1','pyrN','pyr_aux','knob_sort','knob_scoresN','knob_scores0','band_costs0','band_costsN')))
except Exception as e: print(sys.argv[2],'ERR',e)
PY
}
PIPE="--pipeline 0"
run st_p0 X=1
PIPE="--pipeline 1"
run st_p1 X=1
