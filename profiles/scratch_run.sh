mkdir -p gpurun_out/r3
B="--cpu_pairs 0 --cpu_procs 0 --e2e_pairs 0 --e2e_files 0 --extra_workloads 0 --steps 8 --warmup 2 --pipeline 1"
run() { tag=$1; shift; env "$@" timeout -k 10 300 python bench.py $B > gpurun_out/r3/v_$tag.json 2> gpurun_out/r3/v_$tag.err; python3 - gpurun_out/r3/v_$tag.json $tag <<'PY'
import json,sys
try:
    j=json.load(open(sys.argv[1])); s=j['stage_ms_per_step']
    print(sys.argv[2], round(j['value']), round(j['ms_per_step'],2), ' '.join('%s=%.2f'%(k,s[k]) for k in ('pyr0','pyr1','pyrN','knob_sort','knob_scoresN','knob_scores0','band_costs0','band_costsN')))
except Exception as e: print(sys.argv[2],'ERR',e)
PY
}
run sp2 SVX_PIPE_SPLIT=2
run sp4 SVX_PIPE_SPLIT=4
run sp3 SVX_PIPE_SPLIT=3
timeout -k 10 300 python -m pytest tests/test_gpu_pipeline.py -x -q -k "pipeline or stack or golden" 2>&1 | tail -2
