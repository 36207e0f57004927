mkdir -p gpurun_out/r3
export SVX_BENCH_NOCHECK=1
B="--cpu_pairs 0 --cpu_procs 0 --e2e_pairs 0 --e2e_files 0 --extra_workloads 0 --steps 5 --warmup 2 --pipeline 0"
run() { tag=$1; shift; env "$@" timeout -k 10 300 python bench.py $B > gpurun_out/r3/v_$tag.json 2> gpurun_out/r3/v_$tag.err; python3 - gpurun_out/r3/v_$tag.json $tag <<'PY'
import json,sys
try:
    j=json.load(open(sys.argv[1])); s=j['stage_ms_per_step']
    print(sys.argv[2], round(j['value']), round(j['ms_per_step'],2), ' '.join('%s=%.2f'%(k,s[k]) for k in ('pyr0','pyr1','pyrN','pyr_aux','knob_sort')))
except Exception as e: print(sys.argv[2],'ERR',e)
PY
}
for t in e1 e3 e7; do run $t SVX_LIB=$PWD/speech-vecalign_amd/svx/libsvx_$t.so; done
