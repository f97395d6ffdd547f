cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r03n
timeout 200 python -m pytest tests/test_gpu_parity.py -q -k "host_call_in_overlapped_chunks or error_paths or boundary" 2>&1 | tail -2 | tee gpurun_out/r03n/test.txt
timeout 400 python bench.py > gpurun_out/r03n/bench.json 2> gpurun_out/r03n/err.txt
python -c "
import json; d=json.load(open('gpurun_out/r03n/bench.json')); print(d['value'], d['ms_per_step'], d['config']['bit_exact_full']); o=d['other_configs']
for k,v in o.items(): print(k, {a:b for a,b in v.items() if a in ('frames_per_s','value','ms','bit_exact_full','ms_per_step')} if isinstance(v,dict) else v)"
