# DEV TOOL (GPU box): smoke() + a short bench step with the md5 check -- the last look at HEAD's binaries in round 3 (gpurun_out/r03z/).
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r03z
timeout 40 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1 | tee gpurun_out/r03z/smoke.txt
timeout 40 python bench.py --cpu-seconds 0 --no-extras --steps 6 --warmup 2 2>/dev/null | python -c "import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['value'], d['ms_per_step'], d['config']['bit_exact_full'])" | tee gpurun_out/r03z/bench_short.txt
