# DEV TOOL (GPU box): the -DLHIP_TAIL_HELP build on `bursts` and mono (no regression where it cannot help); gpurun_out/r03th/.
cd $GRAFT_REPO_ROOT; O=gpurun_out/r03th; mkdir -p $O
export LAMEJS_HIP_LIB=$GRAFT_REPO_ROOT/lamejs_amd/lib/variants/liblamejs_hip_tailhelp.so
for c in bursts 2; do
  timeout 14 python bench.py --config $c --cpu-seconds 0 --no-extras --steps 4 --warmup 1 --check-frames 0 2>/dev/null | python -c "import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('tailhelp config $c', d['ms_per_step'], d['config']['bit_exact_full'], d['config'].get('seed_repaired_frames'))" | tee -a $O/third_run.txt
done
