# DEV TOOL (GPU box): first device run of the tail-help experiment (lamejs_amd/lib/variants/liblamejs_hip_tailhelp*.so, built by
# tools/build_tailhelp_variants.sh): bench step (stereo 128k, 1e5 frames: md5 against the reference's table) and the host-buffer call, every
# command under its own timeout (the poll loops fault after 2^24 rounds).  gpurun_out/r03th/.
cd $GRAFT_REPO_ROOT; O=gpurun_out/r03th; mkdir -p $O
for v in tailhelp ${TAILHELP_MORE_VARIANTS}; do   # round 3 also ran a -DLHIP_TAIL_NOINLINE build (one copy of the unit behind a call): a third slower, removed
  export LAMEJS_HIP_LIB=$GRAFT_REPO_ROOT/lamejs_amd/lib/variants/liblamejs_hip_$v.so
  b=$(timeout 60 python bench.py --cpu-seconds 0 --no-extras --steps 6 --warmup 2 2>$O/err_$v.txt | python -c "import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['ms_per_step'], d['config']['bit_exact_full'], d['config'].get('seed_repaired_frames'))" 2>&1 | tail -1)
  echo "$v  bench ms_per_step, bit_exact_full, repaired: $b (rc $?)" | tee -a $O/first_run.txt
  h=$(timeout 50 python tests/tools/dropin_sweep.py one 2 2>>$O/err_$v.txt | tail -1)
  echo "$v  host call: $h" | tee -a $O/first_run.txt
done
tail -3 $O/err_tailhelp.txt
