cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r03sw
timeout 300 python -m pytest tests/test_gpu_parity.py -q -k "host_call_in_overlapped_chunks or long_random" 2>&1 | tail -2 | tee gpurun_out/r03sw/test.txt
timeout 500 python tests/tools/dropin_sweep.py 2>&1 | tee gpurun_out/r03sw/dropin_sweep.txt
