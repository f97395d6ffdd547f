# DEV TOOL (GPU box): where does Mp3Encoder.encodeBuffer under Node lose time against the same call through the C ABI from C / Python?
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_diag_node; mkdir -p $O
cd $R
python - <<'PY'
import sys, numpy as np
sys.path.insert(0, 'tests')
import pcm
L, R = pcm.sine(1152 * 100000, 2, seed=12345)
np.stack([L, R], axis=1).astype('<i2').tofile('/tmp/s.pcm')
PY
gcc -O2 -o /tmp/abi_cli tests/tools/abi_cli.c -Llamejs_amd/lib -llamejs_hip -Wl,-rpath,$R/lamejs_amd/lib
echo "== C client, malloc'd buffers, one call (twice)" | tee $O/diag.txt
for i in 1 2; do LAMEJS_HIP_TRACE_CHUNKS=1 /tmp/abi_cli lamejs_amd/tables/t_2_44100_128.bin /tmp/s.pcm /tmp/o.mp3 2 44100 128 2>&1 | tee -a $O/diag.txt; done
echo "== node, 2 reps, chunk trace" | tee -a $O/diag.txt
LAMEJS_HIP_TRACE_CHUNKS=1 node tests/tools/bench_dropin.js sine 2 128 100000 12345 2 2>&1 | cut -c1-900 | tee -a $O/diag.txt
echo "== node, chunks off (one batch)" | tee -a $O/diag.txt
LAMEJS_HIP_NO_HOST_CHUNKS=1 node tests/tools/bench_dropin.js sine 2 128 100000 12345 2 2>&1 | cut -c1-600 | tee -a $O/diag.txt
echo "== python ctypes, chunk trace" | tee -a $O/diag.txt
LAMEJS_HIP_TRACE_CHUNKS=1 python - 2>&1 <<'PY' | tee -a $O/diag.txt
import sys, time, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import lamejs_amd, pcm
L, R = pcm.sine(1152 * 100000, 2, seed=12345)
lib = lamejs_amd.load_library()
for rep in range(3):
    enc = lamejs_amd.Mp3Encoder(2, 44100, 128)
    cap = lib.lhip_encode_output_bytes(enc._h, len(L)); out = np.empty(cap, dtype=np.uint8)
    t = time.perf_counter(); n = lib.lhip_encode(enc._h, L.ctypes.data, R.ctypes.data, len(L), out.ctypes.data, cap); print('python call ms', 1000 * (time.perf_counter() - t), n)
    enc.close()
PY
