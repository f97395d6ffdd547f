R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kx -- python $R/bench.py --cpu-seconds 0 --no-extras --steps 2 --check-frames 0 --frames 50 > /tmp/kx.log 2>&1; echo "rocprofv3 bench 50 frames (no cooperative launch) rc=$?"
timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kx -- python $R/bench.py --cpu-seconds 0 --no-extras --steps 2 --check-frames 0 --frames 500 > /tmp/kx.log 2>&1; echo "rocprofv3 bench 500 frames (cooperative launch) rc=$?"
timeout 150 rocprofv3 --kernel-trace --output-format csv -d /tmp/kx -- python $R/bench.py --cpu-seconds 0 --no-extras --steps 2 --check-frames 0 --frames 500 > /tmp/kx.log 2>&1; echo "rocprofv3 (no --stats) 500 frames rc=$?"
