# DEV TOOL (GPU box): kernel + copy timeline of the host-buffer path (one lhip_encode call of 1e5 stereo / mono frames, 4 repetitions); gpurun_out/r03tr/.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03tr; mkdir -p $O
for ch in 2 1; do
  timeout 200 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/tr$ch -- python $R/tests/tools/dropin_sweep.py one $ch > $O/run$ch.txt 2>&1
  k=$(find /tmp/tr$ch -name "*kernel_trace.csv" | head -1); m=$(find /tmp/tr$ch -name "*memory_copy_trace.csv" | head -1)
  python - "$k" "$m" > $O/timeline_ch$ch.txt <<'P'
import csv, sys
ev = []
for r in csv.DictReader(open(sys.argv[1])):
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K " + r["Kernel_Name"].split("(")[0][:40]))
for r in csv.DictReader(open(sys.argv[2])):
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "C " + r.get("Direction", r.get("Name", "?"))[:30] + " " + r.get("Bytes", "")))
ev.sort()
t0 = ev[0][0]
for s, e, n in ev:
    print(f"{(s - t0) / 1e6:10.3f} {(e - t0) / 1e6:10.3f} {(e - s) / 1e6:8.3f}  {n}")
P
done
tail -2 $O/run2.txt
