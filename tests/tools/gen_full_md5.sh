#!/bin/bash
# TEST TOOL (needs /root/reference + node): regenerates tests/golden/full_md5.json from the unmodified reference.
# ~80 CPU-minutes in total; jobs run in parallel (-P, default 6).
set -e
cd "$(dirname "$0")/../.."
P=${1:-6}
OUT=tests/golden/full_md5.json
TMP=$(mktemp -d)
{
  for s in $(seq 12345 12352); do echo "sine 2 128 100000 $s 1"; done          # SURVEY 8d config 3 (BASELINE configs[2]), rank r = seed 12345 + r
  for s in $(seq 12345 12352); do echo "sine 2 320 100000 $s 1"; done          # config 4 (configs[3])
  for s in $(seq 12345 12352); do echo "sine 1 128 100000 $s 1"; done          # config 2 (configs[1])
  echo "bursts 2 128 100000 777 1"; echo "bursts 1 128 100000 777 1"           # second material (SURVEY 8d config 2 note)
  for b in $(seq 0 31); do echo "sine 1 128 1000 $((1000 + 32 * b)) 32"; done  # config 5 (configs[4]): 1024 streams, seed 1000 + s
  for s in $(seq 12345 12352); do echo "centre_sine 2 128 100000 $s 1 joint"; done   # joint-stereo extension: every frame mid/side
  echo "bursts 2 128 100000 777 1 joint"                                        # joint-stereo extension: mid/side and left/right frames mixed
  for b in $(seq 0 31); do echo "sine 1 128 1000 $((1000 + 16 * b)) 16 reservoir"; done  # bit-reservoir extension: 512 mono streams x 1000 frames (config 5 shape: 128, 256 and 512 streams on one GPU)
} > $TMP/jobs
nl -ba $TMP/jobs | xargs -P $P -L 1 sh -c 'node tests/tools/gen_full_md5.js $1 $2 $3 $4 $5 $6 $7 $8 > '$TMP'/out.$0'
python3 - "$TMP" "$OUT" <<'PY'
import glob, json, sys
rows = []
for f in sorted(glob.glob(sys.argv[1] + "/out.*")):
    rows += [json.loads(l) for l in open(f) if l.strip()]
rows.sort(key=lambda r: (r["corpus"], r["channels"], r["kbps"], r["frames"], r["seed"], r.get("joint", 0), r.get("reservoir", 0)))
json.dump({"generator": "tests/tools/gen_full_md5.sh (unmodified reference under node, encodeBuffer of the whole stream, no flush)", "entries": rows},
          open(sys.argv[2], "w"), indent=0)
print("wrote", len(rows), "entries")
PY
rm -rf $TMP
