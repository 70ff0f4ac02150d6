#!/bin/bash
# On the GPU box: HBM write and fetch counters of the headline frame's kernel for the libraries named
# (each put in place of raytracer_amd/librtx_hip.so in the box's scratch copy), one rocprofv3 pass each.
# Usage: tools/traffic_probe.sh out_dir lib.so [lib.so ...]
set -u
OUT=$1; shift
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for so in "$@"; do
  name=$(basename "$so" .so)
  cp "$so" raytracer_amd/librtx_hip.so
  for ctr in WRITE_SIZE FETCH_SIZE; do
    timeout -s KILL 120 rocprofv3 --pmc $ctr --output-format csv -d "$OUT/$name-$ctr" -o p -- python3 tools/perf_probe.py --only headline > "$OUT/$name-$ctr.log" 2>&1 || { echo "pass $name $ctr failed"; exit 1; }
  done
done
python3 - "$OUT" <<'PY'
import csv, glob, os, sys
for d in sorted(glob.glob(os.path.join(sys.argv[1], "*-*_SIZE"))):
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "renderKernel" in r["Kernel_Name"]]
        if v:
            print(os.path.basename(d), f"{sum(v) / len(v) / 1024:.1f} MB per launch over {len(v)} launches")
PY
