#!/bin/bash
# On the GPU box: VALU / SALU instruction counts of the frame kernel for probe variations (one rocprofv3 pass each).
# Usage: tools/valu_by_variant.sh out_dir "variation" ["variation" ...]
set -u
OUT=$1; shift
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
i=0
for var in "$@"; do
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d "$OUT/v$i" -o p -- python3 tools/perf_probe.py --only "$var" --reps 3 > "$OUT/v$i.log" 2>&1 || { echo "pass $var failed"; exit 1; }
  python3 - "$OUT/v$i" "$var" <<'PY'
import csv, glob, os, sys, collections
acc = collections.defaultdict(list)
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "renderKernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print(sys.argv[2], {k: round(sum(v) / len(v) / 1e6, 1) for k, v in sorted(acc.items())}, "M per launch")
PY
done
