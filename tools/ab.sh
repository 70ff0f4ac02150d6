#!/bin/bash
# A/B on the GPU box: each library named on the command line is put in place of
# raytracer_amd/librtx_hip.so (the box's tree is scratch), then the per-rank scaling
# probe and the workload probe are run on it.
# Usage: tools/ab.sh "variation,variation" exp/a.so exp/b.so ...
set -e
only=$1; shift
cp raytracer_amd/librtx_hip.so /tmp/librtx_hip.orig.so
for so in "$@"; do
  echo "== $so"
  cp "$so" raytracer_amd/librtx_hip.so
  timeout -k 10 100 python -u tools/scale_probe.py --worlds 1,4,8 --steps 30 2>&1 | grep "^world [0-9]:"
  timeout -k 10 150 python -u tools/perf_probe.py --only "$only" --reps 5 2>&1 | grep -v amdgpu.ids
done
cp /tmp/librtx_hip.orig.so raytracer_amd/librtx_hip.so
