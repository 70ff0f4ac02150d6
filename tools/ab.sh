#!/bin/bash
# A/B timing on the GPU box: every exp/*.so is a build of the HIP library; each is
# put in place of raytracer_amd/librtx_hip.so (the box's copy of the tree is
# scratch) and the workload probe is run on it.
# Usage: tools/ab.sh "variation,variation" [reps]
set -e
only=${1:-headline,looking away,close-up d=1.2}
reps=${2:-5}
cp raytracer_amd/librtx_hip.so /tmp/librtx_hip.orig.so
for so in exp/*.so; do
  echo "== $so"
  cp "$so" raytracer_amd/librtx_hip.so
  timeout -k 10 120 python -u tools/perf_probe.py --only "$only" --reps "$reps" 2>&1 | grep -v amdgpu.ids
done
cp /tmp/librtx_hip.orig.so raytracer_amd/librtx_hip.so
