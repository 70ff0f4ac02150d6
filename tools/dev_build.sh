#!/bin/bash
# Development build of the HIP library for experiments: only the kernels of the 16-entry stack, no
# test counters (-DRTX_DEV_BUILD), an eighth of the full library's compile time.  Trees deeper than 16
# levels are refused by such a library.  Usage: tools/dev_build.sh out.so [extra hipcc flags]
set -e
cd "$(dirname "$0")/.."
out=${1:-exp/dev.so}; shift || true
mkdir -p "$(dirname "$out")"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -Wno-pass-failed -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize -Iinclude -Iraytracer_amd/csrc \
  -DRTX_DEV_BUILD "$@" -shared -o "$out" raytracer_amd/csrc/rtx_hip.hip
