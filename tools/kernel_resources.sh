#!/bin/bash
# registers / scratch / LDS of every kernel in the HIP library (cross-compiles, no GPU needed)
set -e
cd "$(dirname "$0")/.."
out=${1:-/tmp/rtx_isa.s}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -Wno-pass-failed -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize -Iinclude -Iraytracer_amd/csrc \
  --cuda-device-only -S $RTX_DEFS -o "$out" raytracer_amd/csrc/rtx_hip.hip 2>/dev/null
python3 - "$out" <<'PY'
import re, sys
txt = open(sys.argv[1]).read()
for m in re.finditer(r"- \.agpr_count:.*?\.wavefront_size", txt, re.S):
    blk = m.group(0)
    g = lambda k: re.search(r"\." + k + r":\s+(\S+)", blk).group(1)
    name = g("name")
    name = re.sub(r"^_ZN12_GLOBAL__N_1\d+", "", name)
    print(f"{name:48s} vgpr {g('vgpr_count'):>4s} sgpr {g('sgpr_count'):>4s} scratch {g('private_segment_fixed_size'):>5s} lds {g('group_segment_fixed_size'):>6s}")
PY
