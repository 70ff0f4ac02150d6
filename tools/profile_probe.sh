#!/bin/bash
# PMC counters of the frame kernel for one probe variation (one rocprofv3 run per pass).
# Usage: tools/profile_probe.sh <outdir> "<variation>"
set -u
OUT=$1; VAR=$2
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
pass() {
  name=$1; shift
  timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -o "$name" -- python3 tools/perf_probe.py --only "$VAR" --reps 2 > "$OUT/$name.log" 2>&1 || { echo "pass $name failed"; tail -3 "$OUT/$name.log"; exit 1; }
}
pass sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY
pass sq2 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VMEM
pass x1 SQ_LEVEL_WAVES SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_BRANCH SQ_IFETCH SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM
pass x2 SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES
echo done
