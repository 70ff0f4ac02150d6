#!/bin/bash
# Extra PMC passes (latency / occupancy / instruction mix) over a short bench run.
set -u
OUT=$1; shift
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
ARGS="--steps 5 --warmup 1 --no-cpu-baseline --no-every-block $*"
pass() {
  name=$1; shift
  timeout -s KILL 180 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -o "$name" -- python3 bench.py $ARGS > "$OUT/$name.log" 2>&1 || { echo "pass $name failed"; tail -3 "$OUT/$name.log"; exit 1; }
}
pass x1 SQ_LEVEL_WAVES SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_BRANCH SQ_IFETCH SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM
pass x2 SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES
echo done
