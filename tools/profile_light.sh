# Counter passes over the area-light + glossy workload (walk, light and resolve kernels); summarise with tools/summarize_counters.py gpurun_out/prof_light <kernel name>
set -u
OUT=gpurun_out/prof_light; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
ARGS="--workload area_glossy --steps 3 --warmup 1 --no-cpu-baseline --no-every-block --frames-in-flight 1"
pass() { name=$1; shift; timeout -s KILL 240 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -o "$name" -- python3 bench.py $ARGS > "$OUT/$name.log" 2>&1 || { echo "pass $name failed"; exit 1; }; }
pass sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY
pass sq2 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VMEM
pass tcp TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum
echo done
