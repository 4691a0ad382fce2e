#!/bin/bash
# PMC passes for the step kernel, one rocprofv3 run per pass (block slot limits: 8 SQ, 4 TCC).
# usage (on the GPU box, from the repo root): bash tools/pmc_passes.sh <outdir> [bench args...]
set -u
OUT=$1; shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$REPO/$OUT"
cd /tmp && export TMPDIR=/tmp
run() {
  name=$1; shift
  timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d "$REPO/$OUT/$name" -- \
    python3 "$REPO/bench.py" --steps 40 --warmup 10 --no-cpu-baseline $BENCH_ARGS > "$REPO/$OUT/$name.log" 2>&1 || { echo "pass $name failed"; tail -5 "$REPO/$OUT/$name.log"; return 1; }
  echo "pass $name ok"
}
BENCH_ARGS="$*"
run sq1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES &&
run sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE &&
run sq3 SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INST_CYCLES_VMEM_WR SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS &&
run fetch FETCH_SIZE GRBM_GUI_ACTIVE &&
run write WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
