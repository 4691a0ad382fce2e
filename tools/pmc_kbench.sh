#!/bin/bash
# PMC passes over tools/kbench (one rocprofv3 run per pass; block slot limits: 8 SQ, 4 TCC).
# usage (on the GPU box, from the repo root): [MG_WORKLOAD=doorkey8x3] bash tools/pmc_kbench.sh <outdir> <n_envs> <lib.so>
set -u
OUT=$1; N=$2; LIB=$3
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$REPO/$OUT"
cd /tmp && export TMPDIR=/tmp
run() {
  name=$1; shift
  timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d "$REPO/$OUT/$name" -- \
    "$REPO/tools/kbench" "$REPO/tests/golden/cluttered_fixed_15x15_seed52.bin" "$N" 20 1 "$REPO/$LIB" > "$REPO/$OUT/$name.log" 2>&1 || { echo "pass $name failed"; tail -5 "$REPO/$OUT/$name.log"; return 1; }
  echo "pass $name ok"
}
run sq1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES &&
run sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE &&
run fetch FETCH_SIZE GRBM_GUI_ACTIVE &&
run write WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
