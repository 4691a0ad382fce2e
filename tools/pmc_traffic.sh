#!/bin/bash
# HBM traffic passes only (FETCH_SIZE, WRITE_SIZE; one rocprofv3 run each) over tools/kbench.
# usage: [MG_DEBUG_NT=0|1] bash tools/pmc_traffic.sh <outdir> <n_envs>
set -u
OUT=$1; N=$2
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$REPO/$OUT"
cd /tmp && export TMPDIR=/tmp
run() {
  name=$1; shift
  timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d "$REPO/$OUT/$name" -- \
    "$REPO/tools/kbench" "$REPO/tests/golden/cluttered_fixed_15x15_seed52.bin" "$N" 20 1 "$REPO/multigrid_amd/libmgrid.so" > "$REPO/$OUT/$name.log" 2>&1 || { echo "pass $name failed"; tail -5 "$REPO/$OUT/$name.log"; return 1; }
  echo "pass $name ok"
}
run fetch FETCH_SIZE GRBM_GUI_ACTIVE &&
run write WRITE_SIZE TCC_HIT_sum TCC_MISS_sum &&
python3 "$REPO/tools/pmc_summary.py" "$REPO/$OUT" mg_kernel
