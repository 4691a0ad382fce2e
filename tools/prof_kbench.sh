#!/bin/bash
# rocprofv3 kernel-trace stats of tools/kbench (no counters).  usage: [MG_WORKLOAD=...] bash tools/prof_kbench.sh <outdir> <n_envs> <lib.so>
set -u
OUT=$1; N=$2; LIB=$3
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$REPO/$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 180 rocprofv3 --kernel-trace --stats --output-format csv -d "$REPO/$OUT" -- \
  "$REPO/tools/kbench" "$REPO/tests/golden/cluttered_fixed_15x15_seed52.bin" "$N" 100 3 "$REPO/$LIB" > "$REPO/$OUT/kbench.log" 2>&1
