#!/bin/bash
# round 3, second half, batch 23: rocprofv3 kernel stats of the fused rollouts at config 2's and config 3's sizes (which kernel runs, its duration per launch)
set -u
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
for nk in "4096 64" "131072 32"; do set -- $nk
  rm -rf $R/gpurun_out/prof_roll_$1
  timeout -s KILL 240 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_roll_$1 -- python3 $R/tools/roll_bench.py --envs $1 --k $2 --reps 20 > $R/gpurun_out/prof_roll_$1.log 2>&1 || { tail -5 $R/gpurun_out/prof_roll_$1.log; exit 1; }
  f=$(find $R/gpurun_out/prof_roll_$1 -name "*kernel_stats.csv" | head -1)
  echo "== $1 envs, K = $2: $f"; head -4 "$f" | cut -c1-200
  find $R/gpurun_out/prof_roll_$1 -name "*kernel_trace.csv" -delete
done
