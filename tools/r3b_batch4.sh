#!/bin/bash
# round 3, second half, batch 4: product library against the session-start sources (shared-map kernel, interleaved rounds);
# fused mutable-grid rollout with a caller's action ring; rocprofv3 kernel stats of the fused mutable-grid rollout
set -u
mkdir -p gpurun_out
bash tools/gpu.sh sweep -o gpurun_out/r3b_product_vs_head.txt -n "1048576 131072" -w "static" -l "tools/libmgrid_head.so multigrid_amd/libmgrid.so" -r 9 || exit 1
: > gpurun_out/r3b_dyn_roll_ring.jsonl
for id in MultiGrid-DoorKey-8x8-v0 MultiGrid-Cluttered-v0; do
  timeout -k 10 120 python tools/roll_bench_dyn.py --env-id $id --envs 131072 --k 32 >> gpurun_out/r3b_dyn_roll_ring.jsonl 2>> gpurun_out/r3b_dyn_roll_ring.err || { tail -5 gpurun_out/r3b_dyn_roll_ring.err; exit 1; }
done
python - <<'PY'
import json
for l in open('gpurun_out/r3b_dyn_roll_ring.jsonl'):
  d = json.loads(l)
  print('%-30s n=%d K=%d fused %.2f us  fused+ring %.2f  single %.2f' % (d['env_id'], d['envs'], d['K'], d['us_per_step_fused'], d['us_per_step_fused_caller_actions'], d['us_per_step_single_step_random']))
PY
mkdir -p gpurun_out/prof_dynroll; R=$(pwd); cd /tmp && export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_dynroll -- python3 $R/tools/roll_bench_dyn.py --env-id MultiGrid-DoorKey-8x8-v0 --envs 131072 --k 32 > $R/gpurun_out/prof_dynroll/run.log 2>&1 || { tail -5 $R/gpurun_out/prof_dynroll/run.log; exit 1; }
cd $R; find gpurun_out/prof_dynroll -name "*kernel_stats.csv" | while read -r f; do echo "$f"; head -6 "$f" | cut -c1-200; done
