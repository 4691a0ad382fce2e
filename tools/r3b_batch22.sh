#!/bin/bash
# round 3, second half, batch 22: mutable-grid fused rollouts on the final launch geometry (regression check against r3_dyn_rollout_agent_split_lines.jsonl / r3b_final b)
set -u
mkdir -p gpurun_out
: > gpurun_out/r3b_dyn_roll_geom.jsonl
for id in MultiGrid-DoorKey-8x8-v0 MultiGrid-Cluttered-v0 MultiGrid-Meetup-Random-8x8-v0; do
  for nk in "4096 64" "49152 32" "65536 32" "100000 32" "131072 32" "1048576 8"; do set -- $nk
    timeout -k 10 150 python tools/roll_bench_dyn.py --env-id $id --envs $1 --k $2 --reps 5 >> gpurun_out/r3b_dyn_roll_geom.jsonl 2>> gpurun_out/r3b_dyn_roll_geom.err || exit 1
  done
done
python - <<'PY'
import json
for l in open('gpurun_out/r3b_dyn_roll_geom.jsonl'):
  d = json.loads(l)
  print('%-32s n=%-8d K=%-3d fused %.2f us/step  ring %.2f  single %.2f  err %d' % (d['env_id'], d['envs'], d['K'], d['us_per_step_fused'], d['us_per_step_fused_caller_actions'], d['us_per_step_single_step_random'], d['errors']))
PY
