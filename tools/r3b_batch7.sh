#!/bin/bash
# round 3, second half, batch 7: rollout-invariant switches folded at compile time in the fused mutable-grid rollout kernel
set -u
mkdir -p gpurun_out
: > gpurun_out/r3b_fold.jsonl
for lib in tools/libmgrid_min.so tools/libmgrid_fold.so tools/libmgrid_foldstd.so; do
  for spec in "MultiGrid-DoorKey-8x8-v0 131072 32" "MultiGrid-DoorKey-8x8-v0 1048576 8" "MultiGrid-Cluttered-v0 131072 32" "MultiGrid-Meetup-Random-8x8-v0 131072 32"; do set -- $spec
    timeout -k 10 150 python tools/roll_bench_dyn.py --lib $lib --env-id $1 --envs $2 --k $3 >> gpurun_out/r3b_fold.jsonl 2>> gpurun_out/r3b_fold.err || { tail -5 gpurun_out/r3b_fold.err; exit 1; }
  done
done
python - <<'PY'
import json
for l in open('gpurun_out/r3b_fold.jsonl'):
  d = json.loads(l)
  print('%-22s %-32s n=%-8d K=%-3d fused %.2f us  ring %.2f  err %d' % (d['lib'][-22:], d['env_id'], d['envs'], d['K'], d['us_per_step_fused'], d['us_per_step_fused_caller_actions'], d['errors']))
PY
