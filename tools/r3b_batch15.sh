#!/bin/bash
# round 3, second half, batch 15: the 34816..61440-env slow spot of the shared-map fused rollout — does the penalty scale with K (per step) or not (per launch)?
set -u
mkdir -p gpurun_out
: > gpurun_out/r3b_spot.jsonl
run() { timeout -k 10 150 python tools/roll_bench.py --envs $1 --k $2 --reps 5 > gpurun_out/r3b_ssplit_one.txt 2>> gpurun_out/r3b_ssplit.err || exit 1
        tail -1 gpurun_out/r3b_ssplit_one.txt | sed "s/^{/{\"mode\": \"$3\", /" >> gpurun_out/r3b_spot.jsonl; }
for n in 32768 43008 65536; do
  for k in 8 16 32 64 128; do
    MG_DEBUG_ROLL_SPLIT=0 run $n $k plain
    MG_DEBUG_ROLL_SPLIT=1 run $n $k split
  done
done
python - <<'PY'
import json
for l in open('gpurun_out/r3b_spot.jsonl'):
  d = json.loads(l)
  print('n=%-7d chunks=%-5d K=%-3d %-6s fused %.2f us/step  %.1f us/launch  traj %.0f MB' % (d['envs'], d['envs'] // 64, d['K'], d['mode'], d['us_per_step_fused'], d['us_per_step_fused'] * d['K'], d['envs'] * d['K'] * 257.5 / 1e6))
PY
