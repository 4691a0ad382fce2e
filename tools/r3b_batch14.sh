#!/bin/bash
# round 3, second half, batch 14: shared-map fused rollout at small batches — old plain geometry (MG_DEBUG_NW=16), new plain, agents over waves, auto
set -u
mkdir -p gpurun_out
: > gpurun_out/r3b_ssplit.jsonl
run() { timeout -k 10 150 python tools/roll_bench.py --envs $1 --k $2 --reps 5 > gpurun_out/r3b_ssplit_one.txt 2>> gpurun_out/r3b_ssplit.err || exit 1
        tail -1 gpurun_out/r3b_ssplit_one.txt | sed "s/^{/{\"mode\": \"$3\", /" >> gpurun_out/r3b_ssplit.jsonl; }
for nk in "4096 64" "16384 64" "32768 32" "43008 32" "49152 32" "65536 32" "81920 32" "98304 32" "114688 32" "131072 32" "262144 16"; do set -- $nk
  MG_DEBUG_ROLL_SPLIT=0 MG_DEBUG_NW=16 run $1 $2 plain_r3a
  MG_DEBUG_ROLL_SPLIT=0 run $1 $2 plain
  MG_DEBUG_ROLL_SPLIT=1 run $1 $2 split
  run $1 $2 auto
done
python - <<'PY'
import json
for l in open('gpurun_out/r3b_ssplit.jsonl'):
  d = json.loads(l)
  print('n=%-7d K=%-3d %-10s fused %.2f us/step  %.3g agent-steps/s  err %d' % (d['envs'], d['K'], d['mode'], d['us_per_step_fused'], d['agent_steps_per_s_fused'], d['errors']))
PY
