#!/bin/bash
# round 3, second half, batch 9: the shared-map rollout with the agents dealt out over waves — parity, then timing against the plain kernel
set -u
mkdir -p gpurun_out
for n in 64 1000 4096 20000; do
  timeout -k 10 200 python -u tests/soak_parity.py --lib tools/libmgrid_split.so --ids Cluttered-Fixed-15x15 --envs $n --steps 330 --seed $n --rollout 33 2>&1 | grep -v amdgpu.ids
done
: > gpurun_out/r3b_split.jsonl
for nk in "4096 64" "16384 64" "32768 32" "43008 32" "65536 32" "131072 32"; do set -- $nk
  for sp in 0 1; do
    MG_DEBUG_ROLL_SPLIT=$sp timeout -k 10 120 python tools/roll_bench.py --lib tools/libmgrid_split.so --envs $1 --k $2 --reps 10 >> gpurun_out/r3b_split.jsonl 2>/dev/null || exit 1
  done
done
python - <<'PY'
import json
for l in open('gpurun_out/r3b_split.jsonl'):
  d = json.loads(l)
  print('n=%-7d K=%-3d split=%s  fused %.2f us/step  %.3e agent-steps/s  err %d' % (d['envs'], d['K'], d['split'], d['us_per_step_fused'], d['agent_steps_per_s_fused'], d['errors']))
PY
