#!/bin/bash
# round 3, second half, batch 18: shared-map split rollout with two waves per chunk (agents {0,2} / {1}) — tests, soak, timing at 81920..131072 envs
set -u
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests/test_gpu_rollout.py -x -q -m gpu --timeout 120 --timeout-method thread > gpurun_out/r3b_b18_pytest.txt 2>&1 || { tail -30 gpurun_out/r3b_b18_pytest.txt; exit 1; }
tail -2 gpurun_out/r3b_b18_pytest.txt
for n in 100000 131072; do
  timeout -k 10 200 python -u tests/soak_parity.py --ids MultiGrid-Cluttered-Fixed-15x15 --envs $n --steps 256 --seed $n --rollout 32 --max-steps 40 2>&1 | grep -v "amdgpu.ids\|^soak: 0"
done
MG_DEBUG_ROLL_SPLIT=2 MG_DEBUG_GRID=9 timeout -k 10 200 python -u tests/soak_parity.py --ids MultiGrid-Cluttered-Fixed-15x15 MultiGrid-Empty-16x16-v0 MultiGrid-Empty-Random-5x5-v0 --envs 9000 --steps 192 --seed 3 --rollout 32 --max-steps 40 2>&1 | grep -v "amdgpu.ids\|^soak: 0"
: > gpurun_out/r3b_ssplit2.jsonl
run() { timeout -k 10 150 python tools/roll_bench.py --envs $1 --k $2 --reps 5 > gpurun_out/r3b_ssplit_one.txt 2>> gpurun_out/r3b_ssplit.err || exit 1
        tail -1 gpurun_out/r3b_ssplit_one.txt | sed "s/^{/{\"mode\": \"$3\", /" >> gpurun_out/r3b_ssplit2.jsonl; }
for nk in "4096 64" "65536 32" "81920 32" "98304 32" "114688 32" "131072 32" "147456 32"; do set -- $nk
  MG_DEBUG_ROLL_SPLIT=0 run $1 $2 plain
  MG_DEBUG_ROLL_SPLIT=2 run $1 $2 split2
  MG_DEBUG_ROLL_SPLIT=3 run $1 $2 split3
  run $1 $2 auto
done
python - <<'PY'
import json
for l in open('gpurun_out/r3b_ssplit2.jsonl'):
  d = json.loads(l)
  print('n=%-7d K=%-3d %-7s fused %.2f us/step  %.3g agent-steps/s  err %d' % (d['envs'], d['K'], d['mode'], d['us_per_step_fused'], d['agent_steps_per_s_fused'], d['errors']))
PY
