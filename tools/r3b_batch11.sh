#!/bin/bash
# round 3, second half, batch 11: product build with both split rollouts in one-workgroup form — rollout tests, soaks, timing
set -u
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests/test_gpu_rollout.py -x -q -m gpu --timeout 120 --timeout-method thread > gpurun_out/r3b_b11_pytest.txt 2>&1 || { tail -30 gpurun_out/r3b_b11_pytest.txt; exit 1; }
tail -2 gpurun_out/r3b_b11_pytest.txt
soak() { timeout -k 10 200 python -u tests/soak_parity.py "$@" 2>&1 | grep -v "amdgpu.ids\|^soak: 0"; }
for id in MultiGrid-Cluttered-Fixed-15x15 MultiGrid-DoorKey-16x16-v0 MultiGrid-FourRooms-v0 MultiGrid-Meetup-v0 MultiGrid-DoorKey-6x6-v0; do
  soak --ids $id --envs 4000 --steps 320 --seed 4000 --rollout 40 --max-steps 30
  MG_DEBUG_ROLL_SPLIT=1 MG_DEBUG_GRID=7 soak --ids $id --envs 9000 --steps 60 --seed 8 --rollout 1 --max-steps 30
done
: > gpurun_out/r3b_dsplit.jsonl
for id in MultiGrid-DoorKey-8x8-v0 MultiGrid-Cluttered-v0 MultiGrid-Meetup-Random-8x8-v0 MultiGrid-DoorKey-6x6-v0 MultiGrid-DoorKey-16x16-v0; do
  for nk in "4096 64" "16384 32" "32768 32" "65536 32"; do set -- $nk
    for sp in 0 1 auto; do
      if [ $sp = auto ]; then unset MG_DEBUG_ROLL_SPLIT; else export MG_DEBUG_ROLL_SPLIT=$sp; fi
      timeout -k 10 150 python tools/roll_bench_dyn.py --env-id $id --envs $1 --k $2 --reps 5 | sed "s/^{/{\"split\": \"$sp\", /" >> gpurun_out/r3b_dsplit.jsonl 2>> gpurun_out/r3b_dsplit.err || exit 1
    done
  done
done
unset MG_DEBUG_ROLL_SPLIT
python - <<'PY'
import json
for l in open('gpurun_out/r3b_dsplit.jsonl'):
  d = json.loads(l)
  print('%-32s n=%-7d K=%-3d split=%-4s  fused %.2f us/step  ring %.2f  err %d' % (d['env_id'], d['envs'], d['K'], d['split'], d['us_per_step_fused'], d['us_per_step_fused_caller_actions'], d['errors']))
PY
