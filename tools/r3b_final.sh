#!/bin/bash
# Final measurement pass of round 3 on the GPU box (one part per gpurun call, from the repo root):
#   tools/r3b_final.sh a   GPU suite, smoke, default bench line, rocprofv3 kernel stats of bench.py at 1M / 4M envs  (tools/gpu.sh final)
#   tools/r3b_final.sh b   PMC traffic passes of the headline kernel; kbench lines of the mutable-grid kernels; fused-rollout lines
#   tools/r3b_final.sh c   soak: every id through the fused rollouts at 60 000 envs, and three ids at 1M envs
set -u
part=${1:-a}
case $part in
a)
  bash tools/gpu.sh final r3b ;;
b)
  bash tools/gpu.sh traffic gpurun_out/r3b_traffic 1048576 > gpurun_out/r3b_traffic_1M_pmc.txt 2>&1 || { tail -5 gpurun_out/r3b_traffic_1M_pmc.txt; exit 1; }
  cat gpurun_out/r3b_traffic_1M_pmc.txt
  find gpurun_out/r3b_traffic -name "*.csv" -size +1M -delete 2>/dev/null
  bash tools/gpu.sh sweep -o gpurun_out/r3b_dyn_final.txt -w "doorkey8x3 doorkey6x2 cluttered15x3 meetup8x3 meetup15x3" -l multigrid_amd/libmgrid.so -s 100 -r 5 || exit 1
  bash tools/gpu.sh sweep -o gpurun_out/r3b_small_final.txt -n "4096 65536 131072 1048576 4194304" -l "multigrid_amd/libmgrid.so tools/libmgrid_head.so" -s 200 -r 5 || exit 1
  : > gpurun_out/r3b_roll_final.txt
  for a in "4096 64" "131072 32" "262144 32" "1048576 16"; do set -- $a
    timeout -k 10 120 python tools/roll_bench.py --envs $1 --k $2 --reps 10 >> gpurun_out/r3b_roll_final.txt 2>/dev/null || exit 1
  done
  cut -c1-330 gpurun_out/r3b_roll_final.txt
  : > gpurun_out/r3b_dyn_roll_final.jsonl
  for id in MultiGrid-DoorKey-8x8-v0 MultiGrid-DoorKey-6x6-v0 MultiGrid-Cluttered-v0 MultiGrid-FourRooms-15x15-v0 MultiGrid-Meetup-Random-8x8-v0 MultiGrid-Meetup-Empty-15x15-v0; do
    for nk in "4096 64" "131072 32" "1048576 8"; do set -- $nk
      timeout -k 10 150 python tools/roll_bench_dyn.py --env-id $id --envs $1 --k $2 >> gpurun_out/r3b_dyn_roll_final.jsonl 2>> gpurun_out/r3b_dyn_roll_final.err || { tail -5 gpurun_out/r3b_dyn_roll_final.err; exit 1; }
    done
  done
  python - <<'PY'
import json
for l in open('gpurun_out/r3b_dyn_roll_final.jsonl'):
  d = json.loads(l)
  print('%-34s n=%-8d K=%-3d fused %.2f us  ring %.2f  single %.2f  r2form %.2f  %.3e agent-steps/s  err %d' % (
    d['env_id'], d['envs'], d['K'], d['us_per_step_fused'], d['us_per_step_fused_caller_actions'], d['us_per_step_single_step_random'],
    d['us_per_step_round2_form'], d['agent_steps_per_s_fused'], d['errors']))
PY
  ;;
c)
  timeout -k 10 900 python -u tests/soak_parity.py --envs 60000 --steps 1536 --rollout 64 --seed 27183 --max-steps 100 > gpurun_out/r3b_soak_rollout_60000.txt 2>&1 || { tail -5 gpurun_out/r3b_soak_rollout_60000.txt; exit 1; }
  tail -2 gpurun_out/r3b_soak_rollout_60000.txt
  : > gpurun_out/r3b_soak_rollout_1M.txt
  for spec in "Cluttered-Fixed-15x15 1048576 128 100" "DoorKey-8x8-v0 1000000 64 60" "Cluttered-v0 1000000 64 60" "Meetup-Random-8x8-v0 1000000 64 60"; do set -- $spec
    timeout -k 10 400 python -u tests/soak_parity.py --ids $1 --envs $2 --steps $3 --rollout 32 --seed 161 --max-steps $4 >> gpurun_out/r3b_soak_rollout_1M.txt 2>&1 || { tail -5 gpurun_out/r3b_soak_rollout_1M.txt; exit 1; }
  done
  cat gpurun_out/r3b_soak_rollout_1M.txt
  ;;
esac
