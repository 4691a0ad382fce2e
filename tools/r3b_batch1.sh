#!/bin/bash
# round 3, second half, batch 1: fused mutable-grid rollout — parity tests, step-kernel A/B against the previous sources, timings
set -u
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests/test_gpu_rollout.py tests/test_gpu_dyn.py tests/test_gpu_scatter.py -m gpu -x -q --timeout 200 --timeout-method thread > gpurun_out/r3b_tests1.log 2>&1 || { tail -30 gpurun_out/r3b_tests1.log; exit 1; }
tail -2 gpurun_out/r3b_tests1.log
bash tools/gpu.sh sweep -o gpurun_out/r3b_step_ab.txt -w "doorkey8x3 cluttered15x3 meetup8x3" -l "tools/libmgrid_head.so tools/libmgrid_min.so" || exit 1
: > gpurun_out/r3b_dyn_roll.jsonl
for id in MultiGrid-DoorKey-8x8-v0 MultiGrid-Cluttered-v0 MultiGrid-Meetup-Random-8x8-v0; do
  for nk in "4096 64" "131072 32" "262144 16" "1048576 8"; do
    set -- $nk
    timeout -k 10 120 python tools/roll_bench_dyn.py --env-id $id --envs $1 --k $2 >> gpurun_out/r3b_dyn_roll.jsonl 2>> gpurun_out/r3b_dyn_roll.err || { tail -5 gpurun_out/r3b_dyn_roll.err; exit 1; }
  done
done
for lib in tools/libmgrid_min.so tools/libmgrid_min1.so; do
  for nk in "131072 32" "1048576 8"; do
    set -- $nk
    timeout -k 10 120 python tools/roll_bench_dyn.py --lib $lib --env-id MultiGrid-Meetup-Random-8x8-v0 --envs $1 --k $2 >> gpurun_out/r3b_dyn_roll.jsonl 2>> gpurun_out/r3b_dyn_roll.err || { tail -5 gpurun_out/r3b_dyn_roll.err; exit 1; }
  done
done
python - <<'PY'
import json
for l in open('gpurun_out/r3b_dyn_roll.jsonl'):
  d = json.loads(l)
  print(d['lib'][-12:], '%-34s n=%-8d K=%-3d fused %.2f us  single %.2f  r2form %.2f  frac(own) %.3f frac(stepB) %.3f single %.3f err %d' % (
    d['env_id'], d['envs'], d['K'], d['us_per_step_fused'], d['us_per_step_single_step_random'], d['us_per_step_round2_form'],
    d['frac_8TBs_fused_own_bytes'], d['frac_8TBs_fused_at_step_bytes'], d['frac_8TBs_single'], d['errors']))
PY
