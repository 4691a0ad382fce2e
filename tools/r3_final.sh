#!/bin/bash
# Round-3 measurement batches on the GPU box (run from the repo root, one part per gpurun call):
#   tools/r3_final.sh a   GPU suite, smoke, default bench line, rocprofv3 kernel stats of bench.py at 1M / 4M envs  (tools/gpu.sh final)
#   tools/r3_final.sh b   PMC traffic passes of the headline kernel; rocprofv3 kernel stats + PMC passes of the mutable-grid kernels
#   tools/r3_final.sh c   kbench sweeps (every mutable-grid workload, small batches), fused rollout lines, PPO bench line
set -u
part=${1:-a}
case $part in
a)
  bash tools/gpu.sh final r3 ;;
b)
  bash tools/gpu.sh traffic gpurun_out/r3_traffic 1048576 > gpurun_out/r3_traffic_1M_pmc.txt 2>&1 || { tail -5 gpurun_out/r3_traffic_1M_pmc.txt; exit 1; }
  cat gpurun_out/r3_traffic_1M_pmc.txt
  for w in doorkey8x3 cluttered15x3; do
    export MG_WORKLOAD=$w MG_KERNEL=mg_dyn_kernel
    bash tools/gpu.sh prof gpurun_out/r3_prof_$w kbench 1048576 multigrid_amd/libmgrid.so > gpurun_out/r3_prof_$w.txt 2>&1 || { tail -5 gpurun_out/r3_prof_$w.txt; exit 1; }
    bash tools/gpu.sh pmc gpurun_out/r3_pmc_$w kbench 1048576 multigrid_amd/libmgrid.so > gpurun_out/r3_pmc_$w.log 2>&1 || { tail -5 gpurun_out/r3_pmc_$w.log; exit 1; }
    python3 tools/pmc_summary.py gpurun_out/r3_pmc_$w mg_dyn_kernel > gpurun_out/r3_pmc_$w.txt 2>&1 || exit 1
    cat gpurun_out/r3_prof_$w.txt | cut -c1-200; cat gpurun_out/r3_pmc_$w.txt
  done
  # the raw per-launch csv files are large; the summaries above are what is kept
  find gpurun_out/r3_pmc_* gpurun_out/r3_traffic gpurun_out/r3_prof_* -name "*.csv" -size +1M -delete 2>/dev/null
  ;;
c)
  bash tools/gpu.sh sweep -o gpurun_out/r3_dyn_final.txt -w "doorkey8x3 doorkey6x2 cluttered15x3 meetup8x3 meetup15x3" -e "MG_DEBUG_DEC=0 MG_DEBUG_DEC=1" -l multigrid_amd/libmgrid.so -s 100 -r 5 || exit 1
  bash tools/gpu.sh sweep -o gpurun_out/r3_small_final.txt -n "4096 65536 131072 1048576 4194304" -l multigrid_amd/libmgrid.so -s 200 -r 5 || exit 1
  : > gpurun_out/r3_roll_final.txt
  for a in "4096 64" "131072 32" "262144 32" "1048576 16"; do set -- $a
    timeout -k 10 120 python tools/roll_bench.py --envs $1 --k $2 --reps 10 >> gpurun_out/r3_roll_final.txt 2>/dev/null || exit 1
  done
  cut -c1-330 gpurun_out/r3_roll_final.txt
  timeout -k 10 400 python tools/bench_ppo.py --envs 4096 --steps 100 --iters 3 > gpurun_out/r3_bench_ppo_final.txt 2> gpurun_out/r3_bench_ppo_final.err || exit 1
  tail -1 gpurun_out/r3_bench_ppo_final.txt ;;
esac
