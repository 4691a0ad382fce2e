#!/bin/bash
# round-3 measurement batch: (1) north-star mapping bound at config-2 sizes (ABL=8: one agent's view per wave), (2) fused rollout
# after the one-Philox-block change, (3) every mutable-grid workload with the table / arithmetic decode
set -u
O=gpurun_out/r3_batch1.txt; : > $O
echo "## small batches: product build vs -DMG_ABL=8 (timing-only: one agent's view per wave)" >> $O
bash tools/gpu.sh sweep -o gpurun_out/r3_small.txt -n "4096 65536 131072" -l "tools/libmgrid_min.so tools/libmgrid_abl8.so" -s 400 >> $O 2>&1 || exit 1
echo "## fused rollout (one Philox block per step)" >> $O
for a in "4096 64" "131072 32" "262144 32" "1048576 16"; do set -- $a
  timeout -k 10 120 python tools/roll_bench.py --envs $1 --k $2 --reps 10 >> $O 2>&1 || exit 1
done
echo "## mutable-grid workloads: MG_DEBUG_DEC=0 (table decode) / 1 (arithmetic decode)" >> $O
bash tools/gpu.sh sweep -o gpurun_out/r3_dyn.txt -w "doorkey8x3 doorkey6x2 cluttered15x3 meetup8x3 meetup15x3" -e "MG_DEBUG_DEC=0 MG_DEBUG_DEC=1" -l multigrid_amd/libmgrid.so >> $O 2>&1 || exit 1
cat $O | cut -c1-200
