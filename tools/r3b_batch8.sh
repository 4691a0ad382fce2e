#!/bin/bash
# round 3, second half, batch 8: FREE instantiation of the mutable-grid step kernel — parity, then A/B against the general one
set -u
mkdir -p gpurun_out
for id in DoorKey-8x8-v0 Cluttered-v0 Meetup-Random-8x8-v0 FourRooms-15x15-v0; do
  timeout -k 10 200 python -u tests/soak_parity.py --lib tools/libmgrid_free.so --ids MultiGrid-$id --envs 5000 --steps 300 --seed 55 --no-env-reward 2>&1 | grep -v amdgpu.ids
done
bash tools/gpu.sh sweep -o gpurun_out/r3b_free_ab.txt -n "1048576 131072" -w "doorkey8x3 cluttered15x3 meetup8x3 meetup15x3" -e "MG_DEBUG_DYN_NO_FREE=1 -" -l "tools/libmgrid_free.so" -r 7 -s 100 || exit 1
