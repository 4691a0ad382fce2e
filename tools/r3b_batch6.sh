#!/bin/bash
# round 3, second half, batch 6: emit_full with computed cell addresses and 4 / 6 vectors in flight — parity, then timing
set -u
mkdir -p gpurun_out
for id in Meetup-Random-8x8-v0 Meetup-Empty-15x15-v0 Meetup-Empty-6x6-v0; do
  timeout -k 10 200 python -u tests/soak_parity.py --lib tools/libmgrid_uf4.so --ids $id --envs 5000 --steps 200 --seed 99 2>&1 | grep -v amdgpu.ids || exit 1
  timeout -k 10 200 python -u tests/soak_parity.py --lib tools/libmgrid_uf4.so --ids $id --envs 5000 --steps 200 --seed 98 --rollout 40 2>&1 | grep -v amdgpu.ids || exit 1
done
bash tools/gpu.sh sweep -o gpurun_out/r3b_full_uf.txt -n "1048576" -w "meetup8x3 meetup15x3" -l "tools/libmgrid_min.so tools/libmgrid_uf4.so tools/libmgrid_uf6.so" -r 7 -s 100 || exit 1
