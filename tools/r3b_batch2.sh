#!/bin/bash
# round 3, second half, batch 2: branch-free shaper — A/B timing against the previous sources, then the whole GPU suite
set -u
mkdir -p gpurun_out
bash tools/gpu.sh sweep -o gpurun_out/r3b_shaper_ab.txt -n "1048576" -w "static doorkey8x3 cluttered15x3 meetup8x3" -l "tools/libmgrid_min.so tools/libmgrid_shp.so" -r 7 || exit 1
bash tools/gpu.sh sweep -o gpurun_out/r3b_shaper_ab_small.txt -n "4096 131072" -w "static doorkey8x3" -l "tools/libmgrid_min.so tools/libmgrid_shp.so" -r 7 || exit 1
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/r3b_gpu_tests2.log 2>&1 || { tail -30 gpurun_out/r3b_gpu_tests2.log; exit 1; }
tail -2 gpurun_out/r3b_gpu_tests2.log
timeout -k 10 100 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r3b_smoke2.log 2>&1 || { tail gpurun_out/r3b_smoke2.log; exit 1; }
tail -1 gpurun_out/r3b_smoke2.log
for nk in "131072 32" "1048576 16"; do set -- $nk; timeout -k 10 120 python tools/roll_bench.py --envs $1 --k $2 || exit 1; done
