#!/bin/bash
# round 3, second half, batch 3: whole GPU suite on the final sources (fused mutable-grid rollout, digest fixtures), smoke, bench line,
# soak of the fused rollouts against the oracle on every registered id
set -u
mkdir -p gpurun_out
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/r3b_gpu_tests3.log 2>&1 || { tail -30 gpurun_out/r3b_gpu_tests3.log; exit 1; }
tail -2 gpurun_out/r3b_gpu_tests3.log
timeout -k 10 100 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r3b_smoke3.log 2>&1 || { tail gpurun_out/r3b_smoke3.log; exit 1; }
tail -1 gpurun_out/r3b_smoke3.log
timeout -k 10 300 python bench.py > gpurun_out/r3b_bench3.json 2> gpurun_out/r3b_bench3.err || { tail gpurun_out/r3b_bench3.err; exit 1; }
cut -c1-600 gpurun_out/r3b_bench3.json
timeout -k 10 600 python -u tests/soak_parity.py --envs 20000 --steps 400 --rollout 50 --seed 4242 > gpurun_out/r3b_soak_rollout.txt 2>&1 || { tail -20 gpurun_out/r3b_soak_rollout.txt; exit 1; }
tail -3 gpurun_out/r3b_soak_rollout.txt
