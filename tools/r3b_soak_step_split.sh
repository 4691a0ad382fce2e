#!/bin/bash
# every registered id step by step at batch sizes whose plain steps run the agents-over-waves kernels (4 000 and 20 000 envs), with and
# without the env_reward output (the latter is the free-running form that takes those kernels) — final sources
set -u
mkdir -p gpurun_out
timeout -k 10 900 python -u tests/soak_parity.py --envs 4000 --steps 800 --seed 1234 --max-steps 50 --no-env-reward > gpurun_out/r3b_soak_stepsplit_4000.txt 2>&1 || { tail -5 gpurun_out/r3b_soak_stepsplit_4000.txt; exit 1; }
tail -1 gpurun_out/r3b_soak_stepsplit_4000.txt
timeout -k 10 900 python -u tests/soak_parity.py --envs 20000 --steps 400 --seed 1235 --max-steps 50 --no-env-reward > gpurun_out/r3b_soak_stepsplit_20000.txt 2>&1 || { tail -5 gpurun_out/r3b_soak_stepsplit_20000.txt; exit 1; }
tail -1 gpurun_out/r3b_soak_stepsplit_20000.txt
