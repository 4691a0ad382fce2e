#!/bin/bash
# Round-end check on the GPU box: full GPU suite, smoke, default bench line, then rocprofv3 kernel stats of bench.py at 1M and 4M envs.
# usage: bash tools/final_run.sh <tag>
T=${1:-final}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
timeout -k 10 700 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/gpu_tests_$T.log 2>&1 || { tail -20 gpurun_out/gpu_tests_$T.log; exit 1; }
tail -2 gpurun_out/gpu_tests_$T.log
timeout -k 10 100 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke_$T.log 2>&1 || { tail gpurun_out/smoke_$T.log; exit 1; }
tail -1 gpurun_out/smoke_$T.log
timeout -k 10 200 python bench.py > gpurun_out/bench_$T.json 2> gpurun_out/bench_$T.err || exit 1
cd /tmp && export TMPDIR=/tmp
for n in 1048576 4194304; do
  timeout -s KILL 180 rocprofv3 --kernel-trace --stats --output-format csv -d "$REPO/gpurun_out/prof_${T}_$n" -- \
    python3 "$REPO/bench.py" --envs $n --steps 500 --warmup 50 --no-cpu-baseline > "$REPO/gpurun_out/bench_${T}_${n}_under_rocprof.json" 2> "$REPO/gpurun_out/prof_${T}_$n.log" || exit 1
done
cd "$REPO" && find gpurun_out/prof_${T}_* -name "*kernel_stats.csv" | while read f; do echo $f; head -4 $f | cut -c1-200; done
