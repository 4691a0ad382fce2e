#!/bin/bash
# One parameterised script for the GPU-box measurements (replaces the round-1 one-off scripts bal / nt_* / nw_big / shape2 /
# small / sweep_bench / pmc_* / prof_kbench / final_run / build_variant).  Run from the repo root.
#
#   tools/gpu.sh build <name> [hipcc flags]           tools/libmgrid_<name>.so  (e.g. build abl1 -DMG_ABL=1)   [CPU]
#   tools/gpu.sh sweep [-o file] [-n "sizes"] [-e "VAR=val VAR=val ..."] [-w "workloads"] [-l "libs"] [-s steps] [-r rounds]
#        tools/kbench over every combination; each -e entry is ONE environment setting (comma-separated pairs for several
#        variables at once, "-" for none); workloads: static doorkey8x3 meetup15x3 cluttered15x3 ...
#   tools/gpu.sh bench-sweep [-o file] [-n "sizes"] [bench.py args]      bench.py over batch sizes, one JSON line each
#   tools/gpu.sh prof <outdir> kbench <n> <lib> | bench [bench.py args]  rocprofv3 --kernel-trace --stats
#   tools/gpu.sh pmc  <outdir> kbench <n> <lib> | bench [bench.py args]  PMC passes, one rocprofv3 run per pass
#   tools/gpu.sh traffic <outdir> <n> [lib]                               FETCH_SIZE / WRITE_SIZE passes + summary
#   tools/gpu.sh final <tag>                                              GPU suite, smoke, bench line, rocprof stats at 1M / 4M
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
MAP=tests/golden/cluttered_fixed_15x15_seed52.bin
cmd=${1:-help}; shift || true

kbench_target() {   # <n> <lib> -> the command line rocprofv3 runs
  echo "$REPO/tools/kbench $REPO/$MAP $1 20 1 $REPO/$2"
}

rocprof_passes() {  # <outdir> <target command...> ; passes from stdin, one per line: name counters...
  local out=$1; shift
  mkdir -p "$REPO/$out"
  cd /tmp && export TMPDIR=/tmp
  while read -r name counters; do
    [ -z "$name" ] && continue
    timeout -s KILL 150 rocprofv3 --pmc $counters --output-format csv -d "$REPO/$out/$name" -- "$@" > "$REPO/$out/$name.log" 2>&1 \
      || { echo "pass $name failed"; tail -5 "$REPO/$out/$name.log"; cd "$REPO"; return 1; }
    echo "pass $name ok"
  done
  cd "$REPO"
}

case $cmd in
build)
  name=$1; shift
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -Iinclude -Imultigrid_amd/csrc \
    "$@" -o tools/libmgrid_$name.so multigrid_amd/csrc/mgrid.hip multigrid_amd/csrc/mgrid_dynroll.hip ;;
sweep)
  out=gpurun_out/sweep.txt; sizes=1048576; envs="-"; works=static; libs=multigrid_amd/libmgrid.so; steps=200; rounds=5
  while getopts "o:n:e:w:l:s:r:" o; do case $o in o) out=$OPTARG;; n) sizes=$OPTARG;; e) envs=$OPTARG;; w) works=$OPTARG;; l) libs=$OPTARG;;
    s) steps=$OPTARG;; r) rounds=$OPTARG;; *) exit 2;; esac; done
  : > "$out"
  for w in $works; do for n in $sizes; do for e in $envs; do
    echo "workload=$w n=$n env=$e" >> "$out"
    ( [ "$e" != "-" ] && export ${e//,/ }; [ "$w" != static ] && export MG_WORKLOAD=$w
      timeout -k 10 100 tools/kbench $MAP $n $steps $rounds $libs ) >> "$out" 2>&1 || exit 1
  done; done; done
  grep -v "^$" "$out" | cut -c1-170 ;;
bench-sweep)
  out=gpurun_out/bench_sweep.jsonl; sizes="4096 65536 262144 1048576 1310720 2097152 4194304"
  while getopts "o:n:" o; do case $o in o) out=$OPTARG;; n) sizes=$OPTARG;; *) exit 2;; esac; done
  shift $((OPTIND - 1)); : > "$out"
  for n in $sizes; do
    timeout -k 10 150 python bench.py --envs $n --steps 1000 --warmup 100 --no-cpu-baseline --also-envs 0 --rollout-envs 0 --config4-envs 0 "$@" >> "$out" 2>> gpurun_out/sweep.err || exit 1
  done
  python - "$out" <<'PY'
import json, sys
for l in open(sys.argv[1]):
  d = json.loads(l)
  print(d['config']['envs_per_gpu'], '%.2f us' % (d['roofline']['kernel_ms'] * 1e3), '%.3e' % d['value'], '%.3f' % d['roofline']['frac'], d['roofline']['launch']['path'])
PY
  ;;
prof)
  out=$1; what=$2; shift 2
  if [ "$what" = kbench ]; then target="$REPO/tools/kbench $REPO/$MAP $1 100 3 $REPO/$2"
  else target="python3 $REPO/bench.py --steps 500 --warmup 50 --no-cpu-baseline --also-envs 0 --rollout-envs 0 --config4-envs 0 $*"; fi
  mkdir -p "$REPO/$out"; cd /tmp && export TMPDIR=/tmp
  timeout -s KILL 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$REPO/$out" -- $target > "$REPO/$out/run.log" 2>&1 || { tail -5 "$REPO/$out/run.log"; exit 1; }
  cd "$REPO" && find "$out" -name "*kernel_stats.csv" | while read -r f; do echo "$f"; head -5 "$f" | cut -c1-220; done ;;
pmc)
  out=$1; what=$2; shift 2
  if [ "$what" = kbench ]; then target=$(kbench_target "$1" "$2")
  else target="python3 $REPO/bench.py --steps 40 --warmup 10 --no-cpu-baseline --also-envs 0 --rollout-envs 0 --config4-envs 0 $*"; fi
  rocprof_passes "$out" $target <<'P'
sq1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES
sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
sq3 SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INST_CYCLES_VMEM_WR SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS
fetch FETCH_SIZE GRBM_GUI_ACTIVE
write WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
P
  ;;
traffic)
  out=$1; n=$2; lib=${3:-multigrid_amd/libmgrid.so}
  rocprof_passes "$out" $(kbench_target "$n" "$lib") <<'P' && python3 tools/pmc_summary.py "$out" ${MG_KERNEL:-mg_kernel}
fetch FETCH_SIZE GRBM_GUI_ACTIVE
write WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
P
  ;;
final)
  T=${1:-final}
  timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/gpu_tests_$T.log 2>&1 || { tail -20 gpurun_out/gpu_tests_$T.log; exit 1; }
  tail -2 gpurun_out/gpu_tests_$T.log
  timeout -k 10 100 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke_$T.log 2>&1 || { tail gpurun_out/smoke_$T.log; exit 1; }
  tail -1 gpurun_out/smoke_$T.log
  timeout -k 10 300 python bench.py > gpurun_out/bench_$T.json 2> gpurun_out/bench_$T.err || exit 1
  for n in 1048576 4194304; do
    bash tools/gpu.sh prof gpurun_out/prof_${T}_$n bench --envs $n > gpurun_out/prof_${T}_$n.txt 2>&1 || exit 1
  done
  cat gpurun_out/prof_${T}_*.txt ;;
*)
  sed -n 2,16p "$0" ;;
esac
