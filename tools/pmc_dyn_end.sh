#!/bin/bash
# PMC passes (one rocprofv3 run each) of the mutable-grid kernels on the round-3 end sources: the general step instantiation against
# FREE (tools/kbench, DoorKey-8x8 x 3 and random Cluttered-15x15 x 3 at 1M envs) and the fused rollout kernel (tools/roll_bench_dyn.py,
# DoorKey-8x8 at 131072 envs x K 32).   tools/pmc_dyn_end.sh <outdir>
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; out=$1
mkdir -p "$REPO/$out"
for w in doorkey8x3 cluttered15x3; do
  for mode in general free; do
    export MG_WORKLOAD=$w
    if [ $mode = general ]; then export MG_DEBUG_DYN_NO_FREE=1; else unset MG_DEBUG_DYN_NO_FREE; fi
    bash tools/gpu.sh pmc $out/${w}_$mode kbench 1048576 multigrid_amd/libmgrid.so > $out/${w}_$mode.log 2>&1 || { tail -5 $out/${w}_$mode.log; exit 1; }
    echo "== $w $mode (tools/kbench, 1048576 envs; per launch)"; python3 tools/pmc_summary.py $out/${w}_$mode mg_dyn_kernel
  done
done
unset MG_DEBUG_DYN_NO_FREE MG_WORKLOAD
cd /tmp && export TMPDIR=/tmp
while read -r name counters; do
  [ -z "$name" ] && continue
  timeout -s KILL 150 rocprofv3 --pmc $counters --output-format csv -d "$REPO/$out/roll_$name" -- python3 $REPO/tools/roll_bench_dyn.py --env-id MultiGrid-DoorKey-8x8-v0 --envs 131072 --k 32 --reps 2 > "$REPO/$out/roll_$name.log" 2>&1 || { echo "pass roll_$name failed"; tail -5 "$REPO/$out/roll_$name.log"; exit 1; }
done <<'P'
sq1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES
sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
P
cd "$REPO"
echo "== fused rollout, DoorKey-8x8 x 3, 131072 envs x K 32 (tools/roll_bench_dyn.py; per launch of 32 steps x 2048 chunks)"
python3 tools/pmc_summary.py $out/roll mg_dyn_rollout_kernel 2>/dev/null || python3 - "$out" <<'PY'
import csv, glob, sys
from collections import defaultdict
out = sys.argv[1]
for f in sorted(glob.glob(out + '/roll_*/*/*counter_collection.csv')):
  acc = defaultdict(list)
  for r in csv.DictReader(open(f)):
    if 'mg_dyn_rollout_kernel' in r['Kernel_Name']:
      acc[r['Counter_Name']].append(float(r['Counter_Value']))
  for c, v in sorted(acc.items()):
    print('%-34s n=%-4d avg=%.6g' % (c, len(v), sum(v) / len(v)))
PY
find "$out" -name "*.csv" -size +1M -delete 2>/dev/null
