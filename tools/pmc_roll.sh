#!/bin/bash
# PMC passes (one rocprofv3 run each) over the fused rollout (tools/roll_bench.py) and the step kernel (bench.py), for comparing
# instructions per chunk-step:  tools/pmc_roll.sh <outdir> [envs] [K]
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; out=$1; n=${2:-1048576}; k=${3:-16}
mkdir -p "$REPO/$out"; cd /tmp && export TMPDIR=/tmp
while read -r name counters; do
  [ -z "$name" ] && continue
  timeout -s KILL 150 rocprofv3 --pmc $counters --output-format csv -d "$REPO/$out/roll_$name" -- python3 $REPO/tools/roll_bench.py --envs $n --k $k --reps 2 > "$REPO/$out/roll_$name.log" 2>&1 || { echo "pass roll_$name failed"; tail -5 "$REPO/$out/roll_$name.log"; exit 1; }
  timeout -s KILL 150 rocprofv3 --pmc $counters --output-format csv -d "$REPO/$out/step_$name" -- python3 $REPO/bench.py --envs $n --steps 40 --warmup 10 --no-cpu-baseline --also-envs 0 --rollout-envs 0 > "$REPO/$out/step_$name.log" 2>&1 || { echo "pass step_$name failed"; tail -5 "$REPO/$out/step_$name.log"; exit 1; }
  echo "pass $name ok"
done <<'P'
sq1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES
sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
P
cd "$REPO"
python3 - "$out" <<'PY'
import csv, glob, sys
from collections import defaultdict
out = sys.argv[1]
for f in sorted(glob.glob(out + '/*/*/*counter_collection.csv')):
  acc = defaultdict(lambda: defaultdict(list))
  for r in csv.DictReader(open(f)):
    kn = r['Kernel_Name']
    if 'mg_rollout_kernel' in kn or 'mg_kernel' in kn:
      acc[kn[:70]][r['Counter_Name']].append(float(r['Counter_Value']))
  for kn, d in acc.items():
    print(f.split('/')[-3], kn)
    for c, v in sorted(d.items()):
      v = v[len(v) // 4:]
      print('   %-26s n=%-4d avg=%.6g' % (c, len(v), sum(v) / len(v)))
PY
find "$out" -name "*.csv" -size +1M -delete 2>/dev/null
