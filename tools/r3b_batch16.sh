#!/bin/bash
# round 3, second half, batch 16: kernel trace of the plain fused rollout at 61440 and 65536 envs (geometry and duration per dispatch)
set -u
mkdir -p gpurun_out
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export MG_DEBUG_ROLL_SPLIT=0
for n in 61440 65536; do
  rm -rf $R/gpurun_out/spot_$n
  timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/spot_$n -o t -- python3 $R/tools/roll_bench.py --envs $n --k 32 --reps 3 > $R/gpurun_out/spot_$n.log 2>&1 || { tail -5 $R/gpurun_out/spot_$n.log; exit 1; }
done
cd $R
python3 - <<'PY'
import csv, glob
for n in (61440, 65536):
  f = glob.glob('gpurun_out/spot_%d/**/*kernel_trace.csv' % n, recursive=True)[0]
  rows = [r for r in csv.DictReader(open(f)) if 'rollout' in r['Kernel_Name']]
  for r in rows[-3:]:
    print(n, r['Kernel_Name'][:60], 'grid', r['Grid_Size_X'], 'wg', r['Workgroup_Size_X'], 'lds', r.get('LDS_Block_Size'), 'vgpr', r.get('VGPR_Count'), 'us', (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
PY
