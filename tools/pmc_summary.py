"""Summarise tools/pmc_passes.sh output: per-launch averages of each counter for the step kernel."""
import csv
import glob
import sys
from collections import defaultdict

out = sys.argv[1]
kern = sys.argv[2] if len(sys.argv) > 2 else 'mg_kernel'
for f in sorted(glob.glob(out + '/*/*/*counter_collection.csv')):
  acc = defaultdict(list)
  for r in csv.DictReader(open(f)):
    if kern in r['Kernel_Name']:
      acc[r['Counter_Name']].append(float(r['Counter_Value']))
  for k, v in sorted(acc.items()):
    v = v[len(v) // 4:]   # skip the reset launch and warm-up
    print('%-34s n=%-4d avg=%.6g' % (k, len(v), sum(v) / len(v)))
