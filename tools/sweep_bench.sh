#!/bin/bash
# bench.py over batch sizes (one JSON line each) -> gpurun_out/<name>.jsonl
O=gpurun_out/${1:-bench_sweep}.jsonl; : > $O
for n in 4096 65536 262144 1048576 1310720 2097152 4194304; do
  timeout -k 10 150 python bench.py --envs $n --steps 1000 --warmup 100 --no-cpu-baseline >> $O 2>> gpurun_out/sweep.err || exit 1
done
python - "$O" <<'PY'
import json, sys
for l in open(sys.argv[1]):
  d = json.loads(l)
  print(d['config']['envs_per_gpu'], '%.2f us' % (d['roofline']['kernel_ms'] * 1e3), '%.3e' % d['value'], '%.3f' % d['roofline']['frac'], d['roofline']['launch']['path'])
PY
