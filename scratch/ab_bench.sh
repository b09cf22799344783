#!/bin/bash
# same-box A/B of libb4c_hip.so variants under scratch/ab: ab_bench.sh v1 v2 ... (each: default bench, ms/step + families)
for v in "$@"; do
  cp scratch/ab/$v.so bert4clickpath_amd/libb4c_hip.so
  timeout -k 10 250 python bench.py --no_cpu_baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); f=d['roofline']['families']
print('%-6s %.3f' % ('$v', d['ms_per_step']), ' '.join('%s=%.2f' % (k[:9], v['ms_per_step']) for k, v in f.items() if v['ms_per_step'] > 0.1))"
done
