#!/bin/bash
# same-box A/B of an environment switch: ab_env.sh VAR v1 v2 ... (default bench per value, ms/step + families)
var=$1; shift
for v in "$@"; do
  env $var=$v timeout -k 10 250 python bench.py --no_cpu_baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); f=d['roofline']['families']
print('%-6s %.3f' % ('$v', d['ms_per_step']), ' '.join('%s=%.2f' % (k[:9], v['ms_per_step']) for k, v in f.items() if v['ms_per_step'] > 0.1))"
done
