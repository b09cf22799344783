#!/bin/bash
# A/B of the step in row parts (model.cloze_step, bench.py --row_parts 2) against the whole-batch step, interleaved.
set -e
B="python bench.py --steps 30 --warmup 8 --no_cpu_baseline --eval_steps 0 --full_steps 0"
pick() { python -c "import json,sys; d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1]); print('%-40s'%sys.argv[2], 'ms/step %.3f'%d['ms_per_step'], 'median %.3f'%d['step_ms']['median'], 'loss %.4f'%d['final_loss'])" $1 "$2"; }
for rep in 1 2; do
  $B --row_parts 1 > gpurun_out/ab_rp1_$rep.json 2>/dev/null; pick gpurun_out/ab_rp1_$rep.json "whole batch"
  $B --row_parts 2 > gpurun_out/ab_rp2_$rep.json 2>/dev/null; pick gpurun_out/ab_rp2_$rep.json "2 row parts"
  i=0
  for v in "$@"; do
    i=$((i+1))
    env $v $B --row_parts 2 > gpurun_out/ab_rpv${i}_$rep.json 2>/dev/null; pick gpurun_out/ab_rpv${i}_$rep.json "2 row parts, $v"
  done
done
