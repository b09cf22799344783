#!/bin/bash
# same-box, interleaved A/B of an environment switch: ab_env2.sh VAR v1 v2 [reps]  (short bench per value: ms/step, median, embed_bwd)
var=$1; a=$2; b=$3; reps=${4:-3}
B="python bench.py --steps 30 --warmup 8 --no_cpu_baseline --eval_steps 0 --full_steps 0"
for rep in $(seq $reps); do
  for v in $a $b; do
    env $var=$v timeout -k 10 250 $B 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); f=d['roofline']['families']
print('$var=%-4s ms/step %.3f median %.3f' % ('$v', d['ms_per_step'], d['step_ms']['median']), ' '.join('%s=%.3f' % (k, f[k]['ms_per_step']) for k in ('embed_bwd','embed_fwd','gemm_nt','attn_fwd','vocab_ce_fwd') if k in f))"
  done
done
