#!/bin/bash
# A/B of library builds on one box: "$@" = paths of libb4c_hip.so variants (the tree's own build is always first);
# per build the step time and the main families of a short bench run, two interleaved repetitions.
B="python bench.py --steps 30 --warmup 8 --no_cpu_baseline --eval_steps 0 --full_steps 0"
pick() { python -c "import json,sys; d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1]); f=dict(d['roofline']['families']); print('%-34s'%sys.argv[2], 'ms/step %.3f'%d['ms_per_step'], 'median %.3f'%d['step_ms']['median'], ' '.join('%s %.3f'%(k, f[k]['ms_per_step']) for k in ('vocab_ce_fwd','gemm_nt','attn_bwd','gemm_tn','add_ln_bwd','gemm_nt_ln','gemm_nt_rows','gemm_tn_rows','attn_mq_bwd') if k in f))" $1 "$2"; }
for rep in 1 2; do
  $B > gpurun_out/abl_0_$rep.json 2>/dev/null; pick gpurun_out/abl_0_$rep.json "tree"
  i=0
  for v in "$@"; do
    i=$((i+1))
    B4C_LIB_PATH=$v $B > gpurun_out/abl_${i}_$rep.json 2>/dev/null; pick gpurun_out/abl_${i}_$rep.json "$v"
  done
done
