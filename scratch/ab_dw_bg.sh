#!/bin/bash
# A/B of the vocabulary head's dW sweep as a background kernel beside the encoder backward (the default,
# B4C_OVERLAP_DW=1) against the foreground order (B4C_OVERLAP_DW=0), interleaved on one box; "$@" = extra variants
# as VAR=value strings (e.g. B4C_VCE_DW_BG=192 B4C_VCE_DW_WEIGHTS=1.4,1,1).
set -e
B="python bench.py --steps 30 --warmup 8 --no_cpu_baseline --eval_steps 0 --full_steps 0"
pick() { python -c "import json,sys; d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1]); f=dict(d['roofline']['families']); f.update(d['roofline'].get('beside', {}).get('families', {})); print('%-44s'%sys.argv[2], 'ms/step %.3f'%d['ms_per_step'], 'median %.3f'%d['step_ms']['median'], 'vce_dw %.3f'%f.get('vocab_ce_dw_bg', f.get('vocab_ce_dw'))['ms_per_step'], 'gemm_nt %.3f'%f['gemm_nt']['ms_per_step'], 'attn_bwd %.3f'%f['attn_bwd']['ms_per_step'], 'add_ln_bwd %.3f'%f['add_ln_bwd']['ms_per_step'], 'gemm_tn %.3f'%f['gemm_tn']['ms_per_step'])" $1 "$2"; }
for rep in 1 2; do
  B4C_OVERLAP_DW=0 $B > gpurun_out/ab_fg_$rep.json 2>/dev/null; pick gpurun_out/ab_fg_$rep.json "foreground"
  $B > gpurun_out/ab_bg_$rep.json 2>/dev/null; pick gpurun_out/ab_bg_$rep.json "background (default)"
  i=0
  for v in "$@"; do
    i=$((i+1))
    env $v $B > gpurun_out/ab_v${i}_$rep.json 2>/dev/null; pick gpurun_out/ab_v${i}_$rep.json "$v"
  done
done
