#!/bin/bash
# EXPERIMENT: the vocabulary head's dW sweep on a side stream that OWNS some CUs (hipExtStreamCreateWithCUMask) in its
# foreground form, against the default background form that shares every CU; optionally the main stream on the complement.
B="python bench.py --steps 30 --warmup 8 --no_cpu_baseline --eval_steps 0 --full_steps 0"
pick() { python -c "import json,sys; d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1]); f=dict(d['roofline']['families']); print('%-28s'%sys.argv[2], 'ms/step %.3f'%d['ms_per_step'], 'median %.3f'%d['step_ms']['median'], ' '.join('%s %.3f'%(k, f[k]['ms_per_step']) for k in ('vocab_ce_fwd','gemm_nt','attn_bwd','gemm_tn','add_ln_bwd','gemm_nt_rows','attn_mq_bwd') if k in f))" $1 "$2"; }
for rep in 1 2; do
  $B > gpurun_out/cum_0_$rep.json 2>gpurun_out/cum_0_$rep.err; pick gpurun_out/cum_0_$rep.json "default"
  for n in 64 96 128; do
    B4C_BG_CU=$n timeout -k 10 200 $B > gpurun_out/cum_${n}_$rep.json 2>gpurun_out/cum_${n}_$rep.err && pick gpurun_out/cum_${n}_$rep.json "side owns $n CUs" || { echo "bg $n failed"; tail -3 gpurun_out/cum_${n}_$rep.err; }
    B4C_BG_CU=$n B4C_MAIN_CU=1 timeout -k 10 200 $B > gpurun_out/cumm_${n}_$rep.json 2>gpurun_out/cumm_${n}_$rep.err && pick gpurun_out/cumm_${n}_$rep.json "side $n, main the rest" || { echo "bg $n main-masked failed"; tail -3 gpurun_out/cumm_${n}_$rep.err; }
  done
done
