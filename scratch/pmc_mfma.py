"""Turns one rocprofv3 PMC pass (SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16, with --kernel-trace)
of bench.py into profiles/r02_mfma_util.json: per launch family, the fraction of the chip's matrix-pipe cycles that were
busy -- MfmaUtil of rocprofv3's derived_counters: sum(SQ_VALU_MFMA_BUSY_CYCLES) / (GUI_ACTIVE x 1024 SIMDs), with
GRBM_GUI_ACTIVE reported as the sum over the 8 XCDs (MI355X_MICROARCH.md, DVFS give-back) -- and the MFMA FLOP count
(MOPS x 512).
    python scratch/pmc_mfma.py <rocprof out dir> profiles/r02_mfma_util.json"""
import collections
import csv
import glob
import json
import sys

FAM = [('gemm_nt_ln', ('gemm_nt_ln',)), ('vocab_proj', ('gemm_nt_wide',)), ('gemm_nt', ('gemm_nt_kernel',)), ('gemm_tn', ('gemm_tn_', 'tn_reduce')),
       ('attn_mq_bwd', ('attn_mq_bwd',)), ('attn_mq_fwd', ('attn_mq_fwd',)), ('attn_bwd', ('attn_bwd',)), ('attn_fwd', ('attn_fwd',)), ('vocab_rank', ('vce_scan_kernel<128, 0,', 'vce_label_logit')), ('vocab_topk', ('vce_scan_kernel', 'vce_tau_kernel', 'vce_select_kernel')), ('vocab_lse', ('vce_token_kernel<128, 0', 'vce_lse_kernel')), ('vocab_ce_fwd', ('vce_token_kernel', 'vce_combine')),
       ('vocab_ce_dw_bg', ('vce_dw_kernel<128, 1>', 'vce_dw_kernel<64, 1>')), ('vocab_ce_dw', ('vce_dw_kernel', 'vce_label')), ('add_ln_bwd', ('add_ln_bwd',)), ('add_ln_fwd', ('add_ln_fwd',)),
       ('embed_fwd', ('embed_fwd',)), ('embed_bwd', ('embed_bwd',)), ('adam', ('adam_kernel',)), ('softmax_rows', ('softmax_rows',)),
       ('topk_rows', ('topk_rows',))]


def fam_of(name):
    for f, pats in FAM:
        if any(p in name for p in pats):
            return f
    return None


def main():
    d, out = sys.argv[1], sys.argv[2]
    f = sorted(glob.glob(d + '/**/*counter_collection.csv', recursive=True))[-1]
    per = collections.defaultdict(lambda: collections.defaultdict(float))     # dispatch -> counter -> value
    kern = {}
    for x in csv.DictReader(open(f)):
        per[x['Dispatch_Id']][x['Counter_Name']] += float(x['Counter_Value'])
        kern[x['Dispatch_Id']] = x['Kernel_Name']
    agg = collections.defaultdict(lambda: [0.0, 0.0, 0.0, 0])
    for disp, c in per.items():
        fam = fam_of(kern[disp])
        if fam is None:
            continue
        a = agg[fam]
        a[0] += c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0)
        a[1] += c.get('GRBM_GUI_ACTIVE', 0.0)
        a[2] += c.get('SQ_INSTS_VALU_MFMA_MOPS_BF16', 0.0)
        a[3] += 1
    res = {}
    for fam, (busy, gui, mops, n) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        cyc = gui / 8.0                                   # per-XCD active cycles of the family's launches
        res[fam] = {'mfma_busy_frac': busy / (cyc * 1024.0) if cyc else 0.0, 'mfma_flop_per_launch': mops * 512.0 / max(n, 1),
                    'gpu_cycles_per_launch': cyc / max(n, 1), 'launches_sampled': n,
                    'method': 'rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16 (one pass, --kernel-trace); '
                              'busy / ((GUI_ACTIVE / 8 XCDs) x 1024 SIMDs)'}
        print('%-14s MFMA pipe busy %5.1f %%   %8.1f GFLOP / launch   n = %d' % (fam, 100 * res[fam]['mfma_busy_frac'], res[fam]['mfma_flop_per_launch'] / 1e9, n))
    json.dump(res, open(out, 'w'), indent=1)


if __name__ == '__main__':
    main()
