"""Kernel sequence of ONE steady-state training step from a rocprofv3 --kernel-trace of bench.py: every launch in order
with its duration and the idle gap before it; library kernels are abbreviated, everything else (PyTorch / rocPRIM / runtime
fills and copies) is listed in full -- the glue that the step still contains."""
import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True))[-1]
rows = sorted(csv.DictReader(open(f)), key=lambda x: int(x['Start_Timestamp']))
# a step starts at the embedding forward kernel
starts = [i for i, x in enumerate(rows) if 'embed_fwd' in x['Kernel_Name'] or 'embed_concat_pe' in x['Kernel_Name']]
want = int(sys.argv[2]) if len(sys.argv) > 2 else len(starts) // 2
a, b = starts[want], starts[want + 1]
ours = ('gemm_', 'attn_', 'vce_', 'add_ln', 'embed_', 'adam', 'mask_', 'topk', 'softmax_', 'pack_', 'nonpad', 'remap', 'compact',
        'rows_kernel', 'tn_reduce', 'ln_bwd', 'padded_index', 'transpose', 'dropout', 'sampled', 'log_uniform', 'sort_', 'label_scale',
        'sum_scaled', 'relu_gate', 'rows_add', 'vce_apply', 'gather_i64', 'chain_ids')
glue_us = lib_us = gap_us = 0.0
prev_end = int(rows[a]['Start_Timestamp'])
n_glue = 0
for x in rows[a:b]:
    n = x['Kernel_Name']
    s, e = int(x['Start_Timestamp']), int(x['End_Timestamp'])
    d = (e - s) / 1e3
    gap = max(0.0, (s - prev_end) / 1e3)
    gap_us += gap
    prev_end = max(prev_end, e)
    lib = any(p in n for p in ours)
    if lib:
        lib_us += d
    else:
        glue_us += d
        n_glue += 1
    print('%s %8.1f us  gap %5.1f  %s' % (' ' if lib else '*', d, gap, n[:100] if not lib else n[:48]))
print('step: %d launches, library %.3f ms, glue %.3f ms in %d launches, gaps %.3f ms' % (b - a, lib_us / 1e3, glue_us / 1e3, n_glue, gap_us / 1e3))
