"""print the family table(s) of bench.py JSON line(s), side by side"""
import json, sys
ds = [json.load(open(p)) for p in sys.argv[1:]]
for p, d in zip(sys.argv[1:], ds):
    print('%s: %.3f ms/step (median %.3f p10 %.3f p90 %.3f) %.3f M items/s; roofline %s frac %.3f; eval %s; full %s; cpu %s' % (
        p, d['ms_per_step'], d['step_ms']['median'], d['step_ms']['p10'], d['step_ms']['p90'], d['value'] / 1e6, d['roofline']['family'],
        d['roofline']['frac'], ('%.2f ms' % d['eval']['ms_per_batch']) if 'eval' in d else '-',
        ('%.2f ms' % d['every_position_of_every_layer']['ms_per_step']) if 'every_position_of_every_layer' in d else '-',
        ('%.0f items/s' % d['cpu_baseline']['value']) if 'cpu_baseline' in d else '-'))
fams = sorted(set(f for d in ds for f in d['roofline']['families']), key=lambda f: -max(d['roofline']['families'].get(f, {'ms_per_step': 0})['ms_per_step'] for d in ds))
for f in fams:
    row = '%-18s' % f
    for d in ds:
        v = d['roofline']['families'].get(f)
        row += ' | %6.3f ms %5.1f l %6.0f GB/s %6.0f TF' % (v['ms_per_step'], v['launches_per_step'], v['GB_per_s'], v['TFLOP_per_s']) if v else ' | %38s' % '-'
    print(row)
for d in ds:
    b = d['roofline'].get('beside')
    if b:
        print('beside:', {k: round(v['ms_per_step'], 3) for k, v in b['families'].items()})
    if 'eval' in d:
        print('eval:', {k: round(v['ms_per_step'], 3) for k, v in d['eval']['roofline']['families'].items()})
