"""Family table of one or more bench.py JSON lines, side by side: python scratch/famtab2.py a.json b.json ..."""
import json
import sys

docs = []
for f in sys.argv[1:]:
    txt = [l for l in open(f) if l.startswith('{')]
    docs.append(json.loads(txt[-1]))
print('%-18s' % 'file', *['%12s' % f.split('/')[-1][:12] for f in sys.argv[1:]])
print('%-18s' % 'ms/step', *['%12.3f' % d['ms_per_step'] for d in docs])
print('%-18s' % 'median', *['%12.3f' % d['step_ms']['median'] for d in docs])
print('%-18s' % 'M items/s', *['%12.3f' % (d['value'] / 1e6) for d in docs])
fams = []
for d in docs:
    for k in d['roofline']['families']:
        if k not in fams:
            fams.append(k)
fams.sort(key=lambda k: -docs[0]['roofline']['families'].get(k, {'ms_per_step': 0})['ms_per_step'])
for k in fams:
    print('%-18s' % k, *['%12.3f' % d['roofline']['families'].get(k, {'ms_per_step': float('nan')})['ms_per_step'] for d in docs])
print('%-18s' % 'sum', *['%12.3f' % sum(v['ms_per_step'] for v in d['roofline']['families'].values()) for d in docs])
for d in docs:
    b = d['roofline'].get('beside')
    if b:
        print('beside:', {k: round(v['ms_per_step'], 3) for k, v in b['families'].items()})
    if 'eval' in d:
        e = d['eval']
        print('eval %.3f ms/batch, fused_topk %.3f' % (e['ms_per_batch'], e.get('fused_topk', {}).get('ms_per_batch', float('nan'))))
    if 'every_position_of_every_layer' in d:
        print('every position: %.3f ms' % d['every_position_of_every_layer']['ms_per_step'])
