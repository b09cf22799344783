"""Summarise one rocprofv3 --pmc pass per kernel name: mean of every counter per dispatch.
    python scratch/pmc_kernel.py <rocprof out dir> [name filter]"""
import collections, csv, glob, sys
d = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ''
f = sorted(glob.glob(d + '/**/*counter_collection.csv', recursive=True))[-1]
per = collections.defaultdict(lambda: collections.defaultdict(float))
kern = {}
for x in csv.DictReader(open(f)):
    per[x['Dispatch_Id']][x['Counter_Name']] += float(x['Counter_Value'])
    kern[x['Dispatch_Id']] = x['Kernel_Name']
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for disp, c in per.items():
    n = kern[disp]
    if flt and flt not in n:
        continue
    for k, v in c.items():
        agg[n[:70]][k].append(v)
for n, cs in agg.items():
    print(n, 'dispatches', len(next(iter(cs.values()))))
    for k, v in sorted(cs.items()):
        print('   %-32s %16.0f' % (k, sum(v) / len(v)))
