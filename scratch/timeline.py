"""Timeline of one steady-state training step from a rocprofv3 --kernel-trace of bench.py: start / end (us from the step's
first kernel) of every launch longer than a threshold, with its queue -- shows what runs BESIDE what (the vocabulary
head's background dW sweep against the encoder backward)."""
import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True))[-1]
rows = sorted(csv.DictReader(open(f)), key=lambda x: int(x['Start_Timestamp']))
starts = [i for i, x in enumerate(rows) if 'embed_fwd' in x['Kernel_Name']]
want = int(sys.argv[2]) if len(sys.argv) > 2 else len(starts) // 2
thr = float(sys.argv[3]) if len(sys.argv) > 3 else 30.0
a, b = starts[want], starts[want + 1]
t0 = int(rows[a]['Start_Timestamp'])
for x in rows[a:b]:
    s, e = (int(x['Start_Timestamp']) - t0) / 1e3, (int(x['End_Timestamp']) - t0) / 1e3
    if e - s < thr:
        continue
    print('q%-3s %9.1f -> %9.1f  (%7.1f us)  %s' % (x['Queue_Id'], s, e, e - s, x['Kernel_Name'][:60]))
