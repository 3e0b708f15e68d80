"""Per-step timeline from a rocprofv3 --kernel-trace CSV: busy time and idle gaps of the main queue."""
import csv, glob, sys, collections
d = sys.argv[1]
f = glob.glob(d + '/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
qkey = 'Queue_Id' if 'Queue_Id' in rows[0] else 'Stream_Id'
# steps are delimited by the Adam kernel
adam = [i for i, r in enumerate(rows) if 'k_adam_flat' in r['Kernel_Name']]
if len(adam) < 6:
    print('not enough steps'); sys.exit(0)
a, b = adam[-4], adam[-3]          # one steady-state step: (a, b]
step = rows[a + 1:b + 1]
t0, t1 = int(rows[a]['End_Timestamp']), int(step[-1]['End_Timestamp'])
print('step wall %.1f us, %d kernels' % ((t1 - t0) / 1e3, len(step)))
byq = collections.defaultdict(list)
for r in step:
    byq[r[qkey]].append(r)
for q, rs in byq.items():
    busy = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in rs)
    print('queue %s: %d kernels, busy %.1f us' % (q, len(rs), busy / 1e3))
mainq = max(byq, key=lambda q: len(byq[q]))
rs = byq[mainq]
gaps = []
prev_end = t0
for r in rs:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    gaps.append((s - prev_end, r['Kernel_Name'][:60]))
    prev_end = max(prev_end, e)
tot_gap = sum(g for g, _ in gaps if g > 0)
print('main queue: idle between kernels %.1f us in total' % (tot_gap / 1e3))
hist = collections.Counter()
for g, _ in gaps:
    hist[min(int(max(g, 0) / 1e3), 20)] += 1
print('gap histogram (us -> count):', dict(sorted(hist.items())))
print('largest gaps:')
for g, n in sorted(gaps, reverse=True)[:12]:
    print('  %7.1f us before %s' % (g / 1e3, n))
