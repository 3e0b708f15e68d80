"""One steady-state training step of a rocprofv3 --kernel-trace run, kernel by kernel in start order:
python tools/one_step_list.py <dir> [which-step-from-the-end]  ->  queue, start (us from the step's first kernel), duration,
gap to the previous kernel of the same queue, name.  Steps are delimited by k_adam_flat."""
import csv, glob, sys, collections
d = sys.argv[1]
back = int(sys.argv[2]) if len(sys.argv) > 2 else 4
f = glob.glob(d + '/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
adam = [i for i, r in enumerate(rows) if 'k_adam_flat' in r['Kernel_Name']]
a, b = adam[-back - 1], adam[-back]
step = rows[a + 1:b + 1]
t0 = int(step[0]['Start_Timestamp'])
last_end = {}
qn = {}
print('%d kernels, %.1f us from the first start to the last end' % (len(step), (max(int(r['End_Timestamp']) for r in step) - t0) / 1e3))
cnt = collections.Counter()
for r in step:
    q = r.get('Queue_Id', '?')
    qn.setdefault(q, len(qn))
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    gap = (s - last_end[q]) / 1e3 if q in last_end else 0.0
    last_end[q] = e
    name = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0][:64]
    cnt[name] += 1
    print('q%d %9.1f %7.2f  gap %6.2f  %s' % (qn[q], (s - t0) / 1e3, (e - s) / 1e3, gap, name))
print('--- launches per step by kernel')
for n, c in cnt.most_common():
    print('%4d  %s' % (c, n))
