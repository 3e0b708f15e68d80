"""Kernel-by-kernel list of one steady-state step of a rocprofv3 --kernel-trace run, steps delimited by any kernel name:
python tools/step_list_by.py <dir> <delimiter substring> [which-step-from-the-end]
-> start (us), duration, gap to the previous kernel, grid / workgroup size, LDS bytes, name; then totals per kernel name."""
import csv, glob, sys, collections
d, delim = sys.argv[1], sys.argv[2]
back = int(sys.argv[3]) if len(sys.argv) > 3 else 2
f = glob.glob(d + '/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
marks = [i for i, r in enumerate(rows) if delim in r['Kernel_Name']]
a, b = marks[-back - 1], marks[-back]
step = rows[a:b]
t0 = int(step[0]['Start_Timestamp'])
print('%d kernels, %.1f us from the first start to the last end' % (len(step), (max(int(r['End_Timestamp']) for r in step) - t0) / 1e3))
tot = collections.defaultdict(lambda: [0, 0.0])
last = t0
for r in step:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    name = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0][:60]
    g = [int(r.get('Grid_Size_' + k, r.get('Grid_Size', 0)) or 0) for k in 'XYZ'] if 'Grid_Size_X' in r else [int(r.get('Grid_Size', 0) or 0)]
    w = int(r.get('Workgroup_Size_X', r.get('Workgroup_Size', 0)) or 0)
    print('%9.1f %8.2f gap %6.2f  grid %-18s wg %4d lds %6s  %s' % ((s - t0) / 1e3, (e - s) / 1e3, (s - last) / 1e3, 'x'.join(str(v) for v in g), w, r.get('LDS_Block_Size', r.get('LDS_Block_Size_v', '?')), name))
    last = e
    tot[name][0] += 1; tot[name][1] += (e - s) / 1e3
print('--- per kernel: launches, total us')
for n, (c, t) in sorted(tot.items(), key=lambda kv: -kv[1][1]):
    print('%4d %9.1f  %s' % (c, t, n))
