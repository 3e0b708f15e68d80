"""Group a rocprofv3 kernel_trace.csv by (kernel name, grid, workgroup size, LDS): calls, total and average duration.
usage: python tools/trace_by_grid.py <dir with *_kernel_trace.csv> <out.csv> [steps]"""
import csv, glob, sys
from collections import defaultdict
src, dst = sys.argv[1], sys.argv[2]
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 1
f = glob.glob(src + '/**/*kernel_trace.csv', recursive=True)[0]
agg = defaultdict(lambda: [0, 0])
for r in csv.DictReader(open(f)):
    key = (r['Kernel_Name'].replace('(anonymous namespace)::', '').split('(')[0][:60], r.get('Grid_Size_X', ''), r.get('Grid_Size_Y', ''), r.get('Grid_Size_Z', ''),
           r.get('Workgroup_Size_X', ''), r.get('LDS_Block_Size', ''))
    a = agg[key]
    a[0] += 1; a[1] += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
with open(dst, 'w') as o:
    o.write('name,grid_x,grid_y,grid_z,wg,lds,calls_per_step,total_ms_per_step,avg_us\n')
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        o.write('"%s",%s,%s,%s,%s,%s,%.1f,%.3f,%.2f\n' % (k + (a[0] / steps, a[1] / 1e6 / steps, a[1] / 1e3 / a[0])))
