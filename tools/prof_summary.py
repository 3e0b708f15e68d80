"""Condense a rocprofv3 --kernel-trace --stats run (kernel_stats.csv) into a small tracked summary."""
import csv, glob, sys
src = sys.argv[1]; dst = sys.argv[2]; steps = int(sys.argv[3]) if len(sys.argv) > 3 else 1
f = glob.glob(src + '/**/*kernel_stats.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
with open(dst, 'w') as o:
    o.write('# rocprofv3 --kernel-trace --stats summary (%s), %d profiled steps\n' % (f.split('/')[-1], steps))
    o.write('# total kernel time %.3f ms = %.3f ms/step\n' % (tot / 1e6, tot / 1e6 / steps))
    o.write('name,calls,calls_per_step,total_ms,avg_us,min_us,max_us,pct\n')
    for r in rows:
        o.write('"%s",%s,%.1f,%.3f,%.2f,%.2f,%.2f,%.2f\n' % (r['Name'][:110], r['Calls'], int(r['Calls']) / steps,
                float(r['TotalDurationNs']) / 1e6, float(r['AverageNs']) / 1e3, float(r['MinNs']) / 1e3,
                float(r['MaxNs']) / 1e3, float(r['Percentage'])))
print(open(dst).read()[:6000])
