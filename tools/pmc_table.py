"""Average per dispatch of every counter of a rocprofv3 --pmc run, by kernel.  usage: python tools/pmc_table.py <dir>... """
import csv, glob, sys
from collections import defaultdict
agg = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for d in sys.argv[1:]:
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name'].replace('(anonymous namespace)::', '').split('(')[0][:48]
            a = agg[k][r['Counter_Name']]
            a[0] += float(r['Counter_Value']); a[1] += 1
for k, cs in agg.items():
    print(k)
    for c, (v, n) in sorted(cs.items()):
        print('   %-28s %16.0f  (x%d)' % (c, v / n, n))
