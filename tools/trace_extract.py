"""Condense a rocprofv3 kernel_trace.csv to (name, queue, start, end) rows and print the per-step timeline statistics:
busy time per queue, idle gaps on the main queue, overlap of the two queues.

usage: python tools/trace_extract.py <kernel_trace.csv> <out.csv> [steps]"""
import csv
import sys
from collections import defaultdict


def main():
    src, dst = sys.argv[1], sys.argv[2]
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    rows = list(csv.DictReader(open(src)))
    ev = []
    for r in rows:
        name = r['Kernel_Name'].split('(')[0].replace(',', ';')[:70]
        ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r.get('Queue_Id', ''), name))
    ev.sort()
    with open(dst, 'w') as f:
        f.write('name,queue,start_ns,dur_ns\n')
        t0 = ev[0][0]
        for s, e, q, n in ev:
            f.write('%s,%s,%d,%d\n' % (n, q, s - t0, e - s))
    byq = defaultdict(list)
    for s, e, q, n in ev:
        byq[q].append((s, e, n))
    for q, lst in byq.items():
        busy = sum(e - s for s, e, _ in lst)
        gaps = [lst[i + 1][0] - lst[i][1] for i in range(len(lst) - 1)]
        small = [g for g in gaps if 0 <= g < 20000]
        print('queue %s: %d kernels, busy %.3f ms/step, gaps<20us: %d, sum %.3f ms/step, median %.2f us' % (
            q, len(lst), busy / 1e6 / steps, len(small), sum(small) / 1e6 / steps,
            sorted(small)[len(small) // 2] / 1e3 if small else 0))


if __name__ == '__main__':
    main()
