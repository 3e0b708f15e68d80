"""Which hardware queue every stream of tools/many_executors.py landed on, per executor (a run of 25 steps each):
reads a rocprofv3 --kernel-trace CSV, cuts it into windows of 25 k_adam_flat launches and prints, per window and
Queue_Id, the number of main-chain kernels (k_gconv_pairs / k_gconv_tile / k_bn_*) and of side-stream kernels
(k_gconv_dw*), the busy time and the window's wall time -- a slow (caller stream, side stream) pair shows up as both
kinds of kernels on ONE queue id, or as two queues whose kernels never overlap in time."""
import csv, glob, sys, collections
d = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 25
f = glob.glob(d + '/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
adam = [i for i, r in enumerate(rows) if 'k_adam_flat' in r['Kernel_Name']]
print('%d kernels, %d steps' % (len(rows), len(adam)))
w0 = 0
for k in range(len(adam) // steps):
    a = adam[k * steps + 5] if k * steps + 5 < len(adam) else adam[-1]      # skip the 5 warm-up steps of the window
    b = adam[(k + 1) * steps - 1]
    win = rows[a + 1:b + 1]
    if not win:
        continue
    wall = (int(win[-1]['End_Timestamp']) - int(win[0]['Start_Timestamp'])) / 1e6
    byq = collections.defaultdict(lambda: [0, 0, 0, 0.0])
    ivs = collections.defaultdict(list)
    for r in win:
        q = (r.get('Queue_Id'), r.get('Stream_Id'))
        n = r['Kernel_Name']
        kind = 1 if 'k_gconv_dw' in n or 'k_dw' in n else (0 if ('k_gconv' in n or 'k_bn' in n) else 2)
        byq[q][kind] += 1
        byq[q][3] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6
        ivs[q].append((int(r['Start_Timestamp']), int(r['End_Timestamp'])))
    print('executor %d: %.3f ms per step over %d steps' % (k + 1, wall / (steps - 6), steps - 6))
    for q, v in sorted(byq.items(), key=lambda kv: -kv[1][3]):
        print('   queue %s stream %s: %5d chain kernels, %5d weight-gradient kernels, %5d other, busy %.2f ms' % (q[0], q[1], v[0], v[1], v[2], v[3]))
    # overlap between the two busiest queues
    qs = sorted(ivs, key=lambda q: -byq[q][3])[:2]
    if len(qs) == 2:
        A, B = sorted(ivs[qs[0]]), sorted(ivs[qs[1]])
        i = j = 0; ov = 0
        while i < len(A) and j < len(B):
            lo, hi = max(A[i][0], B[j][0]), min(A[i][1], B[j][1])
            if hi > lo: ov += hi - lo
            if A[i][1] < B[j][1]: i += 1
            else: j += 1
        print('   time both of the two busiest queues run a kernel: %.2f ms' % (ov / 1e6))
