"""Aggregate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (counter_collection.csv) into profiles/<name>.json:
HBM-side bytes per launch of the gather-conv kernels (forward + input gradient, and weight gradient).
Correction per MI355X_MICROARCH.md (HBM / rocprofv3 section): counters are KB; on gfx950 FETCH_SIZE reports half of
the bytes of wide coalesced reads -> doubled; WRITE_SIZE as is."""
import csv, glob, json, sys
fetch_dir, write_dir, dst = sys.argv[1], sys.argv[2], sys.argv[3]

def collect(d, counter):
    f = glob.glob(d + '/**/*counter_collection.csv', recursive=True)[0]
    agg = {}
    for r in csv.DictReader(open(f)):
        if r.get('Counter_Name') != counter:
            continue
        name = r['Kernel_Name']
        key = 'gconv_fwd_dx' if ('k_gconv_tile' in name or 'k_gconv_pairs' in name) else ('k_gconv_dw' if 'k_gconv_dw' in name and 'small' not in name else None)
        if key is None:
            continue
        a = agg.setdefault(key, [0.0, 0])
        a[0] += float(r['Counter_Value']); a[1] += 1
    return {k: {'avg_kb_per_launch': v[0] / v[1], 'launches': v[1]} for k, v in agg.items()}

raw = {'FETCH_SIZE': collect(fetch_dir, 'FETCH_SIZE'), 'WRITE_SIZE': collect(write_dir, 'WRITE_SIZE')}
fw = raw['FETCH_SIZE']['gconv_fwd_dx']['avg_kb_per_launch']
ww = raw['WRITE_SIZE']['gconv_fwd_dx']['avg_kb_per_launch']
out = {
    'source': 'rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace), python3 bench.py --steps 3 '
              '--warmup 2 --no-cpu-baseline, MI355X, default kernels (k_gconv_pairs / k_gconv_tile by shape, k_gconv_dw2)',
    'raw': raw,
    'correction': 'MI355X_MICROARCH.md section HBM: on gfx950 FETCH_SIZE reports 1/2 of the bytes of a wide coalesced '
                  '(16 B/lane) read -> doubled; WRITE_SIZE taken as is; counters are KB',
    'gconv_traffic_bytes_per_launch': (2.0 * fw + ww) * 1024.0,
}
json.dump(out, open(dst, 'w'), indent=1)
print(json.dumps(out, indent=1))
