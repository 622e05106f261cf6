#!/bin/bash
# HBM traffic (FETCH_SIZE x 2 + WRITE_SIZE, KiB units: MI355X_MICROARCH.md) and kernel time of the two persistent Cholesky kernels, B = 256, M = 512
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_left
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
cd $R
for left in 1 0; do
  export DPGP_POTRF_LEFT=$left
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/f$left -- python3 scratch/prof_linalg.py 3 > $O/f$left.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/w$left -- python3 scratch/prof_linalg.py 3 > $O/w$left.log 2>&1
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/t$left -- python3 scratch/prof_linalg.py 8 > $O/t$left.log 2>&1
done
python3 - $O <<'PY'
import csv, glob, sys, collections
O = sys.argv[1]
for left in (1, 0):
    out = {}
    for tag, col in (('f', 'FETCH_SIZE'), ('w', 'WRITE_SIZE')):
        vals = collections.defaultdict(list)
        for f in glob.glob('%s/%s%d/*/*counter_collection.csv' % (O, tag, left)):
            for r in csv.DictReader(open(f)):
                if 'persistent' in r['Kernel_Name'] and r['Counter_Name'] == col: vals[r['Kernel_Name'][:30]].append(float(r['Counter_Value']))
        for k, v in vals.items(): out.setdefault(k, {})[col] = sum(v) / len(v)
    for f in glob.glob('%s/t%d/*/*kernel_stats.csv' % (O, left)):
        for r in csv.DictReader(open(f)):
            if 'persistent' in r['Name']: out.setdefault(r['Name'][:30], {})['us'] = float(r['AverageNs']) / 1e3
    for k, v in out.items():
        if 'FETCH_SIZE' in v: print('DPGP_POTRF_LEFT=%d %s: fetch %.0f MB (x2 rule), write %.0f MB, total %.2f GB, %.1f us' % (left, k, 2 * v['FETCH_SIZE'] * 1024 / 1e6, v['WRITE_SIZE'] * 1024 / 1e6, (2 * v['FETCH_SIZE'] + v['WRITE_SIZE']) * 1024 / 1e9, v.get('us', 0)))
PY
find $O -name "*.csv" -delete
