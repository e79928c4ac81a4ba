#!/usr/bin/env python
"""Aggregate a rocprofv3 --pmc counter_collection CSV per kernel: tools/pmc_summary.py <dir>"""
import collections, csv, glob, json, sys
d = sys.argv[1]
files = glob.glob(d + '/**/*counter_collection.csv', recursive=True)
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for f in files:
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'][:80]
        c = agg[k][r['Counter_Name']]
        c[0] += 1
        c[1] += float(r['Counter_Value'])
out = {}
for k, cs in sorted(agg.items(), key=lambda kv: -max(v[1] for v in kv[1].values())):
    out[k] = {c: {'launches': v[0], 'sum': v[1], 'avg': v[1] / v[0]} for c, v in cs.items()}
print(json.dumps(out, indent=1))
