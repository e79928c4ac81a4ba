#!/usr/bin/env python
"""Print a rocprofv3 kernel_stats.csv compactly: tools/kstats.py <csv> [n]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 25
for r in rows[:n]:
    print(r['Name'][:100].ljust(100), r['Calls'].rjust(7), ('%.1f' % (float(r['TotalDurationNs']) / 1e6)).rjust(9), 'ms', ('%.2f' % (float(r['AverageNs']) / 1e3)).rjust(10), 'us', r['Percentage'])
