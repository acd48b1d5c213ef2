#!/usr/bin/env python3
"""Mean counter values per kernel from a rocprofv3 --pmc output directory:  python tools/pmc_one.py DIR [kernel substring]"""
import csv, glob, os, sys
csv.field_size_limit(1 << 30)
acc = {}
for f in glob.glob(os.path.join(sys.argv[1], '**', '*counter_collection.csv'), recursive=True):
    for row in csv.DictReader(open(f, newline='')):
        if len(sys.argv) > 2 and sys.argv[2] not in row['Kernel_Name']:
            continue
        d = acc.setdefault(row['Counter_Name'], [0.0, 0])
        d[0] += float(row['Counter_Value']); d[1] += 1
for k, (v, n) in sorted(acc.items()):
    print('%-32s %16.0f  (x%d)' % (k, v / n, n))
