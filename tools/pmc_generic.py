#!/usr/bin/env python3
"""Any set of rocprofv3 PMC counters per libt3d kernel: sums per launch, averaged over the launches sampled, beside the launch duration.
  rocprofv3 --pmc A B C --output-format csv -d <dir> -o run -- python3 bench.py ... ;  python tools/pmc_generic.py <dir> [--per_cycle GRBM_GUI_ACTIVE]
With --per_cycle X every other counter is also printed divided by X / 8 (X summed over the 8 XCDs: per chip-cycle)."""
import argparse
import csv
import glob
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_traffic import label          # noqa: E402

csv.field_size_limit(1 << 30)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('dir')
    ap.add_argument('--per_cycle', default=None)
    ap.add_argument('--top', type=int, default=16)
    a = ap.parse_args()
    per, names = {}, []
    for f in glob.glob(os.path.join(a.dir, '**', '*counter_collection.csv'), recursive=True):
        with open(f, newline='') as fh:
            for row in csv.DictReader(fh):
                k = label(row['Kernel_Name'])
                if k is None:
                    continue
                d = per.setdefault((k, row['Dispatch_Id']), {'k': k})
                d[row['Counter_Name']] = float(row['Counter_Value'])
                d['ns'] = float(row['End_Timestamp']) - float(row['Start_Timestamp'])
                if row['Counter_Name'] not in names:
                    names.append(row['Counter_Name'])
    acc = {}
    for d in per.values():
        e = acc.setdefault(d['k'], {'n': 0, 'ns': 0.0})
        e['n'] += 1
        e['ns'] += d['ns']
        for c in names:
            e[c] = e.get(c, 0.0) + d.get(c, 0.0)
    rows = sorted(acc.items(), key=lambda kv: -kv[1]['ns'])[:a.top]
    for k, e in rows:
        n = e['n']
        s = '%-34s x%-4d %8.1f us ' % (k, n, e['ns'] / n / 1e3)
        for c in names:
            s += '  %s %.4g' % (c, e[c] / n)
            if a.per_cycle and c != a.per_cycle and e.get(a.per_cycle, 0) > 0:
                s += ' (%.3f / chip-cycle)' % (e[c] / (e[a.per_cycle] / 8))
        print(s)


if __name__ == '__main__':
    main()
