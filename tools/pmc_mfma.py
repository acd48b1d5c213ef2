#!/usr/bin/env python3
"""MFMA-pipe utilisation and shader clock per libt3d kernel from one rocprofv3 PMC pass
(--pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE; tools/profile_round.sh).

SQ_VALU_MFMA_BUSY_CYCLES counts busy cycles summed over every SIMD of the chip (MI355X_MICROARCH.md: "counts cycles");
GRBM_GUI_ACTIVE is summed over the 8 XCDs.  Per launch:
    clock  = GRBM_GUI_ACTIVE / 8 / duration
    util   = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 256 CUs * 4 SIMDs)

  python tools/pmc_mfma.py gpurun_out/r01/pmc_mfma -o profiles/pmc_mfma.json
"""
import argparse
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_traffic import label          # noqa: E402

csv.field_size_limit(1 << 30)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('dir')
    ap.add_argument('-o', '--out', default='profiles/pmc_mfma.json')
    ap.add_argument('--note', default='')
    a = ap.parse_args()
    per = {}
    for f in glob.glob(os.path.join(a.dir, '**', '*counter_collection.csv'), recursive=True):
        with open(f, newline='') as fh:
            for row in csv.DictReader(fh):
                k = label(row['Kernel_Name'])
                if k is None:
                    continue
                d = per.setdefault((k, row['Dispatch_Id']), {'k': k})
                d[row['Counter_Name']] = float(row['Counter_Value'])
                d['ns'] = float(row['End_Timestamp']) - float(row['Start_Timestamp'])
    acc = {}
    for d in per.values():
        if 'SQ_VALU_MFMA_BUSY_CYCLES' not in d or 'GRBM_GUI_ACTIVE' not in d:
            continue
        e = acc.setdefault(d['k'], [0.0, 0.0, 0.0, 0])
        e[0] += d['SQ_VALU_MFMA_BUSY_CYCLES']
        e[1] += d['GRBM_GUI_ACTIVE']
        e[2] += d['ns']
        e[3] += 1
    out = {'_doc': __doc__.strip().split('\n\n')[1], '_note': a.note, 'kernels': {}}
    for k, (busy, gui, ns, n) in sorted(acc.items(), key=lambda kv: -kv[1][0]):
        if gui <= 0:
            continue
        out['kernels'][k] = {'launches_sampled': n, 'mfma_util': busy / (gui / 8 * 1024), 'clock_ghz': gui / 8 / ns,
                             # GRBM_GUI_ACTIVE also covers dispatch/drain outside the kernel's own timestamps, which inflates the
                             # clock (and deflates util) of launches of a few microseconds: second figure at the nominal 2.4 GHz
                             'mfma_util_at_2p4ghz_over_kernel_time': busy / (ns * 2.4 * 1024),
                             'avg_duration_us_under_pmc': ns / n / 1e3}
        sys.stdout.write('%-34s x%-4d mfma util %5.1f %% (%5.1f %% of kernel time at 2.4 GHz)   clock %.2f GHz   %7.1f us\n' % (
            k, n, 100 * busy / (gui / 8 * 1024), 100 * busy / (ns * 2.4 * 1024), gui / 8 / ns, ns / n / 1e3))
    with open(a.out, 'w') as fh:
        json.dump(out, fh, indent=1, sort_keys=True)


if __name__ == '__main__':
    main()
