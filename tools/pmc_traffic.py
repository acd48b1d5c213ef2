#!/usr/bin/env python3
"""HBM bytes per launch of every libt3d kernel from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; tools/profile_round.sh).

Units and gfx950 corrections per /opt/skills/guides (cdna_hip_programming.md section 7, MI355X_MICROARCH.md "HBM"):
both counters are in KiB; FETCH_SIZE reads exactly half of a wide coalesced streaming read on gfx950, so
    bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024.
Kernels are keyed the way bench.py labels them: the GEMM kernels by name + tile sizes (k_pointmlp_fwd<BN>,
k_pointmlp_wgrad<BMK,BN>, ...), other kernels by bare name.

  python tools/pmc_traffic.py gpurun_out/r01/pmc_fetch gpurun_out/r01/pmc_write -o profiles/pmc_traffic.json
"""
import argparse
import csv
import glob
import json
import os
import re
import sys

csv.field_size_limit(1 << 30)


def label(kernel_name):
    m = re.search(r'(k_\w+)(?:<([^>]*)>)?\(', kernel_name)
    if not m:
        return None
    name, targs = m.group(1), [t.strip() for t in (m.group(2) or '').split(',') if t.strip()]
    ints = [t for t in targs if t.isdigit()]          # tile sizes; the bool template flags are not part of the label
    if name.startswith('k_pointmlp') and ints:
        return '%s<%s>' % (name, ','.join(ints))
    return name


def collect(dirname, counter):
    acc = {}
    files = glob.glob(os.path.join(dirname, '**', '*counter_collection.csv'), recursive=True)
    assert files, 'no counter_collection.csv under %s' % dirname
    for f in files:
        with open(f, newline='') as fh:
            for row in csv.DictReader(fh):
                if row['Counter_Name'] != counter:
                    continue
                k = label(row['Kernel_Name'])
                if k is None:
                    continue
                d = acc.setdefault(k, [0.0, 0])
                d[0] += float(row['Counter_Value'])
                d[1] += 1
    return acc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('fetch_dir')
    ap.add_argument('write_dir')
    ap.add_argument('-o', '--out', default='profiles/pmc_traffic.json')
    ap.add_argument('--note', default='')
    a = ap.parse_args()
    fe, wr = collect(a.fetch_dir, 'FETCH_SIZE'), collect(a.write_dir, 'WRITE_SIZE')
    out = {'_doc': 'per-launch HBM traffic from rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes); '
                   'bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950: FETCH_SIZE counts 128-B requests at 64 B)', '_note': a.note,
           'kernels': {}}
    for k in sorted(set(fe) | set(wr)):
        f, nf = fe.get(k, (0.0, 0))
        w, nw = wr.get(k, (0.0, 0))
        fk, wk = (f / nf if nf else 0.0), (w / nw if nw else 0.0)
        out['kernels'][k] = {'launches_sampled': [nf, nw], 'fetch_size_kib_avg': fk, 'write_size_kib_avg': wk,
                             'read_bytes_per_launch': 2 * fk * 1024, 'write_bytes_per_launch': wk * 1024,
                             'bytes_per_launch': (2 * fk + wk) * 1024}
    with open(a.out, 'w') as fh:
        json.dump(out, fh, indent=1, sort_keys=True)
    for k, v in sorted(out['kernels'].items(), key=lambda kv: -kv[1]['bytes_per_launch'] * kv[1]['launches_sampled'][0]):
        sys.stdout.write('%-32s x%-5d  read %8.2f MB  write %8.2f MB\n' % (k, v['launches_sampled'][0], v['read_bytes_per_launch'] / 1e6,
                                                                         v['write_bytes_per_launch'] / 1e6))


if __name__ == '__main__':
    main()
