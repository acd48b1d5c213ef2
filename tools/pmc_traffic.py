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


def lib_source_hash():
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from transferable3d_amd.build import lib_source_hash as h
    return h()


def label(kernel_name):
    """bench.py's label of a kernel: GEMM kernels by name + integer template arguments (tile sizes), the others by bare name.
    rocprofv3 leaves names with a __bf16 template argument mangled (_ZN12_GLOBAL__N_114k_pointmlp_fwdILi128ELb0ENS_8PathBF16EDF16bEE...):
    both spellings are handled."""
    if kernel_name.startswith('_Z'):
        m = re.search(r'(\d+)k_', kernel_name)
        if not m:
            return None
        digits, pos = m.group(1), m.end() - 2
        name = rest = None
        for i in range(len(digits)):          # "..._114k_pointmlp_fwd": the length prefix is a suffix of the digit run
            n = int(digits[i:])
            cand, nxt = kernel_name[pos:pos + n], kernel_name[pos + n:pos + n + 1]
            if len(cand) == n and nxt in ('I', 'E') and re.fullmatch(r'k_\w+', cand):
                name, rest = cand, kernel_name[pos + n:]
                break
        if name is None:
            return None
        ints = re.findall(r'Li(\d+)E', rest.split('Ev', 1)[0]) if rest.startswith('I') else []
    else:
        m = re.search(r'(k_\w+)(?:<([^>]*)>)?\(', kernel_name)
        if not m:
            return None
        name, targs = m.group(1), [t.strip() for t in (m.group(2) or '').split(',') if t.strip()]
        ints = [t for t in targs if t.isdigit()]          # tile sizes; the bool / type template arguments are not part of the label
    if 'PathX3' in kernel_name:          # fp32 layers on the bf16 matrix pipe (three-term operands): k_..._x3<...> / k_..._x3_r<...>
        name = name[:-2] + '_x3_r' if name.endswith('_r') else name + '_x3'
    if name.startswith(('k_pointmlp', 'k_pool_bwd_stage')) and ints:
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
    ap.add_argument('--workload', default='A')
    ap.add_argument('--batch_size', type=int, default=32)
    ap.add_argument('--num_point', type=int, default=1024)
    ap.add_argument('--num_channel', type=int, default=4)
    ap.add_argument('--dtype', default='f32')
    a, _unknown = ap.parse_known_args()      # profile_round.sh forwards the bench flags; only the workload keys matter here
    fe, wr = collect(a.fetch_dir, 'FETCH_SIZE'), collect(a.write_dir, 'WRITE_SIZE')
    out = {'_doc': 'per-launch HBM traffic from rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes); '
                   'bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950: FETCH_SIZE counts 128-B requests at 64 B)', '_note': a.note,
           # what the summary was taken for: bench.py reports `traffic` only for the same workload / size / dtype and flags a
           # summary that predates the current kernel sources
           '_meta': {'workload': a.workload, 'B': a.batch_size, 'N': a.num_point, 'C': a.num_channel, 'dtype': a.dtype,
                     'lib_source_hash': lib_source_hash()},
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
