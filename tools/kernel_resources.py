#!/usr/bin/env python3
"""VGPR / SGPR / spill / LDS / scratch figures of every kernel in a gfx950 device assembly (hipcc -save-temps) -- the build check
against register spills in the hot kernels.

  hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -c transferable3d_amd/csrc/pointmlp.hip -o /tmp/pm.o -save-temps=obj
  python tools/kernel_resources.py /tmp/pointmlp-hip-amdgcn-amd-amdhsa-gfx950.s [substring ...]
"""
import re
import shutil
import subprocess
import sys


def demangle(names):
    filt = shutil.which('c++filt') or shutil.which('llvm-cxxfilt')
    if not filt:
        return names
    out = subprocess.run([filt], input='\n'.join(names), capture_output=True, text=True).stdout.split('\n')
    return [o.replace('(anonymous namespace)::', '') for o in out[:len(names)]]


def kernels(path):
    s = open(path).read()
    rows = []
    for b in s.split('  - .agpr_count:')[1:]:
        g = lambda k: int(re.search(r'\.%s:\s+(\d+)' % k, b).group(1))
        rows.append(dict(name=re.search(r'\.name:\s+(\S+)', b).group(1), vgpr=g('vgpr_count'), vgpr_spill=g('vgpr_spill_count'),
                         sgpr=g('sgpr_count'), sgpr_spill=g('sgpr_spill_count'), lds=g('group_segment_fixed_size'),
                         scratch=g('private_segment_fixed_size')))
    for r, n in zip(rows, demangle([r['name'] for r in rows])):
        r['name'] = n
    return rows


if __name__ == '__main__':
    pats = sys.argv[2:]
    for r in kernels(sys.argv[1]):
        if pats and not any(p in r['name'] for p in pats):
            continue
        print('%4d vgpr %3d spill | %3d sgpr %3d spill | %6d lds %5d scratch | %s' %
              (r['vgpr'], r['vgpr_spill'], r['sgpr'], r['sgpr_spill'], r['lds'], r['scratch'], r['name'][:110]))
