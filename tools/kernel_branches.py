#!/usr/bin/env python3
"""Branches, MFMAs and instruction lines per kernel of a gfx950 device assembly (hipcc -save-temps): a hot kernel with hundreds of
s_cbranch usually carries a run-time flag inside an unrolled element loop (a branch per element).
  python tools/kernel_branches.py /tmp/st/pointmlp-hip-amdgcn-amd-amdhsa-gfx950.s [substring ...]"""
import re
import subprocess
import sys


def main():
    s = open(sys.argv[1]).read()
    idx = [(m.start(), m.group(1)) for m in re.finditer(r'^(_Z\S+?):', s, re.M)]
    dem = subprocess.run(['c++filt'], input='\n'.join(n for _, n in idx), capture_output=True, text=True).stdout.split('\n')
    rows = []
    for i, (pos, _) in enumerate(idx):
        b = s[pos:idx[i + 1][0] if i + 1 < len(idx) else len(s)].split('s_endpgm')[0]
        rows.append((b.count('s_cbranch'), b.count('v_mfma'), len(b.splitlines()), dem[i].replace('(anonymous namespace)::', '')[:130]))
    pats = sys.argv[2:]
    for r in sorted(rows, reverse=True):
        if not pats or any(p in r[3] for p in pats):
            print('%5d branches %5d mfma %6d lines  %s' % r)


if __name__ == '__main__':
    main()
