#!/usr/bin/env python3
"""Instruction-mix map of the MFMA loops of a gfx950 assembly listing (hipcc -S --cuda-device-only).

For every kernel whose name matches a filter, every innermost loop (a run of basic blocks closed by a backward branch) that holds
MFMAs is reported as: MFMAs, VALU, LDS, VMEM, SALU, waits; the longest MFMA burst (MFMAs with nothing but scalar instructions between
them), the longest VALU run (vector instructions with no MFMA between them), the number of empty MFMA gaps, and the packed-fp32
vector instructions (v_pk_add/mul/fma_f32: each holds the issue port several times as long as its two single halves beside MFMAs,
MI355X_MICROARCH.md 'price of one filler').  build.py uses check() as the ISA gate of the x3 main loops.

    python tools/isa_loops.py listing.s [name-substring ...] [--map]      # --map prints the M/v/d/g/s sequence of each loop
"""
import re
import sys

PK = re.compile(r'^v_pk_(add|mul|fma)_f32')


def kind(op):
    if op.startswith('v_mfma') or op.startswith('v_smfmac'):
        return 'M'
    if op.startswith('ds_'):
        return 'd'
    if op.startswith(('global_', 'buffer_', 'flat_', 'scratch_')):
        return 'g'
    if op.startswith('v_'):
        return 'v'
    if op.startswith('s_waitcnt'):
        return 'w'
    if op.startswith('s_barrier'):
        return 'b'
    if op.startswith('s_nop'):
        return 'n'
    if op.startswith('s_'):
        return 's'
    return '?'


def kernels(text):
    """{name: [(label or None, [ops])...]} -- basic blocks in layout order."""
    out, cur, blocks = {}, None, None
    for line in text.splitlines():
        s = line.strip()
        m = re.match(r'^(_Z\w+):', line)
        if m and '@' in line:
            cur = m.group(1)
            blocks = out.setdefault(cur, [[None, []]])
            continue
        if cur is None:
            continue
        if s.startswith('.Lfunc_end') or s.startswith('s_endpgm'):
            if s.startswith('.Lfunc_end'):
                cur = None
            continue
        m = re.match(r'^(\.LBB\d+_\d+):', s)
        if m:
            blocks.append([m.group(1), []])
            continue
        if not s or s.startswith((';', '.', '//')):
            continue
        op = s.split()[0]
        blocks[-1][1].append((op, s))
    return out


def loops(blocks):
    """innermost loops: (first block index, last block index) for every backward branch whose span holds no other backward branch target
    inside it that starts later (approximation good enough for hipcc's layout: loops are contiguous)."""
    index = {b[0]: i for i, b in enumerate(blocks) if b[0]}
    spans = []
    for i, (_, ops) in enumerate(blocks):
        for op, s in ops:
            if op.startswith(('s_cbranch', 's_branch')):
                tgt = s.split()[-1]
                if tgt in index and index[tgt] <= i:
                    spans.append((index[tgt], i))
    inner = [sp for sp in spans if not any(o != sp and sp[0] <= o[0] and o[1] <= sp[1] for o in spans)]
    return sorted(set(inner))


def analyse(ops):
    seq = ''.join(kind(op) for op, _ in ops)
    n = {k: seq.count(k) for k in 'Mvdgswbn'}
    pk = sum(1 for op, _ in ops if PK.match(op))
    # s_waitcnt vmcnt(0) inside a loop that also issues loads drains the prefetch (a flat_load forces it: out-of-order return)
    vm0 = sum(1 for op, s_ in ops if op == 's_waitcnt' and re.search(r'vmcnt\(0\)', s_)) if any(kind(op) == 'g' for op, _ in ops) else 0
    flat = sum(1 for op, _ in ops if op.startswith('flat_'))
    movs = sum(1 for op, _ in ops if op.startswith('v_mov_b32') or op.startswith('v_accvgpr'))
    # bursts: MFMAs separated only by scalar / wait / nop
    burst = longest = 0
    for c in seq:
        if c == 'M':
            burst += 1
            longest = max(longest, burst)
        elif c in 'vdg':
            burst = 0
    # the loop is a cycle: rotate so that it starts at an MFMA, then measure the gaps
    k = seq.find('M')
    rot = seq[k:] + seq[:k] if k >= 0 else seq
    gaps = rot.split('M')[1:] if k >= 0 else []
    vrun = max([g.count('v') for g in gaps] or [0])
    vdrun = max([g.count('v') + g.count('d') + g.count('g') for g in gaps] or [0])
    empty = sum(1 for g in gaps if not any(c in g for c in 'vdg'))
    # a burst may wrap around the back edge
    if k >= 0 and n['M'] > 0:
        b2 = cur = 0
        for c in rot + rot:
            if c == 'M':
                cur += 1
                b2 = max(b2, cur)
            elif c in 'vdg':
                cur = 0
        longest = min(max(longest, b2), n['M'])
    return dict(mfma=n['M'], valu=n['v'], lds=n['d'], vmem=n['g'], salu=n['s'], waits=n['w'], barriers=n['b'], nops=n['n'], pk_f32=pk,
                mfma_burst=longest, valu_run=vrun, vec_run=vdrun, empty_gaps=empty, seq=rot, vmcnt0=vm0, flat=flat, movs=movs)


def report(text, filters=(), want_map=False, min_mfma=6):
    rows = []
    for name, blocks in kernels(text).items():
        if filters and not any(f in name for f in filters):
            continue
        for a, b in loops(blocks):
            ops = [o for blk in blocks[a:b + 1] for o in blk[1]]
            # the last block runs on to the next label in the listing: the loop ends at its (last) branch back to the header
            back = [i for i, (op, s_) in enumerate(ops) if op.startswith(('s_cbranch', 's_branch')) and s_.split()[-1] == blocks[a][0]]
            if back:
                ops = ops[:back[-1] + 1]
            r = analyse(ops)
            if r['mfma'] < min_mfma:
                continue
            r['kernel'], r['loop'] = name, blocks[a][0]
            rows.append(r)
    return rows


def check(rows, max_valu_run, max_burst, max_pk=0):
    """the ISA gate: returns the list of violations (strings)."""
    bad = []
    for r in rows:
        if r['valu_run'] > max_valu_run or r['mfma_burst'] > max_burst or r['pk_f32'] > max_pk:
            bad.append('%s %s: longest VALU run %d (allowed %d), longest MFMA burst %d (allowed %d), packed-fp32 %d (allowed %d)' % (
                r['kernel'], r['loop'], r['valu_run'], max_valu_run, r['mfma_burst'], max_burst, r['pk_f32'], max_pk))
    return bad


def short(name):
    """a readable form of a mangled kernel name: k_pointmlp_bwd<128,128,128,PathX3>"""
    m = re.match(r'_ZN12_GLOBAL__N_1\d+(k_\w+?)I(.*?)EEv', name)
    if not m:
        return name[:60]
    targs = []
    for t in re.findall(r'Li(\d+)E|Lb([01])E|NS_\d+(\w+?)E(?=E|L|N|f|$)|(f)|(DF16b)', m.group(2)):
        targs.append(t[0] or ('true' if t[1] == '1' else 'false' if t[1] else '') or t[2] or ('float' if t[3] else '') or ('bf16' if t[4] else ''))
    return '%s<%s>' % (m.group(1), ','.join(a for a in targs if a))


def main():
    args = [a for a in sys.argv[1:] if not a.startswith('--')]
    want_map = '--map' in sys.argv
    text = open(args[0]).read()
    if '--summary' in sys.argv:
        for r in report(text, args[1:]):
            print('%-52s %-9s MFMA %2d VALU %3d (%4.1f/MFMA) LDS %2d VMEM %2d | MFMA burst %2d  VALU run %2d  vector run %2d  empty gaps %2d  packed-fp32 %2d  vmcnt(0) %d  flat %d  v_mov %d' % (
                short(r['kernel'])[:52], r['loop'][4:], r['mfma'], r['valu'], r['valu'] / r['mfma'], r['lds'], r['vmem'], r['mfma_burst'], r['valu_run'],
                r['vec_run'], r['empty_gaps'], r['pk_f32'], r['vmcnt0'], r['flat'], r['movs']))
        return
    for r in report(text, args[1:], want_map):
        print('%s  %s\n    MFMA %d  VALU %d (%.1f per MFMA)  LDS %d  VMEM %d  SALU %d  waits %d  barriers %d  nops %d | MFMA burst %d  VALU run %d  '
              'vector run %d  empty gaps %d  packed-fp32 %d' % (r['kernel'], r['loop'], r['mfma'], r['valu'], r['valu'] / max(r['mfma'], 1), r['lds'], r['vmem'],
                                                           r['salu'], r['waits'], r['barriers'], r['nops'], r['mfma_burst'], r['valu_run'], r['vec_run'],
                                                           r['empty_gaps'], r['pk_f32']))
        if want_map:
            print('    ' + r['seq'])


if __name__ == '__main__':
    main()
