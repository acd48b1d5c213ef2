"""Builds libt3d.so (the C-ABI HIP library, include/t3d.h) in-tree for gfx950 with hipcc."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
INCLUDE = os.path.join(os.path.dirname(HERE), 'include')
LIB_PATH = os.path.join(HERE, 'libt3d.so')
SOURCES = ['pointmlp.hip', 'pointmlp_x3.hip', 'bn_optim.hip', 'fc.hip', 'heads.hip', 'boxpc.hip', 'poolbwd.hip', 'data.hip', 'weak.hip', 'pair.hip', 'version.hip']
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC']
# per-source flags (none at present: -fno-slp-vectorize on the x3 unit made every rider kernel spill two registers)
EXTRA_FLAGS = {}
MARKER = b'T3D_SOURCE_HASH='


def lib_source_hash():
    """sha256 (16 hex digits) over the HIP sources + the ABI header: identifies the build a PMC summary was taken with
    (bench.py `traffic_summary_predates_this_build`, tools/pmc_traffic.py)."""
    import hashlib
    h = hashlib.sha256()
    for f in sorted(os.listdir(CSRC)):
        with open(os.path.join(CSRC, f), 'rb') as fh:
            h.update(fh.read())
    with open(os.path.join(INCLUDE, 't3d.h'), 'rb') as fh:
        h.update(fh.read())
    return h.hexdigest()[:16]


def embedded_hash(path=LIB_PATH):
    """The source hash compiled into a built library (csrc/version.hip), read from the file without loading it; None if absent."""
    try:
        with open(path, 'rb') as fh:
            data = fh.read()
    except OSError:
        return None
    k = data.find(MARKER)
    if k < 0:
        return None
    return data[k + len(MARKER):k + len(MARKER) + 16].decode('ascii', 'replace')


def _stale():
    """The library is current iff the hash compiled into it equals the hash of the sources (mtimes say nothing about a file that
    travelled through a snapshot or a checkout)."""
    return embedded_hash() != lib_source_hash()


def _object_key(src, extra_flags):
    """What an object file depends on: its source, every header of csrc/ and the ABI header, the flags."""
    import hashlib
    h = hashlib.sha256()
    deps = [os.path.join(CSRC, src)] + sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.h')) + [os.path.join(INCLUDE, 't3d.h')]
    if src == 'pointmlp_x3.hip':
        deps.append(os.path.join(CSRC, 'pointmlp.hip'))      # (it is that file, compiled with T3D_X3_TU)
    for d in deps:
        with open(d, 'rb') as fh:
            h.update(fh.read())
    h.update(' '.join(FLAGS + list(extra_flags)).encode())
    return h.hexdigest()


# Kernels that may spill vector registers to scratch (mangled-name substrings -> why).  Everything else must not: a spill in a GEMM
# main loop is a silent 10-30 % (round-2 review: five bf16 kernels spilled 10-49 VGPRs unnoticed).  build() parses hipcc's
# -Rpass-analysis=kernel-resource-usage remarks and fails on any other kernel with `VGPRs Spill` > 0.
SPILL_ALLOWED = {      # substring of the mangled name -> (most VGPRs it may spill, why)
    'k_pointmlp_bwdILi128ELi128ELi128ENS_7PathX3PE': (2, 'opt-in pre-split-weights form (T3D_X3_PRESPLIT=1, off by default) of the fused x3 backward: 1 VGPR beside the hand-placed iteration'),
    'k_pointmlp_bwdILi128ELi128ELi64ENS_7PathX3PE': (2, 'same'),
    'k_pointmlp_bwdILi128ELi64ELi128ENS_7PathX3PE': (2, 'same'),
    'k_pointmlp_bwdILi128ELi64ELi64ENS_7PathX3PE': (2, 'same'),
    'k_pointmlp_bwd1ILi256ELi128ELi64E': (8, 'one-pass bf16 backward 256 -> 128: 5 VGPRs (20 B) in the epilogue, hand-scheduled kernel at the 256-register cap'),
    'k_pointmlp_bwd1ILi128ELi256ELi64E': (16, 'one-pass bf16 backward 128 -> 256: 11 VGPRs (48 B), same'),
    'k_pointmlp_fwd_resILi128ELi2E': (6, 'persistent activation-resident bf16 forward, K = 128: 3 VGPRs (16 B) -- the next panel\'s raw chunks travel in registers across the epilogue (round 3: 3.72 -> 3.55 ms per config-4 step with it)'),
    'k_strong_loss': (0, 'scalar loss program: a dynamically indexed 67-float private array (not a spill of the allocator, reported as scratch)'),
}


def parse_resource_remarks(text):
    """{mangled kernel name: {'vgpr_spill': n, 'sgpr_spill': n, 'scratch': bytes, 'vgprs': n}} from hipcc's kernel-resource-usage remarks."""
    import re
    out, cur = {}, None
    for line in text.splitlines():
        m = re.search(r'remark:\s+Function Name: (\S+)', line)
        if m:
            cur = out.setdefault(m.group(1), {})
            continue
        if cur is None:
            continue
        for key, pat in (('vgpr_spill', r'VGPRs Spill: (\d+)'), ('sgpr_spill', r'SGPRs Spill: (\d+)'),
                         ('scratch', r'ScratchSize \[bytes/lane\]: (\d+)'), ('vgprs', r'remark:\s+VGPRs: (\d+)')):
            m = re.search(pat, line)
            if m:
                cur[key] = int(m.group(1))
    return out


def check_spills(remarks_by_source):
    """Raises if a kernel spills more VGPRs than SPILL_ALLOWED grants it (nothing, for a kernel that is not listed); returns the
    report that is written next to the library.  T3D_ALLOW_SPILLS=1 turns the error into a warning (another hipcc version may
    allocate differently: the library is functional either way, only slower)."""
    bad, report = [], {}
    for src, text in remarks_by_source.items():
        for name, r in parse_resource_remarks(text).items():
            n = r.get('vgpr_spill', 0)
            if n > 0 or r.get('scratch', 0) > 0:
                report[name] = dict(r, source=src)
            cap = max([v[0] for k, v in SPILL_ALLOWED.items() if k in name] or [0])
            if n > cap:
                bad.append((src, name, 'spills %d VGPRs (allowed: %d)' % (n, cap)))
    if bad:
        msg = ('VGPR spills beyond transferable3d_amd/build.py SPILL_ALLOWED (set T3D_ALLOW_SPILLS=1 to build anyway; the spills are '
               'listed in transferable3d_amd/build/kernel_scratch_report.json):\n' + '\n'.join('  %s: %s %s' % b for b in bad))
        if not os.environ.get('T3D_ALLOW_SPILLS'):
            raise RuntimeError(msg)
        sys.stderr.write('warning: ' + msg + '\n')
    return report


# ISA gate of the x3 main loops (round 6).  The hand-placed iteration (csrc/pointmlp.hip: x3_iter_il) is only as good as the code
# hipcc emits for it: the first builds of round 6 had the IR optimisers move whole staging pieces across ten sched_barrier fences, a
# flat_load (a const table selected against a kernel argument) forcing `s_waitcnt vmcnt(0)` in front of every use, and register copies
# of in-flight loads at the loop's back edge -- none visible in a test, each a silent 10-20 %.  build() reads the device assembly of
# the x3 translation unit (-save-temps) and fails on a main loop (a loop with a barrier and >= 6 MFMAs in a PathX3 / PathX3W kernel)
# that has: two MFMAs with nothing between them more than ISA_MAX_BURST times in a row, a packed-fp32 vector instruction
# (v_pk_add/mul/fma_f32), a flat_ access, an `s_waitcnt vmcnt(0)` while the loop issues loads, or more than ISA_MAX_VALU_RUN vector
# instructions with no MFMA between them.  The map of every such loop is written to build/x3_isa_map.txt (copied to profiles/ per round).
ISA_MAX_BURST = 2
# longest run of vector instructions between two MFMAs.  A 128-wide loop (48 MFMAs per two k-tiles) averages 3.2-5.9 per MFMA and its
# longest step is a piece's element-wise work + the coefficient refill: 14.  The 64-wide loops (24 / 12 MFMAs) carry 6-11 vector
# instructions per MFMA by construction (half or a quarter of the MFMAs per staged element) and two micro-steps share a gap: 20.
ISA_MAX_VALU_RUN = 20
ISA_MAX_VALU_RUN_WIDE = 14
ISA_GATED = ('PathX3E', 'PathX3WE', 'PathX3PE', 'PathX3WPE')      # mangled-name substrings: the x3 kernels of the default program (not the PathX3PC experiment)


def check_x3_isa(asm_path, out_path=None):
    """Raises on a violation (T3D_ALLOW_ISA=1: warns); returns the rows."""
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), 'tools'))
    import isa_loops
    text = open(asm_path).read()
    rows = [r for r in isa_loops.report(text, ISA_GATED) if r['barriers'] >= 1 and r['lds'] > 0]
    bad = []
    for r in rows:
        why = []
        if r['mfma_burst'] > ISA_MAX_BURST:
            why.append('MFMA burst %d' % r['mfma_burst'])
        if r['valu_run'] > (ISA_MAX_VALU_RUN_WIDE if r['mfma'] >= 48 else ISA_MAX_VALU_RUN):
            why.append('VALU run %d' % r['valu_run'])
        if r['pk_f32']:
            why.append('%d packed-fp32 instructions' % r['pk_f32'])
        if r['vmcnt0']:
            why.append('%d s_waitcnt vmcnt(0)' % r['vmcnt0'])
        if r['flat']:
            why.append('%d flat_ accesses' % r['flat'])
        if why:
            bad.append('%s %s: %s' % (isa_loops.short(r['kernel']), r['loop'], ', '.join(why)))
    if out_path:
        with open(out_path, 'w') as fh:
            fh.write('# x3 main loops of %s (tools/isa_loops.py; M = MFMA, v = vector ALU, d = LDS, g = global, s = scalar, w = s_waitcnt, b = s_barrier, n = s_nop)\n'
                     % os.path.basename(asm_path))
            for r in rows:
                fh.write('%s %s  MFMA %d  VALU %d (%.1f per MFMA)  LDS %d  VMEM %d | MFMA burst %d  VALU run %d  packed-fp32 %d  vmcnt(0) %d  flat %d\n    %s\n'
                         % (isa_loops.short(r['kernel']), r['loop'], r['mfma'], r['valu'], r['valu'] / max(r['mfma'], 1), r['lds'], r['vmem'],
                            r['mfma_burst'], r['valu_run'], r['pk_f32'], r['vmcnt0'], r['flat'], r['seq']))
    if not rows:
        bad.append('no x3 main loop found in %s (the gate would pass vacuously)' % asm_path)
    if bad:
        msg = 'x3 main-loop ISA gate (transferable3d_amd/build.py check_x3_isa; T3D_ALLOW_ISA=1 builds anyway):\n' + '\n'.join('  ' + b for b in bad)
        if not os.environ.get('T3D_ALLOW_ISA'):
            raise RuntimeError(msg)
        sys.stderr.write('warning: ' + msg + '\n')
    return rows


def build(force=False, verbose=False):
    """hipcc --offload-arch=gfx950 -shared -fPIC csrc/*.hip -> transferable3d_amd/libt3d.so; fails on an unexpected register spill.
    Objects are reused when their source, the headers and the flags are unchanged; the library carries the hash of its sources."""
    if not force and not _stale():
        return LIB_PATH
    import json
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    bdir = os.path.join(HERE, 'build')
    os.makedirs(bdir, exist_ok=True)
    src_hash = lib_source_hash()
    objs, procs, remarks = [], [], {}
    for src in SOURCES:
        path = os.path.join(CSRC, src)
        obj = os.path.join(bdir, src.replace('.hip', '.o'))
        extra = ['-DT3D_SOURCE_HASH="%s"' % src_hash] if src == 'version.hip' else list(EXTRA_FLAGS.get(src, []))
        asm_ok = True
        if src == 'pointmlp_x3.hip':
            extra = extra + ['-save-temps=obj']      # the device assembly stays next to the object: check_x3_isa
            asm_ok = os.path.exists(os.path.join(bdir, 'pointmlp_x3-hip-amdgcn-amd-amdhsa-gfx950.s'))
        key, keyfile, remfile = _object_key(src, extra), obj + '.key', obj + '.remarks'
        objs.append(obj)
        if not force and asm_ok and os.path.exists(obj) and os.path.exists(keyfile) and os.path.exists(remfile) and open(keyfile).read() == key:
            remarks[src] = open(remfile).read()
            continue
        cmd = [hipcc] + FLAGS + extra + ['-I', INCLUDE, '-Rpass-analysis=kernel-resource-usage', '-c', path, '-o', obj]
        if verbose:
            print(' '.join(cmd))
        procs.append((src, key, keyfile, remfile, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for src, key, keyfile, remfile, pr in procs:
        out, _ = pr.communicate()
        if pr.returncode != 0:
            sys.stderr.write('\n'.join(l for l in out.decode().splitlines() if 'kernel-resource-usage' not in l))
            raise RuntimeError('hipcc failed on %s' % src)
        remarks[src] = out.decode()
        with open(remfile, 'w') as fh:
            fh.write(remarks[src])
        with open(keyfile, 'w') as fh:
            fh.write(key)
    report = check_spills(remarks)
    x3_asm = os.path.join(bdir, 'pointmlp_x3-hip-amdgcn-amd-amdhsa-gfx950.s')
    if os.path.exists(x3_asm):
        check_x3_isa(x3_asm, os.path.join(bdir, 'x3_isa_map.txt'))
    else:
        raise RuntimeError('no device assembly of pointmlp_x3.hip (-save-temps=obj): the ISA gate cannot run')
    with open(os.path.join(bdir, 'kernel_scratch_report.json'), 'w') as fh:
        json.dump(report, fh, indent=1, sort_keys=True)
    subprocess.check_call([hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB_PATH] + objs)
    assert embedded_hash() == src_hash, 'the library does not carry the hash of its sources'
    return LIB_PATH


if __name__ == '__main__':
    print(build(force='--force' in sys.argv, verbose=True))
