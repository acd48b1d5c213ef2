"""Builds libt3d.so (the C-ABI HIP library, include/t3d.h) in-tree for gfx950 with hipcc."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
INCLUDE = os.path.join(os.path.dirname(HERE), 'include')
LIB_PATH = os.path.join(HERE, 'libt3d.so')
SOURCES = ['pointmlp.hip', 'bn_optim.hip', 'fc.hip', 'heads.hip', 'boxpc.hip', 'poolbwd.hip', 'data.hip', 'weak.hip', 'pair.hip']


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


def _stale():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(INCLUDE, 't3d.h')]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    """hipcc --offload-arch=gfx950 -shared -fPIC csrc/*.hip -> transferable3d_amd/libt3d.so"""
    if not force and not _stale():
        return LIB_PATH
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    objs = []
    procs = []
    os.makedirs(os.path.join(HERE, 'build'), exist_ok=True)
    for src in SOURCES:
        path = os.path.join(CSRC, src)
        if not os.path.exists(path):
            continue
        obj = os.path.join(HERE, 'build', src.replace('.hip', '.o'))
        cmd = [hipcc, '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-I', INCLUDE, '-c', path, '-o', obj]
        if verbose:
            print(' '.join(cmd))
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
        objs.append(obj)
    for src, pr in procs:
        out, _ = pr.communicate()
        if pr.returncode != 0:
            sys.stderr.write(out.decode())
            raise RuntimeError('hipcc failed on %s' % src)
    cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB_PATH] + objs
    subprocess.check_call(cmd)
    return LIB_PATH


if __name__ == '__main__':
    print(build(force='--force' in sys.argv, verbose=True))
