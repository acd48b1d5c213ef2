"""CPU: libt3d.so builds for gfx950, loads, and exports every entry point include/t3d.h declares (no
compute calls without a GPU); the ctypes structs mirror the header's field order."""
import os
import re

from transferable3d_amd import abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header():
    return open(os.path.join(ROOT, 'include', 't3d.h')).read()


def test_library_builds_and_exports_every_declared_symbol():
    from transferable3d_amd.build import build
    build()
    lib = abi.load()
    declared = set(re.findall(r'\bint\s+(t3d_\w+)\s*\(', _header()))
    assert declared, 'no declarations parsed'
    assert declared == set(abi.ENTRY_POINTS), declared ^ set(abi.ENTRY_POINTS)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.t3d_abi_version() == abi.ABI_VERSION == 3
    assert int(re.search(r'#define T3D_ABI_VERSION (\d+)', _header()).group(1)) == 3


def test_library_is_bound_to_the_sources_it_was_built_from(tmp_path, monkeypatch):
    """csrc/version.hip carries sha256(csrc/*, include/t3d.h); abi.load() refuses a library built from other sources (*.so files are
    git-ignored but ship to the GPU box: a stale file must not be tested or benchmarked as HEAD)."""
    import shutil
    import pytest
    from transferable3d_amd import build as B
    B.build()
    lib = abi.load()
    assert abi.source_hash_of(lib) == B.lib_source_hash() == B.embedded_hash()
    # a source tree that differs from what the library was built from
    fake = tmp_path / 'csrc'
    shutil.copytree(B.CSRC, fake)
    with open(fake / 'version.hip', 'a') as fh:
        fh.write('// edited after the build\n')
    monkeypatch.setattr(B, 'CSRC', str(fake))
    assert B._stale()
    with pytest.raises(abi.T3DError, match='built from other sources'):
        abi.load()
    monkeypatch.setenv('T3D_ALLOW_STALE_LIB', '1')
    abi.load()


def test_ctypes_structs_follow_header_field_order():
    h = _header()
    pairs = {'t3d_act_src': abi.ActSrc, 't3d_dy_src': abi.DySrc, 't3d_pointmlp_fwd_args': abi.PointMlpFwdArgs,
             't3d_bn_fwd_finalize_args': abi.BnFwdFinalizeArgs, 't3d_pool_finalize_args': abi.PoolFinalizeArgs,
             't3d_pointmlp_dgrad_args': abi.PointMlpDgradArgs, 't3d_pointmlp_wgrad_args': abi.PointMlpWgradArgs,
             't3d_bn_bwd_finalize_args': abi.BnBwdFinalizeArgs, 't3d_dy_colsum_args': abi.DyColsumArgs,
             't3d_fc_fwd_args': abi.FcFwdArgs, 't3d_fc_bwd_args': abi.FcBwdArgs, 't3d_fc_dinput_args': abi.FcDinputArgs,
             't3d_seg_head_args': abi.SegHeadArgs, 't3d_seg_finalize_args': abi.SegFinalizeArgs,
             't3d_strong_loss_args': abi.StrongLossArgs, 't3d_slab_desc': abi.SlabDesc, 't3d_schedule': abi.Schedule,
             't3d_strong_weights': abi.StrongWeights, 't3d_boxpc_rep_args': abi.BoxPcRepArgs,
             't3d_boxpc_rep_bwd_args': abi.BoxPcRepBwdArgs, 't3d_boxpc_loss_args': abi.BoxPcLossArgs,
             't3d_pointmlp_dgrad_narrow_args': abi.DgradNarrowArgs, 't3d_semi_final_loss_args': abi.SemiFinalLossArgs,
             't3d_anchor_reg_bwd_args': abi.AnchorRegBwdArgs, 't3d_pool_bwd_prep_args': abi.PoolBwdPrepArgs,
             't3d_pool_sparse_rows_args': abi.PoolSparseRowsArgs, 't3d_pointmlp_dgrad_gram_args': abi.PointMlpDgradGramArgs,
             't3d_pointmlp_gram_args': abi.PointMlpGramArgs, 't3d_act_colsum_args': abi.ActColsumArgs,
             't3d_pool_wgrad_finish_args': abi.PoolWgradFinishArgs, 't3d_box_refine_step_args': abi.BoxRefineStepArgs, 't3d_batch_assemble_args': abi.BatchAssembleArgs}
    for cname, cls in pairs.items():
        m = re.search(r'typedef struct \{([^}]*)\}\s*%s;' % cname, h)
        assert m, cname
        body = re.sub(r'/\*.*?\*/', '', m.group(1), flags=re.S)
        names = []
        for decl in body.split(';'):
            decl = decl.strip()
            if not decl:
                continue
            for part in decl.split(','):
                names.append(re.findall(r'(\w+)(?:\[\d+\])?\s*$', part.strip())[0])
        py = [f[0].rstrip('_') for f in cls._fields_]
        assert names == py, (cname, names, py)


def test_load_fails_loudly_without_the_library(tmp_path):
    import pytest
    with pytest.raises(abi.T3DError):
        abi.load(str(tmp_path / 'missing.so'))


SIZED = {'t3d_pointmlp_fwd_args': 'PointMlpFwdArgs', 't3d_pointmlp_dgrad_args': 'PointMlpDgradArgs', 't3d_pointmlp_wgrad_args': 'PointMlpWgradArgs',
         't3d_pointmlp_gram_args': 'PointMlpGramArgs', 't3d_pointmlp_dgrad_gram_args': 'PointMlpDgradGramArgs',
         't3d_seg_head_args': 'SegHeadArgs', 't3d_boxpc_rep_args': 'BoxPcRepArgs'}


def test_sized_structs_have_the_size_the_c_compiler_gives_them(tmp_path):
    """ABI version 2: the structs that have grown start with `struct_size`.  gcc's sizeof of every one of them (and of a few plain
    ones) equals the ctypes mirror's, and the mirror fills the field in by itself."""
    import ctypes as C
    import subprocess
    plain = {'t3d_act_src': 'ActSrc', 't3d_dy_src': 'DySrc', 't3d_fc_fwd_args': 'FcFwdArgs', 't3d_bn_fwd_finalize_args': 'BnFwdFinalizeArgs',
             't3d_rider_set': 'RiderSet', 't3d_small_op': 'SmallOp'}
    names = list(SIZED) + list(plain)
    src = tmp_path / 'sizes.c'
    src.write_text('#include <stdio.h>\n#include "t3d.h"\nint main(void) {\n' +
                   ''.join('  printf("%s %%zu\\n", sizeof(%s));\n' % (n, n) for n in names) + '  return 0;\n}\n')
    exe = tmp_path / 'sizes'
    subprocess.check_call(['gcc', '-I', os.path.join(ROOT, 'include'), str(src), '-o', str(exe)])
    got = dict(line.split() for line in subprocess.check_output([str(exe)], text=True).splitlines())
    for cname, pyname in {**SIZED, **plain}.items():
        cls = getattr(abi, pyname)
        assert int(got[cname]) == C.sizeof(cls), (cname, got[cname], C.sizeof(cls))
    for pyname in SIZED.values():
        cls = getattr(abi, pyname)
        assert cls._fields_[0][0] == 'struct_size' and cls().struct_size == C.sizeof(cls)


def test_a_struct_of_another_size_is_refused_before_anything_is_launched():
    """A caller built against an older header passes a shorter struct: every entry point that takes a sized struct answers T3D_ERR_ABI
    (no GPU needed: the check is the first thing a launcher does)."""
    import ctypes as C
    lib = abi.load()
    null = C.c_void_p(0)

    def short(cls):
        a = cls()
        a.struct_size -= 8
        return a
    f, d, w = short(abi.PointMlpFwdArgs), short(abi.PointMlpDgradArgs), short(abi.PointMlpWgradArgs)
    g, dg = short(abi.PointMlpGramArgs), short(abi.PointMlpDgradGramArgs)
    assert lib.t3d_pointmlp_fwd(C.byref(f), null) == abi.ERR_ABI
    assert lib.t3d_pointmlp_fwd_hosts_riders(C.byref(f)) == abi.ERR_ABI
    assert lib.t3d_pointmlp_dgrad(C.byref(d), null) == abi.ERR_ABI
    assert lib.t3d_pointmlp_wgrad(C.byref(w), null) == abi.ERR_ABI
    assert lib.t3d_pointmlp_bwd(C.byref(d), C.byref(abi.PointMlpWgradArgs()), null) == abi.ERR_ABI
    assert lib.t3d_pointmlp_bwd(C.byref(abi.PointMlpDgradArgs()), C.byref(w), null) in (abi.ERR_ABI, -1)      # (the intact one is checked first: empty)
    assert lib.t3d_pointmlp_gram(C.byref(g), null) == abi.ERR_ABI
    assert lib.t3d_pointmlp_dgrad_gram(C.byref(dg), null) == abi.ERR_ABI
    assert lib.t3d_seg_head(C.byref(short(abi.SegHeadArgs)), null) == abi.ERR_ABI
    assert lib.t3d_boxpc_rep(C.byref(short(abi.BoxPcRepArgs)), null) == abi.ERR_ABI
    # the right size gets past the check (and is then refused for its null pointers)
    assert lib.t3d_pointmlp_fwd(C.byref(abi.PointMlpFwdArgs()), null) == -1


def test_a_newer_caller_with_zeroed_unknown_fields_is_accepted_a_non_zero_one_refused():
    """struct_size > the library's sizeof: a caller built against a LATER header (fields appended).  Accepted while every byte the library
    does not know is 0 (the documented default of an appended field), T3D_ERR_ABI once one is not.  No GPU needed."""
    import ctypes as C
    lib = abi.load()
    null = C.c_void_p(0)
    for cls, fn in ((abi.PointMlpFwdArgs, lib.t3d_pointmlp_fwd), (abi.PointMlpDgradArgs, lib.t3d_pointmlp_dgrad),
                    (abi.PointMlpWgradArgs, lib.t3d_pointmlp_wgrad), (abi.PointMlpGramArgs, lib.t3d_pointmlp_gram),
                    (abi.PointMlpDgradGramArgs, lib.t3d_pointmlp_dgrad_gram), (abi.SegHeadArgs, lib.t3d_seg_head),
                    (abi.BoxPcRepArgs, lib.t3d_boxpc_rep)):
        n = C.sizeof(cls)
        buf = (C.c_ubyte * (n + 16))()
        C.memmove(buf, C.byref(cls()), n)
        C.cast(buf, C.POINTER(C.c_uint32))[0] = n + 16            # struct_size of the "newer" header
        assert fn(C.cast(buf, C.POINTER(cls)), null) == -1, cls   # past the ABI gate: refused for its null pointers (T3D_ERR_ARG)
        buf[n + 4] = 1                                            # a field this library does not know, in use
        assert fn(C.cast(buf, C.POINTER(cls)), null) == abi.ERR_ABI, cls


def test_an_older_callers_struct_is_completed_with_zeros(tmp_path):
    """csrc/abi_take.h against a struct that has GROWN: the library's declaration has one field more than the caller's.  The caller's
    bytes arrive intact, the field it does not know reads 0, a struct shorter than the version-2 size is refused, the same size is
    used in place (compiled with g++: the header has no HIP dependency)."""
    src = tmp_path / 't.cpp'
    src.write_text('''
#include <stdio.h>
#include "abi_take.h"
struct old_args { uint32_t struct_size; int M; const float* x; int K; };                 // the caller's header (version 2)
struct new_args { uint32_t struct_size; int M; const float* x; int K; long long appended; };   // the library's header: one field appended (past the padding)
static_assert(sizeof(new_args) > sizeof(old_args), "the struct grew");
static int entry(const new_args* a, int* seen_K, int* seen_appended, int* in_place) {
  const new_args* a0 = a;
  new_args local;
  const int e = t3d_abi_take(a, local, (uint32_t)sizeof(old_args));
  if (e) return e;
  *seen_K = a->K; *seen_appended = (int)a->appended; *in_place = (a == a0);
  return 0;
}
int main() {
  float x = 1.f;
  int K, ap, ip;
  old_args o = {(uint32_t)sizeof(old_args), 128, &x, 64};
  unsigned char buf[sizeof(new_args) + 8];
  memset(buf, 0xff, sizeof(buf));                       // whatever lies behind the caller's struct must not leak in
  memcpy(buf, &o, sizeof(o));
  int e = entry(reinterpret_cast<const new_args*>(buf), &K, &ap, &ip);
  if (e != 0 || K != 64 || ap != 0 || ip != 0) { printf("older caller: e=%d K=%d appended=%d in_place=%d\\n", e, K, ap, ip); return 1; }
  new_args n = {(uint32_t)sizeof(new_args), 128, &x, 64, 7};
  e = entry(&n, &K, &ap, &ip);
  if (e != 0 || K != 64 || ap != 7 || ip != 1) { printf("same header: e=%d\\n", e); return 2; }
  o.struct_size = (uint32_t)sizeof(old_args) - 4;
  memcpy(buf, &o, sizeof(o));
  if (entry(reinterpret_cast<const new_args*>(buf), &K, &ap, &ip) != -4) return 3;
  printf("ok\\n");
  return 0;
}
''')
    import subprocess
    csrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'transferable3d_amd', 'csrc')
    exe = str(tmp_path / 't')
    subprocess.check_call(['g++', '-std=c++17', '-O1', '-I', csrc, str(src), '-o', exe])
    assert subprocess.check_output([exe]).decode().strip() == 'ok'


def test_gemm_arithmetic_is_a_request_in_the_launch_struct_not_an_environment_variable(monkeypatch):
    """t3d_gemm_arithmetic: what a launch of this request and shape takes.  An explicit request ignores T3D_X3; only T3D_ARITH_AUTO
    (a zero-initialised struct) reads it."""
    lib = abi.load()
    for env in ('0', '1'):
        monkeypatch.setenv('T3D_X3', env)
        assert lib.t3d_gemm_arithmetic(abi.ARITH_BF16X3, abi.F32, 512, 256, 0) == abi.ARITH_BF16X3
        assert lib.t3d_gemm_arithmetic(abi.ARITH_FP32_MFMA, abi.F32, 512, 256, 0) == abi.ARITH_FP32_MFMA
        assert lib.t3d_gemm_arithmetic(abi.ARITH_BF16X3, abi.F32, 64, 512, 1) == abi.ARITH_BF16X3         # (round 6: the narrow-input backward too)
        assert lib.t3d_gemm_arithmetic(abi.ARITH_BF16X3, abi.F32, 192, 64, 1) == abi.ARITH_FP32_MFMA      # the launcher's own rule (K % 128)
        assert lib.t3d_gemm_arithmetic(abi.ARITH_BF16X3, abi.F32, 4, 64, 0) == abi.ARITH_FP32_MFMA        # no x3 kernel for K = 4
        assert lib.t3d_gemm_arithmetic(abi.ARITH_BF16X3, abi.BF16, 512, 256, 0) == abi.ARITH_BF16
        assert lib.t3d_gemm_arithmetic(abi.ARITH_AUTO, abi.F32, 512, 256, 0) == (abi.ARITH_BF16X3 if env == '1' else abi.ARITH_FP32_MFMA)
    import fake_t3d
    fake = fake_t3d.FakeLib()
    for req in (abi.ARITH_BF16X3, abi.ARITH_FP32_MFMA):
        for (K, N) in ((512, 256), (64, 512), (4, 64), (128, 128), (192, 64), (96, 512), (128, 1024), (8192, 64)):
            for kind in range(6):      # T3D_GEMM_FWD ... T3D_GEMM_DGRAD_GRAM: every launcher's own rule
                assert fake.t3d_gemm_arithmetic(req, abi.F32, K, N, kind) == lib.t3d_gemm_arithmetic(req, abi.F32, K, N, kind), (req, K, N, kind)
    assert lib.t3d_gemm_arithmetic(abi.ARITH_BF16X3, abi.F32, 192, 64, 2) == abi.ARITH_BF16X3      # a data gradient alone takes K = 192 ...
    assert lib.t3d_gemm_arithmetic(abi.ARITH_BF16X3, abi.F32, 192, 64, 1) == abi.ARITH_FP32_MFMA   # ... the fused backward does not
    assert lib.t3d_gemm_arithmetic(abi.ARITH_BF16X3, abi.F32, 192, 64, 9) < 0
