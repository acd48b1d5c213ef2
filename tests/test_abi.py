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
    assert lib.t3d_abi_version() == 1


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
