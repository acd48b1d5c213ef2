"""CPU: the build-time ISA gate of the x3 main loops (transferable3d_amd/build.py:check_x3_isa, tools/isa_loops.py).  The analyser is
checked on a synthetic listing (a loop it must pass and the four defects it must name); the gate itself on the device assembly the build
of this tree left behind (build/ travels with the snapshot; a tree that was never built here skips that part)."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tools'))
import isa_loops      # noqa: E402
from transferable3d_amd import build as B      # noqa: E402

HEAD = '_ZN12_GLOBAL__N_114k_pointmlp_fwdILi128ELb0ENS_6PathX3EfEEv21t3d_pointmlp_fwd_args: ; @_ZN12_GLOBAL__N_114k_pointmlp_fwdILi128ELb0ENS_6PathX3EfEEv21t3d_pointmlp_fwd_args\n'
MFMA = '\tv_mfma_f32_32x32x16_bf16 v[0:15], v[16:19], v[20:23], v[0:15]\n'


def listing(body):
    return (HEAD + '; %bb.0:\n\ts_load_dword s0, s[0:1], 0x0\n.LBB0_1:\n' + body +
            '\ts_waitcnt lgkmcnt(0)\n\ts_barrier\n\tds_read_b128 v[16:19], v1\n\ts_cbranch_scc1 .LBB0_1\n; %bb.2:\n\ts_endpgm\n.Lfunc_end0:\n')


def rows_of(body):
    return isa_loops.report(listing(body), ('PathX3E',))


def test_the_analyser_maps_a_hand_placed_loop_and_names_each_defect():
    good = ''.join(MFMA + '\tv_fma_f32 v30, v31, v32, v33\n' * 5 + '\tds_write_b64 v1, v[2:3]\n' for _ in range(8)) + '\tglobal_load_dwordx4 v[40:43], v[44:45], off\n'
    r, = rows_of(good)
    assert (r['mfma'], r['valu'], r['mfma_burst'], r['valu_run'], r['pk_f32'], r['vmcnt0'], r['flat'], r['barriers']) == (8, 40, 1, 5, 0, 0, 0, 1)
    assert not isa_loops.check([r], max_valu_run=8, max_burst=2)
    burst, = rows_of(MFMA * 6 + '\tv_fma_f32 v30, v31, v32, v33\n' * 30 + '\tds_write_b64 v1, v[2:3]\n')
    assert burst['mfma_burst'] == 6 and burst['valu_run'] == 30 and isa_loops.check([burst], 8, 2)
    packed, = rows_of(''.join(MFMA + '\tv_pk_fma_f32 v[30:31], v[32:33], v[34:35], v[36:37]\n' for _ in range(6)))
    assert packed['pk_f32'] == 6
    drained, = rows_of(''.join(MFMA + '\tv_fma_f32 v30, v31, v32, v33\n' for _ in range(6)) + '\tglobal_load_dwordx4 v[40:43], v[44:45], off\n\ts_waitcnt vmcnt(0)\n')
    assert drained['vmcnt0'] == 1
    flat, = rows_of(''.join(MFMA + '\tv_fma_f32 v30, v31, v32, v33\n' for _ in range(6)) + '\tflat_load_dwordx4 v[40:43], v[44:45]\n')
    assert flat['flat'] == 1
    assert isa_loops.short(HEAD.split(':')[0]) == 'k_pointmlp_fwd<128,false,PathX3,float>'


def test_the_gate_passes_on_the_assembly_of_this_build_and_fails_on_a_regrouped_loop(tmp_path):
    asm = os.path.join(ROOT, 'transferable3d_amd', 'build', 'pointmlp_x3-hip-amdgcn-amd-amdhsa-gfx950.s')
    if not os.path.exists(asm):
        pytest.skip('no device assembly in this tree (build() writes it)')
    rows = B.check_x3_isa(asm)
    assert len(rows) >= 40                                           # every x3 main loop of the default program
    assert all(r['mfma_burst'] <= B.ISA_MAX_BURST and r['pk_f32'] == 0 and r['vmcnt0'] == 0 and r['flat'] == 0 for r in rows)
    assert max(r['valu_run'] for r in rows if r['mfma'] >= 48) <= B.ISA_MAX_VALU_RUN_WIDE
    bad = tmp_path / 'regrouped.s'
    bad.write_text(listing(MFMA * 12 + '\tv_fma_f32 v30, v31, v32, v33\n' * 60 + '\tds_write_b64 v1, v[2:3]\n'))
    with pytest.raises(RuntimeError, match='MFMA burst 12'):
        B.check_x3_isa(str(bad))
