"""CPU: the TensorFlow checkpoint (tensor bundle) reader / writer of SURVEY 8f-1.  TensorFlow itself is absent, so the format
is pinned through published known answers of its building blocks and through round trips of whole model checkpoints."""
import os
import struct

import numpy as np
import pytest

from fake_t3d import FakeLib
from transferable3d_amd import tf_checkpoint as T
from transferable3d_amd.engine import Runtime


def test_crc32c_known_answers():
    # RFC 3720 (iSCSI) appendix B.4 check values and the customary "123456789" check
    assert T.crc32c(b'123456789') == 0xE3069283
    assert T.crc32c(bytes(32)) == 0x8A9136AA
    assert T.crc32c(b'\xff' * 32) == 0x62A8AB43
    assert T.crc32c(bytes(range(32))) == 0x46DD794E
    assert T.crc32c(bytes(range(31, -1, -1))) == 0x113FDB5C
    assert T.crc32c(b'') == 0


def test_crc32c_blocked_path_equals_bytewise():
    r = np.random.RandomState(0)
    for n in (4096, 4097, 65536 + 3, 300001):
        data = r.randint(0, 256, size=n, dtype=np.uint8).tobytes()
        reg = np.full(1, 0xffffffff, np.uint32)
        ref = int(T._update_lanes(reg, np.frombuffer(data, np.uint8).reshape(1, -1))[0]) ^ 0xffffffff
        assert T.crc32c(data) == ref, n


def test_crc_mask_is_the_leveldb_formula():
    # leveldb util/crc32c.h: Mask(crc) = ((crc >> 15) | (crc << 17)) + 0xa282ead8
    crc = T.crc32c(b'foo')
    assert T.mask_crc(crc) != crc and T.unmask_crc(T.mask_crc(crc)) == crc
    assert T.mask_crc(0) == 0xa282ead8
    assert T.mask_crc(0x00008000) == (0x00000001 + 0xa282ead8) & 0xffffffff


def test_varint_and_proto_encodings():
    assert T.put_varint(0) == b'\x00' and T.put_varint(300) == b'\xac\x02'      # protobuf encoding guide: 300 -> AC 02
    assert T.get_varint(b'\xac\x02', 0) == (300, 2)
    assert T.encode_header(1) == b'\x08\x01\x1a\x02\x08\x01'
    assert T.decode_header(T.encode_header(1)) == {'num_shards': 1, 'endianness': 0, 'producer': 1}
    e = T.encode_entry(1, (1, 1, 64, 128), 0, 1024, 32768, 0xdeadbeef)
    # dtype DT_FLOAT, shape {dim{size:1} dim{size:1} dim{size:64} dim{size:128}}, offset, size, fixed32 crc
    assert e == (b'\x08\x01' b'\x12\x11' b'\x12\x02\x08\x01' b'\x12\x02\x08\x01' b'\x12\x02\x08\x40' b'\x12\x03\x08\x80\x01'
                 b'\x20\x80\x08' b'\x28\x80\x80\x02' b'\x35' + struct.pack('<I', 0xdeadbeef))
    d = T.decode_entry(e)
    assert (d['dtype'], d['shape'], d['offset'], d['size'], d['crc32c']) == (1, [1, 1, 64, 128], 1024, 32768, 0xdeadbeef)
    scalar = T.encode_entry(3, (), 0, 0, 4, 1)
    assert scalar.startswith(b'\x08\x03\x12\x00') and T.decode_entry(scalar)['shape'] == []


def test_table_layout_and_multi_block_round_trip():
    items = [(b'', b'header')] + [(('scope/var_%04d/weights' % i).encode(), os.urandom(1 + i % 50)) for i in range(500)]
    for block_size in (64, 1000, T.BLOCK_SIZE):
        buf = T.build_table(items, block_size)
        assert struct.unpack('<Q', buf[-8:])[0] == 0xdb4775248b80fb57 and buf[-8:] == bytes.fromhex('57fb808b247547db')
        assert T.parse_table(buf) == items
    # a flipped bit inside a block is caught by the block trailer
    bad = bytearray(T.build_table(items, 1000))
    bad[10] ^= 1
    with pytest.raises(ValueError):
        T.parse_table(bytes(bad))
    with pytest.raises(ValueError):
        T.build_table([(b'b', b''), (b'a', b'')])
    # prefix compression really shares key bytes: the table is smaller than the raw keys + values
    assert len(T.build_table(items)) < sum(len(k) + len(v) for k, v in items)


def test_bundle_round_trip_and_state_file(tmp_path):
    r = np.random.RandomState(1)
    tensors = {'conv1/weights': r.randn(1, 4, 1, 64).astype(np.float32), 'conv1/biases': np.zeros(64, np.float32),
               'fc3/weights': r.randn(256, 67).astype(np.float32), 'Variable': np.asarray(1234, np.int32),
               'beta1_power': np.asarray(0.9 ** 5, np.float32), 'big/weights': r.randn(1088, 512).astype(np.float32)}
    p0 = T.write_checkpoint(str(tmp_path / 'model_epoch_0.ckpt'), tensors)
    p1 = T.write_checkpoint(str(tmp_path / 'model_epoch_1.ckpt'), tensors)
    assert sorted(os.listdir(tmp_path)) == ['checkpoint', 'model_epoch_0.ckpt.data-00000-of-00001', 'model_epoch_0.ckpt.index',
                                            'model_epoch_1.ckpt.data-00000-of-00001', 'model_epoch_1.ckpt.index']
    assert T.latest_checkpoint(str(tmp_path)) == p1
    assert open(tmp_path / 'checkpoint').read() == ('model_checkpoint_path: "model_epoch_1.ckpt"\n'
                                                    'all_model_checkpoint_paths: "model_epoch_0.ckpt"\n'
                                                    'all_model_checkpoint_paths: "model_epoch_1.ckpt"\n')
    back = T.read_checkpoint(p0)
    assert sorted(back) == sorted(tensors)
    for k, v in tensors.items():
        assert back[k].dtype == v.dtype and back[k].shape == v.shape and np.array_equal(back[k], v), k
    assert dict(T.list_variables(p0))['conv1/weights'] == [1, 4, 1, 64]
    # the data file is the tensors back to back in key order
    assert os.path.getsize(p0 + '.data-00000-of-00001') == sum(v.nbytes for v in tensors.values())
    # a Saver with a var_list reads a subset and reports a missing key the way restore does
    assert list(T.read_checkpoint(p0, names=['fc3/weights'])) == ['fc3/weights']
    with pytest.raises(KeyError):
        T.read_checkpoint(p0, names=['nope'])
    # corrupt one tensor byte: the per-tensor checksum trips
    with open(p0 + '.data-00000-of-00001', 'r+b') as f:
        f.seek(100)
        b = f.read(1)
        f.seek(100)
        f.write(bytes([b[0] ^ 0x40]))
    with pytest.raises(ValueError):
        T.read_checkpoint(p0)


def test_three_stage_hand_off_through_saver_bundles(tmp_path):
    """Recipe a -> b -> c of README.md:55-99 with checkpoints in the Saver's format: stage c restores `class_agnostic/` and
    `D_boxpc_branch/` from bundles saved without those prefixes (train_semisup_adv.py:224-237,450-467), and a restored run
    continues from the saved global step and Adam state."""
    from transferable3d_amd import train_boxpc, train_semisup, train_semisup_adv
    rt = lambda: Runtime(device='cpu', lib=FakeLib())
    quiet = lambda *_: None
    small = ['--num_point', '128', '--batch_size', '4', '--num_channels', '4', '--max_epoch', '1', '--steps_per_epoch', '2', '--synthetic',
             '--ckpt_format', 'tf']
    a_dir, b_dir, c_dir, r_dir = [str(tmp_path / d) for d in 'abcr']
    stage_a = ['--SEMI_MODEL', 'A', '--WEAK_WEIGHT_REPROJECTION', '0', '--WEAK_WEIGHT_SURFACE', '0']
    sd_a, _ = train_semisup.train(train_semisup.build_flags(stage_a + ['--log_dir', a_dir] + small), rt=rt(), log=quiet)
    sd_b, _ = train_boxpc.train(train_boxpc.build_flags(['--BOX_PC_MASK_REPRESENTATION', 'A', '--BOXPC_WEIGHT_DELTA', '4', '--log_dir', b_dir] + small),
                                rt=rt(), log=quiet)
    pa, pb = T.latest_checkpoint(a_dir), T.latest_checkpoint(b_dir)
    assert pa == os.path.join(a_dir, 'model_epoch_0.ckpt') and pb == os.path.join(b_dir, 'model_epoch_0.ckpt')
    ck = T.read_checkpoint(pa)
    assert int(ck['Variable']) == 2 and abs(float(ck['beta1_power']) - 0.9 ** 3) < 1e-7
    assert np.array_equal(ck['inst_seg/conv5/weights'], sd_a['inst_seg/conv5/weights'])
    assert ck['inst_seg/conv5/weights/Adam'].shape == (1, 1, 128, 1024) and np.abs(ck['inst_seg/conv5/weights/Adam_1']).max() > 0
    assert 'inst_seg/conv5/bn/moving_mean/Adam' not in ck                      # slots exist for trainable variables only
    logs = []
    flags_c = train_semisup_adv.build_flags(
        ['--SEMI_MODEL', 'F', '--BOX_PC_MASK_REPRESENTATION', 'A', '--use_one_hot', '--SEMI_BOXPC_FIT_ONLY_ON_2D_CLS', '1',
         '--WEAK_WEIGHT_INTRACLASSVAR', '2', '--WEAK_WEIGHT_REPROJECTION', '0', '--SEMI_MULTIPLIER_FOR_WEAK_LOSS', '0.05',
         '--SUNRGBD_SEMI_TEST_CLS', 'table', 'sofa', 'dresser', 'night_stand', 'bookshelf', '--init_class_ag_path', pa,
         '--init_boxpc_path', pb, '--log_dir', c_dir] + small)
    sd_c, loss_c = train_semisup_adv.train(flags_c, rt=rt(), log=logs.append)
    assert 'restored %d class_agnostic variables' % len(sd_a) in logs, logs[:3]
    assert 'restored %d D_boxpc_branch variables' % len(sd_b) in logs, logs[:3]
    k = 'D_boxpc_branch/box_pc_mask_model/fc1/weights'                 # frozen branch: exactly stage b's weights
    assert np.isfinite(loss_c) and np.array_equal(sd_c[k], sd_b['box_pc_mask_model/fc1/weights'])
    names = dict(T.list_variables(T.latest_checkpoint(c_dir)))
    assert 'class_dependent/box_refine/fc0/weights/Adam' in names and k + '/Adam' not in names   # slots only for the var_list
    # restore stage a's bundle into a fresh graph: it continues from the saved global step with the saved Adam state
    flags_r = train_semisup.build_flags(stage_a + ['--log_dir', r_dir, '--restore_model_path', pa] + small)
    train_semisup.train(flags_r, rt=rt(), log=quiet)
    ck2 = T.read_checkpoint(T.latest_checkpoint(r_dir))
    assert int(ck2['Variable']) == 4
    # ... and trained on
    assert not np.array_equal(ck2['inst_seg/conv1/weights'], ck['inst_seg/conv1/weights'])


def test_saver_keeps_the_last_five_checkpoints_it_wrote(tmp_path):
    """tf.train.Saver(max_to_keep=5) of the drivers (train_semisup.py:259): the sixth save removes the first, in both formats; files the
    saver did not write stay; the TensorFlow state file lists the kept prefixes."""
    import os
    import types
    import numpy as np
    from transferable3d_amd import tf_checkpoint as T

    class Vars:
        index = {'a/weights': (0, (2, 3), True), 'a/bn/moving_mean': (0, (3,), False)}
        adam_m = adam_v = None

        def state_dict(self):
            return {'a/weights': np.arange(6, dtype=np.float32).reshape(2, 3), 'a/bn/moving_mean': np.zeros(3, np.float32)}

    g = types.SimpleNamespace(vars=Vars(), hyper=np.array([7.0], np.float32))
    d = str(tmp_path)
    open(os.path.join(d, 'model_epoch_99.npz'), 'wb').close()           # somebody else's file
    s = T.Saver(max_to_keep=5)
    for epoch in range(8):
        p = s.save(d, epoch, g, 'npz')
        assert os.path.exists(p)
    left = sorted(f for f in os.listdir(d) if f.endswith('.npz'))
    assert left == ['model_epoch_%d.npz' % e for e in (3, 4, 5, 6, 7)] + ['model_epoch_99.npz'], left
    assert [os.path.basename(p) for p in s.last_checkpoints] == ['model_epoch_%d.npz' % e for e in (3, 4, 5, 6, 7)]
    # TensorFlow bundles: index + data shard removed together, state file rewritten
    d2 = tmp_path / 'tf'
    d2.mkdir()
    real_sv = T.saver_variables
    T.saver_variables = lambda vars_, step, optimizer_scopes=None: dict(vars_.state_dict(), Variable=np.asarray(step, np.int32))
    try:
        s2 = T.Saver(max_to_keep=2)
        for epoch in range(4):
            s2.save(str(d2), epoch, g, 'tf')
    finally:
        T.saver_variables = real_sv
    names = sorted(os.listdir(str(d2)))
    assert names == ['checkpoint', 'model_epoch_2.ckpt.data-00000-of-00001', 'model_epoch_2.ckpt.index',
                     'model_epoch_3.ckpt.data-00000-of-00001', 'model_epoch_3.ckpt.index'], names
    assert T.latest_checkpoint(str(d2)).endswith('model_epoch_3.ckpt')
    assert T._state_paths(os.path.join(str(d2), 'checkpoint')) == ['model_epoch_2.ckpt', 'model_epoch_3.ckpt']
    assert T.read_checkpoint(T.latest_checkpoint(str(d2)))['a/weights'].shape == (2, 3)
    # max_to_keep=None keeps everything
    s3 = T.Saver(max_to_keep=None)
    d3 = tmp_path / 'all'
    d3.mkdir()
    for epoch in range(7):
        s3.save(str(d3), epoch, g, 'npz')
    assert len(os.listdir(str(d3))) == 7
