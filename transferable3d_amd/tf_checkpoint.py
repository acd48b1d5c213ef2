"""TensorFlow V2 checkpoint ("tensor bundle") reader / writer without TensorFlow.

The reference saves and restores with `tf.train.Saver` (train_semisup.py:259,279,317; train_boxpc.py:261,280,313;
train_semisup_adv.py:433,475,514; test_semisup.py:158-159) and hands stage-a / stage-b weights to stage c by rewriting scope
prefixes (train_semisup_adv.py:224-237,450-467).  A Saver checkpoint `<prefix>` is

  <prefix>.index                   an SSTable (the LevelDB table format) : key "" -> BundleHeaderProto,
                                   key <variable name> -> BundleEntryProto {dtype, shape, shard_id, offset, size, crc32c}
  <prefix>.data-00000-of-00001     the tensors' bytes back to back in key order, little endian
  checkpoint                       text proto naming the latest prefix (tf.train.latest_checkpoint)

TensorFlow is a third-party dependency of the reference that is absent from this image and from /root/reference (its
version is not pinned there either), so this module restates the published on-disk format: LevelDB table blocks (shared-prefix
key compression, restart array, 1-byte type + masked CRC-32C trailer, 48-byte footer with magic 0xdb4775248b80fb57) and the
proto3 wire encoding of the two bundle messages.  The state-dict keys of this package are already the reference's variable
names with TensorFlow's shapes, so reading a checkpoint gives a state dict and writing one takes a state
dict.  tests/test_tf_checkpoint.py pins the pieces that have published known answers (CRC-32C check values, the masked-CRC
formula, varint / proto encodings, the footer layout) and round-trips whole model checkpoints.
"""
import os
import struct

import numpy as np

TABLE_MAGIC = 0xdb4775248b80fb57
FOOTER_LEN = 48
RESTART_INTERVAL = 16
BLOCK_SIZE = 256 * 1024             # table::Options().block_size used by the bundle writer
MASK_DELTA = 0xa282ead8

# DataType enum (types.proto) <-> numpy
_DTYPES = {1: np.float32, 2: np.float64, 3: np.int32, 4: np.uint8, 5: np.int16, 6: np.int8, 9: np.int64, 10: np.bool_}
_DTYPE_IDS = {np.dtype(v): k for k, v in _DTYPES.items()}


# ---- CRC-32C (Castagnoli, reflected polynomial 0x82F63B78) ---------------------------------------------------------------
def _make_table():
    t = np.zeros(256, np.uint32)
    for i in range(256):
        c = i
        for _ in range(8):
            c = (c >> 1) ^ (0x82F63B78 if c & 1 else 0)
        t[i] = c
    return t


_CRC_TABLE = _make_table()
_SHIFT_TABLES = {}


def _update_lanes(reg, blocks):
    """Advance one CRC register per row of `blocks` ([lanes, L] uint8) by that row's bytes."""
    for j in range(blocks.shape[1]):
        reg = _CRC_TABLE[(reg ^ blocks[:, j]) & 0xff] ^ (reg >> np.uint32(8))
    return reg


def _shift_table(L):
    """Tables of the linear map reg -> register after L zero bytes, split by the byte of reg."""
    if L not in _SHIFT_TABLES:
        basis = (np.uint32(1) << np.arange(32, dtype=np.uint32)).astype(np.uint32)
        img = _update_lanes(basis, np.zeros((32, L), np.uint8))
        tabs = np.zeros((4, 256), np.uint32)
        for byte in range(4):
            for bit in range(8):
                sel = (np.arange(256) >> bit) & 1
                tabs[byte] ^= np.where(sel == 1, img[byte * 8 + bit], 0).astype(np.uint32)
        _SHIFT_TABLES[L] = tabs
    return _SHIFT_TABLES[L]


def crc32c(data):
    """CRC-32C of a bytes-like object.  Long inputs are cut into equal blocks whose registers advance together as NumPy
    lanes; the per-block registers are then chained with the zero-shift operator (CRC is linear over GF(2))."""
    buf = np.frombuffer(bytes(data) if not isinstance(data, (bytes, bytearray, memoryview, np.ndarray)) else data, np.uint8)
    n = buf.size
    if n < 4096:
        reg = np.full(1, 0xffffffff, np.uint32)
        reg = _update_lanes(reg, buf.reshape(1, n)) if n else reg
        return int(reg[0]) ^ 0xffffffff
    L = 1 << max(6, int(np.log2(np.sqrt(n))))
    pad = (-n) % L
    work = np.zeros(n + pad, np.uint8)
    work[pad:] = buf
    work[pad:pad + 4] ^= 0xff                 # the all-ones initial register == complementing the first four bytes
    regs = _update_lanes(np.zeros((n + pad) // L, np.uint32), work.reshape(-1, L))   # leading zero bytes leave a 0 register at 0
    tabs = _shift_table(L)
    reg = 0
    for r in regs.tolist():
        reg = int(tabs[0][reg & 0xff] ^ tabs[1][(reg >> 8) & 0xff] ^ tabs[2][(reg >> 16) & 0xff] ^ tabs[3][reg >> 24]) ^ r
    return reg ^ 0xffffffff


def mask_crc(crc):
    """crc32c::Mask — stored CRCs are rotated right by 15 and offset so that a CRC of data containing CRCs stays sound."""
    return (((crc >> 15) | (crc << 17)) + MASK_DELTA) & 0xffffffff


def unmask_crc(masked):
    rot = (masked - MASK_DELTA) & 0xffffffff
    return ((rot >> 17) | (rot << 15)) & 0xffffffff


# ---- varints and the two protos -------------------------------------------------------------------------------------------
def put_varint(v):
    v &= (1 << 64) - 1
    out = bytearray()
    while v >= 0x80:
        out.append((v & 0x7f) | 0x80)
        v >>= 7
    out.append(v)
    return bytes(out)


def get_varint(buf, pos):
    shift = v = 0
    while True:
        b = buf[pos]
        pos += 1
        v |= (b & 0x7f) << shift
        if b < 0x80:
            return v, pos
        shift += 7
        if shift > 63:
            raise ValueError('varint too long')


def _parse_fields(buf):
    """proto wire format -> list of (field number, wire type, value)."""
    pos, out = 0, []
    while pos < len(buf):
        key, pos = get_varint(buf, pos)
        f, wt = key >> 3, key & 7
        if wt == 0:
            v, pos = get_varint(buf, pos)
        elif wt == 1:
            v, pos = struct.unpack_from('<Q', buf, pos)[0], pos + 8
        elif wt == 2:
            n, pos = get_varint(buf, pos)
            v, pos = bytes(buf[pos:pos + n]), pos + n
        elif wt == 5:
            v, pos = struct.unpack_from('<I', buf, pos)[0], pos + 4
        else:
            raise ValueError('unsupported wire type %d' % wt)
        out.append((f, wt, v))
    return out


def encode_header(num_shards=1):
    """BundleHeaderProto {num_shards=1; endianness=2 (LITTLE=0, omitted); version=3 {producer=1}} (tensor_bundle.proto)."""
    return b'\x08' + put_varint(num_shards) + b'\x1a\x02\x08\x01'


def decode_header(buf):
    h = {'num_shards': 0, 'endianness': 0, 'producer': 0}
    for f, _, v in _parse_fields(buf):
        if f == 1:
            h['num_shards'] = v
        elif f == 2:
            h['endianness'] = v
        elif f == 3:
            for g, _, w in _parse_fields(v):
                if g == 1:
                    h['producer'] = w
    return h


def encode_entry(dtype_id, shape, shard_id, offset, size, crc_masked):
    """BundleEntryProto {dtype=1; shape=2 {dim=2 {size=1}}; shard_id=3; offset=4; size=5; crc32c=6 (fixed32)}; proto3 leaves
    zero scalars out."""
    dims = b''.join(b'\x12' + put_varint(len(d)) + d for d in ((b'\x08' + put_varint(s)) if s else b'' for s in shape))
    out = b'\x08' + put_varint(dtype_id) + b'\x12' + put_varint(len(dims)) + dims
    if shard_id:
        out += b'\x18' + put_varint(shard_id)
    if offset:
        out += b'\x20' + put_varint(offset)
    if size:
        out += b'\x28' + put_varint(size)
    if crc_masked:
        out += b'\x35' + struct.pack('<I', crc_masked)
    return out


def decode_entry(buf):
    e = {'dtype': 0, 'shape': [], 'shard_id': 0, 'offset': 0, 'size': 0, 'crc32c': 0, 'slices': 0}
    for f, _, v in _parse_fields(buf):
        if f == 1:
            e['dtype'] = v
        elif f == 2:
            for g, _, w in _parse_fields(v):
                if g == 2:
                    size = 0
                    for h, _, x in _parse_fields(w):
                        if h == 1:
                            size = x if x < (1 << 63) else x - (1 << 64)
                    e['shape'].append(size)
                elif g == 3 and w:
                    raise ValueError('tensor of unknown rank in a checkpoint')
        elif f == 3:
            e['shard_id'] = v
        elif f == 4:
            e['offset'] = v
        elif f == 5:
            e['size'] = v
        elif f == 6:
            e['crc32c'] = v
        elif f == 7:
            e['slices'] += 1
    return e


# ---- LevelDB table ----------------------------------------------------------------------------------------------------------
class _BlockBuilder:
    def __init__(self, restart_interval):
        self.ri = restart_interval
        self.buf = bytearray()
        self.restarts = [0]
        self.count = 0
        self.last = b''

    def add(self, key, value):
        shared = 0
        if self.count % self.ri == 0 and self.count:
            self.restarts.append(len(self.buf))
        elif self.count:
            m = min(len(key), len(self.last))
            while shared < m and key[shared] == self.last[shared]:
                shared += 1
        self.buf += put_varint(shared) + put_varint(len(key) - shared) + put_varint(len(value)) + key[shared:] + value
        self.last = key
        self.count += 1

    def size(self):
        return len(self.buf) + 4 * (len(self.restarts) + 1)

    def finish(self):
        return bytes(self.buf) + b''.join(struct.pack('<I', r) for r in self.restarts) + struct.pack('<I', len(self.restarts))


def _emit_block(out, contents):
    """Append block + trailer (type 0 = uncompressed, masked CRC over contents + type); returns the BlockHandle bytes."""
    handle = put_varint(len(out)) + put_varint(len(contents))
    out += contents + b'\x00' + struct.pack('<I', mask_crc(crc32c(contents + b'\x00')))
    return handle


def build_table(items, block_size=BLOCK_SIZE):
    """Sorted (key, value) byte pairs -> the bytes of an SSTable."""
    out = bytearray()
    index = _BlockBuilder(1)
    blk = _BlockBuilder(RESTART_INTERVAL)
    prev = None
    for key, value in items:
        if prev is not None and key <= prev:
            raise ValueError('table keys must be strictly increasing')
        if blk.count and blk.size() >= block_size:
            index.add(blk.last, _emit_block(out, blk.finish()))      # the block's own last key is a valid separator
            blk = _BlockBuilder(RESTART_INTERVAL)
        blk.add(key, value)
        prev = key
    if blk.count:
        index.add(blk.last, _emit_block(out, blk.finish()))
    meta_handle = _emit_block(out, _BlockBuilder(RESTART_INTERVAL).finish())
    index_handle = _emit_block(out, index.finish())
    footer = meta_handle + index_handle
    out += footer + b'\x00' * (FOOTER_LEN - 8 - len(footer)) + struct.pack('<Q', TABLE_MAGIC)
    return bytes(out)


def _read_block(buf, handle_bytes, pos=0, verify=True):
    off, pos = get_varint(handle_bytes, pos)
    size, pos = get_varint(handle_bytes, pos)
    contents = buf[off:off + size]
    btype = buf[off + size]
    if verify:
        stored = struct.unpack_from('<I', buf, off + size + 1)[0]
        if unmask_crc(stored) != crc32c(bytes(contents) + bytes([btype])):
            raise ValueError('checkpoint index: block checksum mismatch at offset %d' % off)
    if btype != 0:
        raise NotImplementedError('compressed table block (type %d); the bundle writer stores uncompressed blocks' % btype)
    return contents, pos


def _block_entries(contents):
    nrestart = struct.unpack_from('<I', contents, len(contents) - 4)[0]
    end = len(contents) - 4 * (nrestart + 1)
    pos, key = 0, b''
    while pos < end:
        shared, pos = get_varint(contents, pos)
        non_shared, pos = get_varint(contents, pos)
        vlen, pos = get_varint(contents, pos)
        key = key[:shared] + bytes(contents[pos:pos + non_shared])
        pos += non_shared
        yield key, bytes(contents[pos:pos + vlen])
        pos += vlen


def parse_table(buf, verify=True):
    """Bytes of an SSTable -> list of (key, value) in table order."""
    if len(buf) < FOOTER_LEN or struct.unpack_from('<Q', buf, len(buf) - 8)[0] != TABLE_MAGIC:
        raise ValueError('not a TensorFlow checkpoint index (bad table magic)')
    footer = buf[len(buf) - FOOTER_LEN:]
    _, pos = _read_block(buf, footer, 0, verify)             # metaindex (unused)
    index, _ = _read_block(buf, footer, pos, verify)
    out = []
    for _, handle in _block_entries(index):
        out.extend(_block_entries(_read_block(buf, handle, 0, verify)[0]))
    return out


# ---- bundle ---------------------------------------------------------------------------------------------------------------
def _data_path(prefix, shard, num_shards):
    return '%s.data-%05d-of-%05d' % (prefix, shard, num_shards)


def write_checkpoint(prefix, tensors, update_state_file=True):
    """Write {name: array} as `<prefix>.index` + `<prefix>.data-00000-of-00001` (what `saver.save(sess, prefix)` leaves) and,
    like the Saver, point the directory's `checkpoint` state file at it."""
    items = [(b'', encode_header(1))]
    offset = 0
    with open(_data_path(prefix, 0, 1) + '.tmp', 'wb') as f:
        for name in sorted(tensors, key=lambda s: s.encode()):
            a = np.asarray(tensors[name])
            if a.dtype not in _DTYPE_IDS:
                raise TypeError('%s: dtype %s has no checkpoint encoding here' % (name, a.dtype))
            raw = a.astype(a.dtype.newbyteorder('<'), copy=False).tobytes()
            f.write(raw)
            items.append((name.encode(), encode_entry(_DTYPE_IDS[a.dtype], a.shape, 0, offset, len(raw), mask_crc(crc32c(raw)))))
            offset += len(raw)
    os.replace(_data_path(prefix, 0, 1) + '.tmp', _data_path(prefix, 0, 1))
    with open(prefix + '.index.tmp', 'wb') as f:
        f.write(build_table(items))
    os.replace(prefix + '.index.tmp', prefix + '.index')
    if update_state_file:
        d, base = os.path.split(prefix)
        state = os.path.join(d, 'checkpoint')
        older = [p for p in _state_paths(state) if p != base]
        with open(state, 'w') as f:
            f.write('model_checkpoint_path: "%s"\n' % base)
            for p in older + [base]:
                f.write('all_model_checkpoint_paths: "%s"\n' % p)
    return prefix


def _state_paths(state_file):
    if not os.path.exists(state_file):
        return []
    out = []
    for line in open(state_file):
        if line.startswith('all_model_checkpoint_paths:'):
            out.append(line.split('"')[1])
    return out


def latest_checkpoint(directory):
    """tf.train.latest_checkpoint: the prefix named by `<directory>/checkpoint`, or None."""
    state = os.path.join(directory, 'checkpoint')
    if not os.path.exists(state):
        return None
    for line in open(state):
        if line.startswith('model_checkpoint_path:'):
            p = line.split('"')[1]
            return p if os.path.isabs(p) else os.path.join(directory, p)
    return None


def list_variables(prefix):
    """tf.train.list_variables: [(name, shape)] in key order."""
    with open(prefix + '.index', 'rb') as f:
        kv = parse_table(f.read())
    return [(k.decode(), decode_entry(v)['shape']) for k, v in kv if k]


def read_checkpoint(prefix, names=None, verify=True):
    """`<prefix>` -> {name: array}.  `names` restricts the read (a Saver built with var_list); checksums are verified."""
    with open(prefix + '.index', 'rb') as f:
        kv = parse_table(f.read(), verify)
    if not kv or kv[0][0] != b'':
        raise ValueError('checkpoint index without a bundle header')
    header = decode_header(kv[0][1])
    if header['endianness'] != 0:
        raise NotImplementedError('big-endian checkpoint')
    shards = {}
    out = {}
    for k, v in kv[1:]:
        name = k.decode()
        if names is not None and name not in names:
            continue
        e = decode_entry(v)
        if e['slices']:
            raise NotImplementedError('%s is a partitioned variable (tensor slices)' % name)
        if e['dtype'] not in _DTYPES:
            raise NotImplementedError('%s: checkpoint dtype %d' % (name, e['dtype']))
        if e['shard_id'] not in shards:
            shards[e['shard_id']] = np.memmap(_data_path(prefix, e['shard_id'], header['num_shards']), np.uint8, 'r')
        raw = np.asarray(shards[e['shard_id']][e['offset']:e['offset'] + e['size']])
        dt = np.dtype(_DTYPES[e['dtype']])
        if raw.size != int(np.prod(e['shape'], dtype=np.int64)) * dt.itemsize:
            raise ValueError('%s: %d bytes for shape %s' % (name, raw.size, e['shape']))
        if verify and unmask_crc(e['crc32c']) != crc32c(raw):
            raise ValueError('%s: tensor checksum mismatch' % name)
        out[name] = raw.view(dt.newbyteorder('<')).astype(dt).reshape(e['shape'])
    if names is not None:
        missing = [n for n in names if n not in out]
        if missing:                                   # Saver.restore: NotFoundError "Key ... not found in checkpoint"
            raise KeyError('Key %s not found in checkpoint %s' % (missing[0], prefix))
    return out


def is_checkpoint(path):
    return os.path.exists(path + '.index')


# ---- the Saver's view of a VarStore ----------------------------------------------------------------------------------------
def saver_variables(vars_, global_step, beta1=0.9, beta2=0.999, step_names=('Variable',), optimizer_scopes=None):
    """Everything `tf.train.Saver()` (all global variables) writes for this graph: the model variables, the global step(s)
    `Variable[_1]` (train_semisup.py:214,217), and the Adam state `beta1_power`, `beta2_power`, `<var>/Adam`, `<var>/Adam_1`
    (tf.train.AdamOptimizer slot naming; beta powers are beta^(t+1) after t updates)."""
    out = vars_.state_dict()
    t = int(global_step)
    for n in step_names:
        out[n] = np.asarray(t, np.int32)
    momentum = getattr(vars_, 'optimizer_kind', 'adam') == 'momentum'      # tf.train.MomentumOptimizer: one slot, `<var>/Momentum`
    if not momentum:
        out['beta1_power'] = np.asarray(beta1 ** (t + 1), np.float32)
        out['beta2_power'] = np.asarray(beta2 ** (t + 1), np.float32)
    for name, (off, shape, trainable) in vars_.index.items():
        if not trainable or (optimizer_scopes is not None and not any(name.startswith(p) for p in optimizer_scopes)):
            continue
        n = int(np.prod(shape))
        if momentum:
            out[name + '/Momentum'] = vars_.adam_m[off:off + n].detach().cpu().numpy().reshape(shape).copy()
            continue
        out[name + '/Adam'] = vars_.adam_m[off:off + n].detach().cpu().numpy().reshape(shape).copy()
        out[name + '/Adam_1'] = vars_.adam_v[off:off + n].detach().cpu().numpy().reshape(shape).copy()
    return out


def restore_variables(vars_, tensors, strict=False):
    """`saver.restore`: model variables by name and the Adam slots when the checkpoint holds them.  Returns (number of model
    variables restored, global step or None) — the step lives in the graph's schedule state, which the caller owns."""
    import torch
    model = {k: v for k, v in tensors.items() if k in vars_.index}
    if strict:
        missing = [k for k in vars_.index if k not in tensors]
        if missing:
            raise KeyError('Key %s not found in checkpoint' % missing[0])
    vars_.load_state_dict(model)
    for name, (off, shape, trainable) in vars_.index.items():
        n = int(np.prod(shape))
        if trainable and name + '/Adam' in tensors:
            vars_.adam_m[off:off + n].copy_(torch.as_tensor(np.asarray(tensors[name + '/Adam'], np.float32)).reshape(-1))
            vars_.adam_v[off:off + n].copy_(torch.as_tensor(np.asarray(tensors[name + '/Adam_1'], np.float32)).reshape(-1))
        if trainable and name + '/Momentum' in tensors:
            vars_.adam_m[off:off + n].copy_(torch.as_tensor(np.asarray(tensors[name + '/Momentum'], np.float32)).reshape(-1))
    return len(model), (int(tensors['Variable']) if 'Variable' in tensors else None)


# ---- what the command lines call ---------------------------------------------------------------------------------------------
def load_state(path):
    """A checkpoint named the way the reference's flags name it (`log/model_epoch_30.ckpt`, a Saver prefix) or this package's
    `.npz` state dict -> {name: array}."""
    if path.endswith('.npz'):
        return dict(np.load(path))
    if not is_checkpoint(path):
        raise FileNotFoundError('%s: neither an .npz state dict nor a TensorFlow checkpoint prefix (%s.index)' % (path, path))
    return read_checkpoint(path)


def save_model(log_dir, epoch, graph, fmt='npz', optimizer_scopes=None):
    """`saver.save(sess, LOG_DIR/model_epoch_<n>.ckpt)` (train_semisup.py:317): fmt 'tf' writes the Saver's bundle with the
    optimiser state and global step; 'npz' the plain state dict."""
    if fmt == 'tf':
        step = int(round(float(graph.hyper[0])))
        return write_checkpoint(os.path.join(log_dir, 'model_epoch_%d.ckpt' % epoch),
                                saver_variables(graph.vars, step, optimizer_scopes=optimizer_scopes))
    path = os.path.join(log_dir, 'model_epoch_%d.npz' % epoch)
    np.savez(path, **graph.vars.state_dict())
    return path


class Saver:
    """`tf.train.Saver(max_to_keep=5)` of the three drivers (train_semisup.py:259, train_boxpc.py:261, train_semisup_adv.py:433): `save`
    writes `LOG_DIR/model_epoch_<n>` like `save_model` and then removes the oldest checkpoint THIS saver wrote once it holds more than
    `max_to_keep` (files that were in the directory before are never touched, as in TensorFlow); a TensorFlow-format `checkpoint`
    state file lists the kept prefixes only.  `max_to_keep` None or 0: keep everything."""

    def __init__(self, max_to_keep=5):
        self.max_to_keep = max_to_keep
        self.last_checkpoints = []

    def save(self, log_dir, epoch, graph, fmt='npz', optimizer_scopes=None):
        path = save_model(log_dir, epoch, graph, fmt, optimizer_scopes=optimizer_scopes)
        if path in self.last_checkpoints:          # the same epoch saved again: it becomes the newest
            self.last_checkpoints.remove(path)
        self.last_checkpoints.append(path)
        while self.max_to_keep and len(self.last_checkpoints) > self.max_to_keep:
            self._delete(self.last_checkpoints.pop(0))
        if fmt == 'tf':
            state = os.path.join(log_dir, 'checkpoint')
            with open(state, 'w') as f:
                f.write('model_checkpoint_path: "%s"\n' % os.path.basename(path))
                for p in self.last_checkpoints:
                    f.write('all_model_checkpoint_paths: "%s"\n' % os.path.basename(p))
        return path

    @staticmethod
    def _delete(path):
        import glob
        victims = [path] if path.endswith('.npz') else [path + '.index'] + glob.glob(glob.escape(path) + '.data-*')
        for v in victims:
            try:
                os.remove(v)
            except FileNotFoundError:
                pass


def restore_model(graph, path):
    """`saver.restore(sess, path)`: weights, moving statistics and — from a Saver bundle — Adam slots and the global step."""
    n, step = restore_variables(graph.vars, load_state(path))
    if step is not None:
        graph.hyper[0] = float(step)
    return n
