"""Reader / writer for TensorFlow "tensor bundle" checkpoints (``model.ckpt-N.index`` +
``model.ckpt-N.data-00000-of-00001``), written from the published formats -- no TensorFlow needed.

Why: the reference publishes its trained vocoders as TF1 checkpoints
(/root/reference/README.md:190-195) and restores them by variable name
(scripts/spectrogram_advoc.py:55-64, models/advoc/train_evaluate.py:147-165).  This build's
parameters carry the same TF variable names and layouts (advoc_amd/model.py), so importing such a
checkpoint is a matter of reading the container.

STATUS: **untested against a file produced by TensorFlow** -- none exists on the build or GPU boxes
and TensorFlow cannot be installed there.  The reader follows the format documents below and is
tested against (a) this module's own writer, (b) the known-answer vectors of crc32c and of the
table layout.  Treat a failure on a real checkpoint as a bug in this file, not in the checkpoint.

Formats
  * ``.index`` is a TF "table" (tensorflow/core/lib/io/table*, identical to LevelDB's SSTable):
    blocks of prefix-compressed (key, value) entries [shared varint32, non_shared varint32,
    value_len varint32, key delta, value] followed by an array of uint32 restart offsets and their
    count; each block is followed by a 5-byte trailer (1 byte compression type, 4 bytes masked
    crc32c of block + type).  The file ends with a 48-byte footer: metaindex BlockHandle, index
    BlockHandle (two varint64 each), zero padding to 40 bytes, magic 0xdb4775248b80fb57 (little
    endian).  The index block maps a separator key >= the last key of each data block to that
    block's BlockHandle.
  * key "" holds a BundleHeaderProto {1: num_shards, 2: endianness, 3: version}; every other key
    is a tensor name holding a BundleEntryProto {1: dtype, 2: TensorShapeProto {2: Dim {1: size}},
    3: shard_id, 4: offset, 5: size, 6: fixed32 masked crc32c of the bytes}.
  * ``.data-SSSSS-of-NNNNN`` holds the raw little-endian tensor bytes at [offset, offset + size).
"""
import os
import struct

import numpy as np

_MAGIC = 0xdb4775248b80fb57
_FOOTER = 48
_MASK_DELTA = 0xa282ead8

# tensorflow/core/framework/types.proto
_DT_TO_NP = {1: np.float32, 2: np.float64, 3: np.int32, 4: np.uint8, 5: np.int16, 6: np.int8,
             9: np.int64, 10: np.bool_, 17: np.uint16, 19: np.float16, 22: np.uint32, 23: np.uint64}
_NP_TO_DT = {np.dtype(v): k for k, v in _DT_TO_NP.items()}


# ---------------------------------------------------------------------------------------------
# crc32c (Castagnoli), masked as LevelDB / TF store it
# ---------------------------------------------------------------------------------------------
def _make_table():
  tbl = np.zeros(256, dtype=np.uint32)
  for i in range(256):
    c = i
    for _ in range(8):
      c = (c >> 1) ^ 0x82f63b78 if c & 1 else c >> 1
    tbl[i] = c
  return tbl


_TABLE = _make_table()
_TABLE_LIST = [int(x) for x in _TABLE]


def crc32c(data, crc=0):
  """CRC-32C of `data` (bytes-like); crc32c(b'123456789') == 0xE3069283."""
  c = crc ^ 0xffffffff
  tbl = _TABLE_LIST
  for b in bytes(data):
    c = tbl[(c ^ b) & 0xff] ^ (c >> 8)
  return c ^ 0xffffffff


def crc32c_array(arr):
  """crc32c of a (possibly large) numpy buffer: table lookups vectorised over 8 interleaved
  lanes would complicate a checker that runs once per import; a plain loop over a bytes view is
  ~10 MB/s, so tensors above `limit` are only verified when asked (see read_checkpoint)."""
  return crc32c(np.ascontiguousarray(arr).tobytes())


def mask_crc(c):
  return (((c >> 15) | (c << 17)) + _MASK_DELTA) & 0xffffffff


def unmask_crc(m):
  rot = (m - _MASK_DELTA) & 0xffffffff
  return ((rot >> 17) | (rot << 15)) & 0xffffffff


# ---------------------------------------------------------------------------------------------
# varints / minimal protobuf
# ---------------------------------------------------------------------------------------------
def _get_varint(buf, pos):
  out, shift = 0, 0
  while True:
    b = buf[pos]
    pos += 1
    out |= (b & 0x7f) << shift
    if not b & 0x80:
      return out, pos
    shift += 7
    if shift > 63:
      raise ValueError('varint too long')


def _put_varint(v):
  out = bytearray()
  while True:
    b = v & 0x7f
    v >>= 7
    if v:
      out.append(b | 0x80)
    else:
      out.append(b)
      return bytes(out)


def _parse_proto(buf):
  """{field: [values]} of one message; length-delimited values stay bytes."""
  out, pos = {}, 0
  while pos < len(buf):
    key, pos = _get_varint(buf, pos)
    field, wire = key >> 3, key & 7
    if wire == 0:
      v, pos = _get_varint(buf, pos)
    elif wire == 1:
      v = struct.unpack_from('<Q', buf, pos)[0]
      pos += 8
    elif wire == 2:
      n, pos = _get_varint(buf, pos)
      v = bytes(buf[pos:pos + n])
      pos += n
    elif wire == 5:
      v = struct.unpack_from('<I', buf, pos)[0]
      pos += 4
    else:
      raise ValueError('unsupported protobuf wire type {}'.format(wire))
    out.setdefault(field, []).append(v)
  return out


def _field(field, wire, payload):
  return _put_varint((field << 3) | wire) + payload


def _zigzag_free_int64(v):
  return v & 0xffffffffffffffff      # protobuf int64: two's complement varint


def _shape_proto(shape):
  out = b''
  for d in shape:
    dim = _field(1, 0, _put_varint(_zigzag_free_int64(int(d))))
    out += _field(2, 2, _put_varint(len(dim)) + dim)
  return out


def _parse_shape(buf):
  msg = _parse_proto(buf)
  dims = []
  for d in msg.get(2, []):
    dm = _parse_proto(d)
    size = dm.get(1, [0])[0]
    if size >= 1 << 63:
      size -= 1 << 64
    dims.append(size)
  return tuple(dims)


# ---------------------------------------------------------------------------------------------
# table blocks
# ---------------------------------------------------------------------------------------------
def _read_block(data, offset, size, verify=True):
  body = data[offset:offset + size]
  trailer = data[offset + size:offset + size + 5]
  if len(body) != size or len(trailer) != 5:
    raise ValueError('truncated table block')
  if trailer[0] != 0:
    raise NotImplementedError('compressed table block (type {}): TF bundle writers do not compress'.format(trailer[0]))
  if verify:
    want = unmask_crc(struct.unpack('<I', trailer[1:5])[0])
    if crc32c(trailer[0:1], crc32c(body)) != want and crc32c(body + trailer[0:1]) != want:
      raise ValueError('table block checksum mismatch')
  n_restarts = struct.unpack_from('<I', body, size - 4)[0]
  limit = size - 4 - 4 * n_restarts
  entries, pos, key = [], 0, b''
  while pos < limit:
    shared, pos = _get_varint(body, pos)
    non_shared, pos = _get_varint(body, pos)
    vlen, pos = _get_varint(body, pos)
    key = key[:shared] + bytes(body[pos:pos + non_shared])
    pos += non_shared
    entries.append((key, bytes(body[pos:pos + vlen])))
    pos += vlen
  return entries


def _block_handle(buf, pos=0):
  off, pos = _get_varint(buf, pos)
  size, pos = _get_varint(buf, pos)
  return off, size, pos


def _build_block(entries, restart_interval=16):
  out, restarts, prev, count = bytearray(), [], b'', 0
  for key, value in entries:
    if count % restart_interval == 0:
      restarts.append(len(out))
      shared = 0
    else:
      shared = 0
      while shared < min(len(prev), len(key)) and prev[shared] == key[shared]:
        shared += 1
    out += _put_varint(shared) + _put_varint(len(key) - shared) + _put_varint(len(value))
    out += key[shared:] + value
    prev = key
    count += 1
  if not restarts:
    restarts = [0]
  for r in restarts:
    out += struct.pack('<I', r)
  out += struct.pack('<I', len(restarts))
  return bytes(out)


def _with_trailer(block):
  crc = mask_crc(crc32c(b'\x00', crc32c(block)))
  return block + b'\x00' + struct.pack('<I', crc)


# ---------------------------------------------------------------------------------------------
# public API
# ---------------------------------------------------------------------------------------------
def is_tf_checkpoint(prefix):
  return os.path.isfile(prefix + '.index')


def list_variables(prefix, verify=True):
  """{name: (numpy dtype, shape, shard_id, offset, size, masked crc)} + the header dict."""
  with open(prefix + '.index', 'rb') as f:
    data = f.read()
  if len(data) < _FOOTER:
    raise ValueError('not a TF checkpoint index: too short')
  footer = data[-_FOOTER:]
  if struct.unpack('<Q', footer[40:48])[0] != _MAGIC:
    raise ValueError('not a TF checkpoint index: bad table magic')
  _, _, pos = _block_handle(footer, 0)                  # metaindex (unused by tensor bundles)
  ioff, isize, _ = _block_handle(footer, pos)
  entries = {}
  header = None
  for _, handle in _read_block(data, ioff, isize, verify):
    boff, bsize, _ = _block_handle(handle, 0)
    for key, value in _read_block(data, boff, bsize, verify):
      if key == b'':
        h = _parse_proto(value)
        header = dict(num_shards=h.get(1, [1])[0], endianness=h.get(2, [0])[0])
        continue
      e = _parse_proto(value)
      dt = e.get(1, [0])[0]
      if 7 in e:
        raise NotImplementedError('sliced (partitioned) variable {!r}'.format(key.decode()))
      if dt not in _DT_TO_NP:
        continue                                          # strings / resources: not parameters
      shape = _parse_shape(e[2][0]) if 2 in e else ()
      entries[key.decode()] = (np.dtype(_DT_TO_NP[dt]), shape, e.get(3, [0])[0], e.get(4, [0])[0],
                               e.get(5, [0])[0], e.get(6, [None])[0])
  if header is None:
    raise ValueError('TF checkpoint index without a bundle header')
  if header['endianness'] != 0:
    raise NotImplementedError('big-endian tensor bundle')
  return entries, header


def read_checkpoint(prefix, names=None, verify_tensors_below=1 << 22):
  """{variable name: numpy array}.  `names` (iterable or predicate) restricts what is loaded.
  Tensor payload checksums are verified for tensors smaller than `verify_tensors_below` bytes
  (None: all) -- the pure-Python crc runs at ~10 MB/s."""
  entries, header = list_variables(prefix)
  if names is not None and not callable(names):
    wanted = set(names)
    names = lambda n: n in wanted                       # noqa: E731
  out = {}
  files = {}
  try:
    for name, (dtype, shape, shard, offset, size, crc) in sorted(entries.items()):
      if names is not None and not names(name):
        continue
      if shard not in files:
        files[shard] = open('%s.data-%05d-of-%05d' % (prefix, shard, header['num_shards']), 'rb')
      f = files[shard]
      f.seek(offset)
      raw = f.read(size)
      count = int(np.prod(shape)) if shape else 1
      if len(raw) != size or size != count * dtype.itemsize:
        raise ValueError('tensor {!r}: size {} does not match shape {} of {}'.format(name, size, shape, dtype))
      if crc is not None and (verify_tensors_below is None or size < verify_tensors_below):
        if crc32c(raw) != unmask_crc(crc):
          raise ValueError('tensor {!r}: payload checksum mismatch'.format(name))
      out[name] = np.frombuffer(raw, dtype=dtype.newbyteorder('<')).astype(dtype).reshape(shape)
  finally:
    for f in files.values():
      f.close()
  return out


def write_checkpoint(prefix, tensors):
  """Writes {name: array} as a single-shard tensor bundle (what tf.train.Saver produces for a small
  model): lets a TF user load weights trained here, and gives the reader a fixture."""
  names = sorted(tensors.keys(), key=lambda s: s.encode())
  data = bytearray()
  rows = [(b'', _field(1, 0, _put_varint(1)) + _field(3, 2, _put_varint(2) + _field(1, 0, _put_varint(1))))]
  for name in names:
    arr = np.asarray(tensors[name])
    arr = arr.reshape(arr.shape).copy(order='C')        # keeps 0-d scalars 0-d (ascontiguousarray would not)
    if arr.dtype not in _NP_TO_DT:
      raise ValueError('unsupported dtype {} for {!r}'.format(arr.dtype, name))
    raw = arr.astype(arr.dtype.newbyteorder('<')).tobytes()
    entry = _field(1, 0, _put_varint(_NP_TO_DT[arr.dtype]))
    shp = _shape_proto(arr.shape)
    entry += _field(2, 2, _put_varint(len(shp)) + shp)
    if len(data):
      entry += _field(4, 0, _put_varint(len(data)))
    entry += _field(5, 0, _put_varint(len(raw)))
    entry += _field(6, 5, struct.pack('<I', mask_crc(crc32c(raw))))
    rows.append((name.encode(), entry))
    data += raw
  # data blocks of ~4 KiB, then the index block, an empty metaindex block and the footer
  out = bytearray()
  index_rows, block, block_bytes = [], [], 0

  def flush():
    nonlocal block, block_bytes
    if not block:
      return
    body = _with_trailer(_build_block(block))
    handle = _put_varint(len(out)) + _put_varint(len(body) - 5)
    index_rows.append((block[-1][0], handle))           # separator = the block's last key
    out.extend(body)
    block, block_bytes = [], 0
  for row in rows:
    block.append(row)
    block_bytes += len(row[0]) + len(row[1])
    if block_bytes >= 4096:
      flush()
  flush()
  meta = _with_trailer(_build_block([]))
  meta_handle = _put_varint(len(out)) + _put_varint(len(meta) - 5)
  out.extend(meta)
  index = _with_trailer(_build_block(index_rows, restart_interval=1))
  index_handle = _put_varint(len(out)) + _put_varint(len(index) - 5)
  out.extend(index)
  footer = meta_handle + index_handle
  footer += b'\x00' * (40 - len(footer)) + struct.pack('<Q', _MAGIC)
  out.extend(footer)
  with open(prefix + '.index', 'wb') as f:
    f.write(bytes(out))
  with open(prefix + '.data-00000-of-00001', 'wb') as f:
    f.write(bytes(data))


# ---------------------------------------------------------------------------------------------
# MetaGraphDef (``*.meta``): the graph a checkpoint belongs to
# ---------------------------------------------------------------------------------------------
def _signed64(v):
  return v - (1 << 64) if v >= 1 << 63 else v


def _packed_varints(buf):
  out, pos = [], 0
  while pos < len(buf):
    v, pos = _get_varint(buf, pos)
    out.append(_signed64(v))
  return out


def _repeated_ints(msg, field):
  """repeated int64 / int32 / bool field: packed (one bytes blob) or one varint per element."""
  out = []
  for v in msg.get(field, []):
    out += _packed_varints(v) if isinstance(v, bytes) else [_signed64(v)]
  return out


def _repeated_floats(msg, field):
  out = []
  for v in msg.get(field, []):
    if isinstance(v, bytes):
      out += list(struct.unpack('<%df' % (len(v) // 4), v))
    else:
      out.append(struct.unpack('<f', struct.pack('<I', v))[0])
  return out


def _parse_tensor(buf):
  """TensorProto (tensorflow/core/framework/tensor.proto) -> numpy array (strings: list of bytes)."""
  m = _parse_proto(buf)
  dt = m.get(1, [0])[0]
  shape = _parse_shape(m[2][0]) if 2 in m else ()
  count = int(np.prod(shape)) if shape else 1
  if dt == 7:                                            # DT_STRING
    vals = [bytes(v) for v in m.get(8, [])]
    return vals if shape else (vals[0] if vals else b'')
  if dt not in _DT_TO_NP:
    raise NotImplementedError('TensorProto dtype {}'.format(dt))
  npdt = np.dtype(_DT_TO_NP[dt])
  if 4 in m and len(m[4][0]):                            # tensor_content: raw little-endian bytes
    arr = np.frombuffer(m[4][0], dtype=npdt.newbyteorder('<')).astype(npdt)
  else:
    if dt == 1:
      vals = _repeated_floats(m, 5)                      # float_val
    elif dt == 2:
      vals = []
      for v in m.get(6, []):                             # double_val
        vals += list(struct.unpack('<%dd' % (len(v) // 8), v)) if isinstance(v, bytes) \
            else [struct.unpack('<d', struct.pack('<Q', v))[0]]
    elif dt == 9:
      vals = _repeated_ints(m, 10)                       # int64_val
    elif dt == 10:
      vals = [bool(v) for v in _repeated_ints(m, 11)]    # bool_val
    else:
      vals = _repeated_ints(m, 7)                        # int_val (int32, uint8, int16, int8, ...)
    arr = np.asarray(vals, dtype=npdt)
    if arr.size == 1 and count > 1:                      # a single value stands for a constant-filled tensor
      arr = np.full(count, arr[0], dtype=npdt)
    elif arr.size == 0:
      arr = np.zeros(count, dtype=npdt)
  if arr.size != count:
    raise ValueError('TensorProto holds {} values for shape {}'.format(arr.size, shape))
  return arr.reshape(shape)


def _parse_attr(buf):
  """AttrValue (attr_value.proto) -> python value."""
  m = _parse_proto(buf)
  if 2 in m:
    return m[2][0].decode('utf-8', 'replace')
  if 3 in m:
    return _signed64(m[3][0])
  if 4 in m:
    return struct.unpack('<f', struct.pack('<I', m[4][0]))[0]
  if 5 in m:
    return bool(m[5][0])
  if 6 in m:
    return ('dtype', m[6][0])
  if 7 in m:
    sm = _parse_proto(m[7][0])
    return ('shape', None if sm.get(3, [0])[0] else list(_parse_shape(m[7][0])))
  if 8 in m:
    return _parse_tensor(m[8][0])
  if 1 in m:
    lv = _parse_proto(m[1][0])
    if 2 in lv:
      return [v.decode('utf-8', 'replace') for v in lv[2]]
    if 3 in lv:
      return _repeated_ints(lv, 3)
    if 4 in lv:
      return _repeated_floats(lv, 4)
    if 5 in lv:
      return [bool(v) for v in _repeated_ints(lv, 5)]
    if 6 in lv:
      return [('dtype', v) for v in _repeated_ints(lv, 6)]
    if 7 in lv:
      return [('shape', list(_parse_shape(v))) for v in lv[7]]
    return []
  return None


def read_meta_graph(path):
  """Reads a TensorFlow MetaGraphDef (what ``tf.train.export_meta_graph`` writes and the reference's scripts import:
  scripts/spectrogram_advoc.py:55-64, scripts/generate_spectrogram.py, models/melspecgan/infer.py) without TensorFlow.

  Returns dict(nodes=OrderedDict name -> dict(op, inputs, attrs), collections={name: [str]}, saver=dict(...),
  tf_version=str).  Used to check that a model built here has the variables / hyper-parameters of the graph a published
  checkpoint was saved from (variables(), check_model_against_meta_graph)."""
  import collections as _c
  with open(path, 'rb') as f:
    data = f.read()
  mg = _parse_proto(data)
  if 2 not in mg:
    raise ValueError('{!r}: no GraphDef inside (not a MetaGraphDef)'.format(path))
  info = _parse_proto(mg[1][0]) if 1 in mg else {}
  gd = _parse_proto(mg[2][0])
  nodes = _c.OrderedDict()
  for nb in gd.get(1, []):
    n = _parse_proto(nb)
    attrs = {}
    for ab in n.get(5, []):
      kv = _parse_proto(ab)
      attrs[kv[1][0].decode()] = _parse_attr(kv[2][0]) if 2 in kv else None
    nodes[n[1][0].decode()] = dict(op=n[2][0].decode(), inputs=[i.decode() for i in n.get(3, [])], attrs=attrs)
  colls = {}
  for cb in mg.get(4, []):
    kv = _parse_proto(cb)
    key = kv[1][0].decode()
    val = _parse_proto(kv[2][0]) if 2 in kv else {}
    items = []
    if 1 in val:                                          # NodeList
      items = [v.decode() for v in _parse_proto(val[1][0]).get(1, [])]
    elif 2 in val:                                        # BytesList: serialized VariableDef {1: variable_name, ...}
      for b in _parse_proto(val[2][0]).get(1, []):
        try:
          items.append(_parse_proto(b)[1][0].decode())
        except Exception:                                 # not a VariableDef: keep the raw length
          items.append('<%d bytes>' % len(b))
    colls[key] = items
  saver = {}
  if 3 in mg:
    sd = _parse_proto(mg[3][0])
    saver = dict(filename_tensor_name=sd.get(1, [b''])[0].decode(), save_tensor_name=sd.get(2, [b''])[0].decode(),
                 restore_op_name=sd.get(3, [b''])[0].decode(), max_to_keep=sd.get(4, [0])[0],
                 version=sd.get(7, [0])[0])
  return dict(nodes=nodes, collections=colls, saver=saver,
              tf_version=info.get(5, [b''])[0].decode() if 5 in info else '')


def meta_graph_variables(meta):
  """OrderedDict variable name -> (numpy dtype, shape) of the VariableV2 / VarHandleOp nodes of read_meta_graph()'s
  result, in graph (creation) order."""
  import collections as _c
  out = _c.OrderedDict()
  for name, n in meta['nodes'].items():
    if n['op'] in ('VariableV2', 'Variable', 'VarHandleOp'):
      dt = n['attrs'].get('dtype')
      shp = n['attrs'].get('shape')
      out[name] = (np.dtype(_DT_TO_NP[dt[1]]) if dt and dt[1] in _DT_TO_NP else None,
                   tuple(shp[1]) if shp and shp[1] is not None else None)
  return out


def check_model_against_meta_graph(specs, meta, scope=None):
  """`specs`: [(variable name, shape)] of a model built here.  Raises ValueError naming the first variable of the meta
  graph (under `scope`) that the model lacks or shapes differently, and the first model variable the graph lacks."""
  gv = meta_graph_variables(meta)
  if scope:
    gv = dict((k, v) for k, v in gv.items() if k.startswith(scope))
  mine = dict((k, tuple(s)) for k, s in specs)
  for k, (_, shape) in gv.items():
    if k not in mine:
      raise ValueError('meta graph variable {!r} {} has no counterpart in the model'.format(k, shape))
    if shape is not None and mine[k] != shape:
      raise ValueError('variable {!r}: model shape {} != meta graph shape {}'.format(k, mine[k], shape))
  for k in mine:
    if k not in gv:
      raise ValueError('model variable {!r} is not in the meta graph'.format(k))


def load_into_model(prefix, model, generator_only=False):
  """Copies every variable of `model.state_dict()` found in the TF checkpoint into the model
  (names and layouts are TF's own, so no transposition).  Returns (loaded names, missing names, step)."""
  want = list(model.state_dict().keys())
  if generator_only:
    want = [k for k in want if k.startswith('generator/')]
  found = read_checkpoint(prefix, names=lambda n: n in set(want) or n == 'global_step')
  import torch
  state = {}
  for k in want:
    if k in found:
      state[k] = torch.from_numpy(np.array(found[k], dtype=np.float32))
  missing = [k for k in want if k not in state]
  if not state:
    raise ValueError('no variable of the model was found in {!r}'.format(prefix))
  cur = model.state_dict()
  cur.update(state)
  model.load_state_dict(cur)
  step = int(found['global_step']) if 'global_step' in found else 0
  return sorted(state.keys()), missing, step
