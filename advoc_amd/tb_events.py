"""TensorBoard event files (scalars) without TensorFlow: the reference logs its losses with
tf.summary.scalar (models/advoc/advoc_model.py:259-281, train_evaluate.py:139-141) and watches them in
TensorBoard; this writer produces files TensorBoard reads from the same WORK_DIR.

Formats (tensorflow/core/lib/io/record_writer.cc, tensorflow/core/util/event.proto,
framework/summary.proto):
  record  = uint64 length | uint32 masked_crc32c(length bytes) | data | uint32 masked_crc32c(data)
  Event   = {1: double wall_time, 2: int64 step, 3: string file_version | 5: Summary}
  Summary = {1: repeated Value {1: string tag, 2: float simple_value}}
The first record of a file is Event{file_version: "brain.Event:2"}.
UNTESTED against TensorBoard itself (not installed here); the reader below parses what the writer
produces and the checksums use the crc32c with published test vectors (advoc_amd/tf_checkpoint.py)."""
import os
import socket
import struct
import time

from advoc_amd.tf_checkpoint import _field, _parse_proto, _put_varint, crc32c, mask_crc, unmask_crc


def _record(data):
  head = struct.pack('<Q', len(data))
  return head + struct.pack('<I', mask_crc(crc32c(head))) + data + struct.pack('<I', mask_crc(crc32c(data)))


def _event(wall_time, step=None, file_version=None, scalars=None):
  out = _field(1, 1, struct.pack('<d', wall_time))
  if step is not None:
    out += _field(2, 0, _put_varint(int(step) & 0xffffffffffffffff))
  if file_version is not None:
    v = file_version.encode()
    out += _field(3, 2, _put_varint(len(v)) + v)
  if scalars:
    summary = b''
    for tag, value in scalars:
      t = tag.encode()
      val = _field(1, 2, _put_varint(len(t)) + t) + _field(2, 5, struct.pack('<f', float(value)))
      summary += _field(1, 2, _put_varint(len(val)) + val)
    out += _field(5, 2, _put_varint(len(summary)) + summary)
  return out


class EventWriter(object):
  """add_scalars({'disc_loss': 1.2, ...}, step) -> WORK_DIR/events.out.tfevents.<time>.<host>"""

  def __init__(self, logdir):
    os.makedirs(logdir, exist_ok=True)
    self.path = os.path.join(logdir, 'events.out.tfevents.%010d.%s' % (int(time.time()), socket.gethostname()))
    self._f = open(self.path, 'ab')
    self._f.write(_record(_event(time.time(), file_version='brain.Event:2')))
    self._f.flush()

  def add_scalars(self, scalars, step, wall_time=None):
    items = sorted(scalars.items()) if isinstance(scalars, dict) else list(scalars)
    self._f.write(_record(_event(wall_time if wall_time is not None else time.time(), step=step, scalars=items)))
    self._f.flush()

  def close(self):
    self._f.close()


def read_events(path):
  """[(step, {tag: value})] of the scalar events in a file (used by the tests; verifies every checksum)."""
  data = open(path, 'rb').read()
  pos, out = 0, []
  while pos < len(data):
    (n,) = struct.unpack_from('<Q', data, pos)
    if unmask_crc(struct.unpack_from('<I', data, pos + 8)[0]) != crc32c(data[pos:pos + 8]):
      raise ValueError('event record: length checksum mismatch')
    body = data[pos + 12:pos + 12 + n]
    if unmask_crc(struct.unpack_from('<I', data, pos + 12 + n)[0]) != crc32c(body):
      raise ValueError('event record: data checksum mismatch')
    pos += 16 + n
    ev = _parse_proto(body)
    if 5 not in ev:
      continue
    step = ev.get(2, [0])[0]
    vals = {}
    for v in _parse_proto(ev[5][0]).get(1, []):
      m = _parse_proto(v)
      vals[m[1][0].decode()] = struct.unpack('<f', struct.pack('<I', m[2][0]))[0]
    out.append((step, vals))
  return out
