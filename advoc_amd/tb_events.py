"""TensorBoard event files (scalars, images, audio) without TensorFlow: the reference logs its losses with
tf.summary.scalar (models/advoc/advoc_model.py:259-281, train_evaluate.py:139-141) and watches them in
TensorBoard; this writer produces files TensorBoard reads from the same WORK_DIR.

Formats (tensorflow/core/lib/io/record_writer.cc, tensorflow/core/util/event.proto,
framework/summary.proto):
  record  = uint64 length | uint32 masked_crc32c(length bytes) | data | uint32 masked_crc32c(data)
  Event   = {1: double wall_time, 2: int64 step, 3: string file_version | 5: Summary}
  Summary = {1: repeated Value {1: string tag, 2: float simple_value | 4: Image | 6: Audio}}
  Image   = {1: height, 2: width, 3: colorspace (1 = grayscale), 4: bytes encoded_image_string (PNG)}
  Audio   = {1: float sample_rate, 2: num_channels, 3: length_frames, 4: bytes encoded_audio_string (WAV), 5: content_type}
Image / audio values follow tf.summary.image / tf.summary.audio of TF 1.x (tags `<name>/image/<i>`, `<name>/audio/<i>`,
at most max_outputs = 3 per call; float images normalised per image as tensorflow/core/kernels/summary_image_op.cc
does; audio clipped to [-1, 1] and stored as 16-bit PCM WAV as summary_audio_op.cc does).
The first record of a file is Event{file_version: "brain.Event:2"}.
UNTESTED against TensorBoard itself (not installed here); the reader below parses what the writer
produces and the checksums use the crc32c with published test vectors (advoc_amd/tf_checkpoint.py)."""
import io
import os
import socket
import struct
import time
import wave
import zlib

import numpy as np

from advoc_amd.tf_checkpoint import _field, _parse_proto, _put_varint, crc32c, mask_crc, unmask_crc


def _record(data):
  head = struct.pack('<Q', len(data))
  return head + struct.pack('<I', mask_crc(crc32c(head))) + data + struct.pack('<I', mask_crc(crc32c(data)))


def _bytes_field(num, data):
  return _field(num, 2, _put_varint(len(data)) + data)


def png_gray(img_u8):
  """8-bit grayscale PNG of a [H, W] uint8 array (zlib + the three mandatory chunks)."""
  img_u8 = np.ascontiguousarray(img_u8, dtype=np.uint8)
  h, w = img_u8.shape

  def chunk(kind, data):
    body = kind + data
    return struct.pack('>I', len(data)) + body + struct.pack('>I', zlib.crc32(body) & 0xffffffff)
  raw = b''.join(b'\x00' + img_u8[r].tobytes() for r in range(h))       # filter type 0 per scanline
  return (b'\x89PNG\r\n\x1a\n' + chunk(b'IHDR', struct.pack('>IIBBBBB', w, h, 8, 0, 0, 0, 0))
          + chunk(b'IDAT', zlib.compress(raw, 6)) + chunk(b'IEND', b''))


def normalize_image(img):
  """float [H, W] -> uint8 as the tf.summary.image kernel does for float input (summary_image_op.cc,
  NormalizeFloatImage, restated from memory of TF 1.x): no negative value -> scale 255 / max, offset 0; otherwise
  scale 127 / max|x|, offset 128; values cast after clamping to [0, 255]."""
  img = np.asarray(img, dtype=np.float32)
  lo, hi = float(img.min()), float(img.max())
  if lo >= 0:
    scale, offset = (255.0 / hi if hi > 1e-6 else 0.0), 0.0
  else:
    m = max(abs(lo), abs(hi))
    scale, offset = (127.0 / m if m > 1e-6 else 0.0), 128.0
  return np.clip(img * scale + offset, 0, 255).astype(np.uint8)


def wav16(samples, rate):
  samples = np.clip(np.asarray(samples, dtype=np.float32).reshape(-1), -1.0, 1.0)
  buf = io.BytesIO()
  with wave.open(buf, 'wb') as f:
    f.setnchannels(1)
    f.setsampwidth(2)
    f.setframerate(int(rate))
    f.writeframes((samples * 32767.0).astype('<i2').tobytes())
  return buf.getvalue()


def _image_value(tag, img):
  u8 = img if np.asarray(img).dtype == np.uint8 else normalize_image(img)
  body = (_field(1, 0, _put_varint(u8.shape[0])) + _field(2, 0, _put_varint(u8.shape[1])) + _field(3, 0, _put_varint(1))
          + _bytes_field(4, png_gray(u8)))
  return _bytes_field(1, tag.encode()) + _bytes_field(4, body)


def _audio_value(tag, samples, rate):
  n = int(np.asarray(samples).size)
  body = (_field(1, 5, struct.pack('<f', float(rate))) + _field(2, 0, _put_varint(1)) + _field(3, 0, _put_varint(n))
          + _bytes_field(4, wav16(samples, rate)) + _bytes_field(5, b'audio/wav'))
  return _bytes_field(1, tag.encode()) + _bytes_field(6, body)


def _event(wall_time, step=None, file_version=None, scalars=None, values=()):
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
  if values:
    summary = b''.join(_bytes_field(1, v) for v in values)
    out += _bytes_field(5, summary)
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

  def add_images(self, images, step, wall_time=None, max_outputs=3):
    """{name: [N, H, W] (or [H, W]) float / uint8}: tags `name/image/i` (`name/image` when max_outputs == 1), i <
    min(N, max_outputs), grayscale PNG -- tf.summary.image(name, tensor[N, H, W, 1])."""
    vals = []
    for name, batch in sorted(images.items()):
      batch = np.asarray(batch)
      batch = batch[None] if batch.ndim == 2 else batch
      for i in range(min(len(batch), max_outputs)):
        vals.append(_image_value('%s/image/%d' % (name, i) if max_outputs > 1 else name + '/image', batch[i]))
    self._write(wall_time, step, vals)

  def add_audio(self, clips, step, sample_rate, wall_time=None, max_outputs=3):
    """{name: [N, samples] (or [samples]) float in [-1, 1]}: tags `name/audio/i` -- tf.summary.audio."""
    vals = []
    for name, batch in sorted(clips.items()):
      batch = np.asarray(batch, dtype=np.float32)
      batch = batch[None] if batch.ndim == 1 else batch
      for i in range(min(len(batch), max_outputs)):
        vals.append(_audio_value('%s/audio/%d' % (name, i) if max_outputs > 1 else name + '/audio', batch[i], sample_rate))
    self._write(wall_time, step, vals)

  def _write(self, wall_time, step, values):
    if values:
      self._f.write(_record(_event(wall_time if wall_time is not None else time.time(), step=step, values=values)))
      self._f.flush()

  def close(self):
    self._f.close()


def read_events(path, kinds=('scalar',)):
  """[(step, {tag: value})] of the events in a file (used by the tests; verifies every checksum).  Scalars by default;
  with 'image' / 'audio' in `kinds` those values come back as dicts (height, width, png) / (sample_rate, frames, wav)."""
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
      tag = m[1][0].decode()
      if 2 in m and 'scalar' in kinds:
        vals[tag] = struct.unpack('<f', struct.pack('<I', m[2][0]))[0]
      elif 4 in m and 'image' in kinds:
        im = _parse_proto(m[4][0])
        vals[tag] = dict(height=im[1][0], width=im[2][0], colorspace=im[3][0], png=im[4][0])
      elif 6 in m and 'audio' in kinds:
        au = _parse_proto(m[6][0])
        vals[tag] = dict(sample_rate=struct.unpack('<f', struct.pack('<I', au[1][0]))[0], channels=au[2][0],
                         frames=au[3][0], wav=au[4][0], content_type=au[5][0].decode())
    if vals:
      out.append((step, vals))
  return out
