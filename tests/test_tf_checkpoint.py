"""TensorFlow tensor-bundle checkpoints without TensorFlow (advoc_amd/tf_checkpoint.py).

No file written by TensorFlow exists on the build or GPU boxes, so the reader is checked against
known-answer vectors of its building blocks (crc32c, the LevelDB checksum mask, the table footer)
and against this module's own writer -- see the STATUS note in the module."""
import os
import struct

import numpy as np
import pytest

from advoc_amd import tf_checkpoint as T

gpu = pytest.mark.gpu


def test_crc32c_known_answers():
  assert T.crc32c(b'123456789') == 0xE3069283                      # the standard check value
  assert T.crc32c(b'') == 0
  assert T.crc32c(bytes(32)) == 0x8A9136AA                         # RFC 3720 B.4: 32 zero bytes
  assert T.crc32c(bytes([0xff] * 32)) == 0x62A8AB43                # RFC 3720 B.4: 32 0xff bytes
  assert T.crc32c(bytes(range(32))) == 0x46DD794E                  # RFC 3720 B.4: 0..31
  assert T.crc32c(b'6789', T.crc32c(b'12345')) == 0xE3069283       # Extend
  for c in (0, 1, 0xdeadbeef, 0xffffffff):
    assert T.unmask_crc(T.mask_crc(c)) == c
  assert T.mask_crc(0) == 0xa282ead8


def test_varints_and_proto():
  for v in (0, 1, 127, 128, 300, 2 ** 31, 2 ** 63 - 1):
    buf = T._put_varint(v)
    assert T._get_varint(buf, 0) == (v, len(buf))
  assert T._put_varint(300) == b'\xac\x02'
  shape = T._shape_proto((4, 4, 1, 32))
  assert T._parse_shape(shape) == (4, 4, 1, 32)
  assert T._parse_shape(b'') == ()


def test_round_trip_many_variables(tmp_path):
  rng = np.random.default_rng(0)
  tensors = {'global_step': np.array(1234, dtype=np.int64),
             'beta1_power': np.array(0.5, dtype=np.float32)}
  for i in range(300):      # enough entries for several 4 KiB data blocks and a multi-entry index block
    tensors['generator/layer_%03d/conv2d/kernel' % i] = rng.standard_normal((3, 2, 1 + i % 3)).astype(np.float32)
    tensors['generator/layer_%03d/conv2d/bias' % i] = rng.standard_normal(1 + i % 5).astype(np.float32)
  prefix = str(tmp_path / 'model.ckpt-1234')
  T.write_checkpoint(prefix, tensors)
  assert T.is_tf_checkpoint(prefix)
  raw = open(prefix + '.index', 'rb').read()
  assert struct.unpack('<Q', raw[-8:])[0] == 0xdb4775248b80fb57
  entries, header = T.list_variables(prefix)
  assert header == dict(num_shards=1, endianness=0) and len(entries) == len(tensors)
  got = T.read_checkpoint(prefix, verify_tensors_below=None)
  assert set(got) == set(tensors)
  for k, v in tensors.items():
    assert got[k].dtype == v.dtype and got[k].shape == v.shape and np.array_equal(got[k], v), k
  only = T.read_checkpoint(prefix, names=['global_step'])
  assert list(only) == ['global_step'] and int(only['global_step']) == 1234


def test_corruption_is_detected(tmp_path):
  prefix = str(tmp_path / 'm')
  T.write_checkpoint(prefix, {'a': np.arange(10, dtype=np.float32)})
  data = bytearray(open(prefix + '.data-00000-of-00001', 'rb').read())
  data[3] ^= 0x40
  open(prefix + '.data-00000-of-00001', 'wb').write(bytes(data))
  with pytest.raises(ValueError, match='checksum'):
    T.read_checkpoint(prefix)
  idx = bytearray(open(prefix + '.index', 'rb').read())
  idx[2] ^= 0x01
  open(prefix + '.index', 'wb').write(bytes(idx))
  with pytest.raises(ValueError):
    T.list_variables(prefix)
  open(prefix + '.index', 'wb').write(b'not a table')
  with pytest.raises(ValueError):
    T.list_variables(prefix)


@gpu
def test_model_restores_from_a_tf_checkpoint(hip, tmp_path):
  """A generator saved in TF's container under TF's variable names loads into the model, through
  restore_checkpoint and through the vocoding script's loader."""
  import torch
  from advoc_amd.infer import load_generator
  from advoc_amd.model import AdvocSmall, Modes
  from advoc_amd.train_evaluate import latest_checkpoint, restore_checkpoint
  src = AdvocSmall(Modes.TRAIN)
  src.subseq_len = 32
  src.build(batch_size=1, seed=5)
  sd = {k: v.cpu().numpy() for k, v in src.state_dict().items()}
  tensors = dict(sd)
  tensors['global_step'] = np.array(77, dtype=np.int64)
  tensors['generator/encoder_1/conv2d/kernel/Adam'] = np.zeros_like(sd['generator/encoder_1/conv2d/kernel'])
  prefix = str(tmp_path / 'model.ckpt-77')
  T.write_checkpoint(prefix, tensors)
  open(str(tmp_path / 'checkpoint'), 'w').write('model_checkpoint_path: "model.ckpt-77"\nall_model_checkpoint_paths: "model.ckpt-77"\n')
  assert latest_checkpoint(str(tmp_path)) == prefix
  dst = AdvocSmall(Modes.TRAIN)
  dst.subseq_len = 32
  dst.build(batch_size=1, seed=6)
  assert restore_checkpoint(prefix, dst) == 77
  for k, v in dst.state_dict().items():
    if k == 'global_step':
      assert int(v) == 77
    else:
      assert torch.equal(v.cpu(), torch.from_numpy(sd[k])), k
  g = load_generator(prefix, 'small', subseq_len=32)
  x = torch.rand(1, 32, 513, 1)
  src.set_dropout_masks(None)
  assert tuple(g.build_generator(x).shape) == (1, 32, 513, 1)


def test_tensorboard_event_file_round_trip(tmp_path):
  from advoc_amd.tb_events import EventWriter, read_events
  w = EventWriter(str(tmp_path))
  w.add_scalars({'disc_loss': 1.25, 'gen_loss_L1': 0.5}, 10, wall_time=123.0)
  w.add_scalars({'disc_loss': 1.0}, 2 ** 40)
  w.close()
  raw = open(w.path, 'rb').read()
  assert os.path.basename(w.path).startswith('events.out.tfevents.')
  assert b'brain.Event:2' in raw[:64]
  ev = read_events(w.path)
  assert ev == [(10, {'disc_loss': 1.25, 'gen_loss_L1': 0.5}), (2 ** 40, {'disc_loss': 1.0})]
  bad = bytearray(raw)
  bad[-6] ^= 1
  open(w.path, 'wb').write(bytes(bad))
  with pytest.raises(ValueError):
    read_events(w.path)


@gpu
def test_tf_format_export_next_to_native_checkpoints(hip, tmp_path, monkeypatch):
  """ADVOC_EXPORT_TF_CKPT=1: save_checkpoint also writes a tensor bundle + a CheckpointState `checkpoint` index (what
  tf.train.latest_checkpoint reads); the bundle lists the TF variable names, the Adam slots and global_step, restores
  into a fresh model, and latest_checkpoint keeps resolving to the native file (with optimiser state) for resuming."""
  import torch
  from advoc_amd import tf_checkpoint
  from advoc_amd import train_evaluate as TE
  from advoc_amd.model import AdvocSmall, Modes
  monkeypatch.setenv('ADVOC_EXPORT_TF_CKPT', '1')
  m = AdvocSmall(Modes.TRAIN)
  m.subseq_len = 32
  m.build(batch_size=2, seed=3)
  m.step = 7
  d = str(tmp_path)
  fp = TE.save_checkpoint(d, m)
  assert fp.endswith('model.ckpt-7.pt')
  text = open(os.path.join(d, 'checkpoint')).read()
  assert text.startswith('model_checkpoint_path: "model.ckpt-7"')
  prefix = os.path.join(d, 'model.ckpt-7')
  assert tf_checkpoint.is_tf_checkpoint(prefix)
  names, _ = tf_checkpoint.list_variables(prefix)
  for k in ('generator/encoder_1/conv2d/kernel', 'discriminator/layer_5/conv2d/bias', 'global_step', 'beta1_power',
            'beta2_power_1', 'generator/decoder_1/conv2d_transpose/kernel/Adam', 'discriminator/layer_1/conv2d/kernel/Adam_1'):
    assert k in names, k
  assert TE.latest_checkpoint(d) == fp                      # resume from the native file
  m2 = AdvocSmall(Modes.INFER)
  m2.subseq_len = 32
  m2.build(batch_size=2, seed=99)
  assert TE.restore_checkpoint(prefix, m2, with_optimizer=False) == 7
  a, b = m.state_dict(), m2.state_dict()
  for k in a:
    if k != 'global_step':
      assert torch.equal(a[k].cpu(), b[k].cpu()), k


def test_prune_removes_the_tensorflow_twins_of_a_pruned_checkpoint(tmp_path):
  """max_to_keep bounds disk use also with ADVOC_EXPORT_TF_CKPT=1: the .index / .data-* bundle of a pruned step goes
  with its .pt."""
  from advoc_amd import train_evaluate as TE
  for step in (3, 10, 20, 100):
    for ext in ('.pt', '.index', '.data-00000-of-00001'):
      (tmp_path / ('model.ckpt-%d%s' % (step, ext))).write_bytes(b'x')
  TE._prune(str(tmp_path), 'model.ckpt-*.pt', 2)
  left = sorted(os.listdir(tmp_path))
  assert left == sorted('model.ckpt-%d%s' % (s_, e) for s_ in (20, 100) for e in ('.pt', '.index', '.data-00000-of-00001'))


def test_event_file_image_and_audio_summaries(tmp_path):
  """tf.summary.image / tf.summary.audio values (advoc_model.py:268-281): tags, PNG and WAV payloads decode back."""
  import io
  import wave
  import zlib
  from advoc_amd.tb_events import EventWriter, normalize_image, read_events
  rng = np.random.RandomState(0)
  imgs = rng.rand(4, 5, 7).astype(np.float32)
  imgs[1] -= 0.5
  clip = np.sin(np.arange(2205) * 0.05).astype(np.float32) * 1.2        # beyond [-1, 1]: clipped
  w = EventWriter(str(tmp_path))
  w.add_scalars({'disc_loss': 1.5}, 3)
  w.add_images({'generated_magspec': imgs}, 3)
  w.add_audio({'gen_audio': clip}, 3, 22050)
  w.close()
  assert read_events(w.path) == [(3, {'disc_loss': 1.5})]
  ev = read_events(w.path, kinds=('image', 'audio'))
  (s1, im), (s2, au) = ev
  assert s1 == s2 == 3
  assert sorted(im) == ['generated_magspec/image/%d' % i for i in range(3)]           # max_outputs = 3
  for i in range(3):
    v = im['generated_magspec/image/%d' % i]
    assert (v['height'], v['width'], v['colorspace']) == (5, 7, 1)
    png = v['png']
    assert png[:8] == b'\x89PNG\r\n\x1a\n'
    pos, idat = 8, b''
    while pos < len(png):
      n, kind = struct.unpack('>I', png[pos:pos + 4])[0], png[pos + 4:pos + 8]
      body = png[pos + 8:pos + 8 + n]
      assert struct.unpack('>I', png[pos + 8 + n:pos + 12 + n])[0] == zlib.crc32(kind + body) & 0xffffffff
      if kind == b'IDAT':
        idat += body
      pos += 12 + n
    rows = np.frombuffer(zlib.decompress(idat), np.uint8).reshape(5, 8)
    assert (rows[:, 0] == 0).all()
    assert (rows[:, 1:] == normalize_image(imgs[i])).all()
  assert normalize_image(imgs[0]).max() >= 254 and normalize_image(imgs[1]).min() >= 0
  a = au['gen_audio/audio/0']
  assert a['content_type'] == 'audio/wav' and a['frames'] == 2205 and a['sample_rate'] == 22050.0
  with wave.open(io.BytesIO(a['wav'])) as f:
    assert (f.getnchannels(), f.getsampwidth(), f.getframerate(), f.getnframes()) == (1, 2, 22050, 2205)
    pcm = np.frombuffer(f.readframes(2205), '<i2')
  assert abs(pcm).max() == 32767
  assert np.abs(pcm / 32767.0 - np.clip(clip, -1, 1)).max() < 1e-4
