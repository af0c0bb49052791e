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
