"""train / eval / infer driver for AdVoc on MI355X -- same command line as the reference's
models/advoc/train_evaluate.py (argparse :338-371, datacfg parsing :375-382, dispatch :387-397).

    python models/advoc/train_evaluate.py train WORK_DIR --data_cfg datacfg/ljspeech.txt \
        --data_dir /path/to/wavs [--model_type small] [--model_overrides "train_batch_size=32"]

Multi-GPU: launch one process per GPU with torch.distributed.run; every rank takes a strided
shard of the file list and gradients are all-reduced over RCCL (advoc_amd/parallel.py).

Differences from the reference, by necessity: checkpoints are torch files
(WORK_DIR/model.ckpt-<step>.pt + a `checkpoint` index; variables keep their TF names; TensorFlow
checkpoint prefixes can be restored too), scalar summaries go to WORK_DIR/summaries.jsonl AND to a
TensorBoard event file written without TensorFlow (advoc_amd/tb_events.py); image / audio summaries
are not produced.
"""
import glob
import json
import os
import time

import numpy as np
import torch


def _make_model(args, mode):
  from advoc_amd.model import Advoc, AdvocSmall, override_model_attrs
  if args.model_type == 'regular':
    model = Advoc(mode)
  elif args.model_type == 'small':
    model = AdvocSmall(mode)
  else:
    raise NotImplementedError()
  model, summary = override_model_attrs(model, args.model_overrides)
  model.audio_fs = args.data_sample_rate
  print('-' * 80)
  print(summary)
  print('-' * 80)
  return model


def latest_checkpoint(train_dir):
  """Path of the newest checkpoint recorded in WORK_DIR/checkpoint (like tf.train.latest_checkpoint)."""
  idx = os.path.join(train_dir, 'checkpoint')
  if not os.path.isfile(idx):
    return None
  text = open(idx).read().strip()
  name = text
  if 'model_checkpoint_path' in text:        # TensorFlow's CheckpointState text proto
    import re
    m = re.search(r'model_checkpoint_path:\s*"([^"]+)"', text)
    name = m.group(1) if m else text
  fp = name if os.path.isabs(name) else os.path.join(train_dir, name)
  if os.path.isfile(fp + '.pt'):       # a TF-format export (ADVOC_EXPORT_TF_CKPT=1) sits next to the native file
    return fp + '.pt'
  from advoc_amd import tf_checkpoint
  if os.path.isfile(fp) or tf_checkpoint.is_tf_checkpoint(fp):
    return fp
  return None


def _prune(train_dir, pattern, keep):
  """Keeps the newest `keep` files matching `pattern` (step number = the integer before the extension), like
  tf.train.Saver(max_to_keep=...) does for the reference (default 5; its eval saver keeps 1,
  train_evaluate.py:150)."""
  import glob
  import re

  def step_of(fp):
    m = re.search(r'-(\d+)\.pt$', fp)
    return int(m.group(1)) if m else -1
  fps = sorted(glob.glob(os.path.join(train_dir, pattern)), key=step_of)
  for fp in fps[:-keep] if keep > 0 else []:
    # the TensorFlow bundle of the same step (ADVOC_EXPORT_TF_CKPT=1) goes with its .pt: max_to_keep bounds disk use
    for twin in glob.glob(fp[:-3] + '.index') + glob.glob(fp[:-3] + '.data-*'):
      try:
        os.remove(twin)
      except OSError:
        pass
    try:
      os.remove(fp)
    except OSError:
      pass


def save_checkpoint(train_dir, model, generator_only=False, name=None, max_to_keep=5):
  state = {'model': {k: v.cpu() for k, v in model.state_dict().items()}, 'step': model.step}
  if generator_only:
    state['model'] = {k: v for k, v in state['model'].items()
                      if k.startswith('generator') or k == 'global_step'}
  else:
    tensors, steps = model.optimizer_state()
    state['optimizer'] = {k: v.cpu() for k, v in tensors.items()}
    state['optimizer_steps'] = steps
  name = name or 'model.ckpt-%d.pt' % model.step
  tmp = os.path.join(train_dir, name + '.tmp')
  torch.save(state, tmp)
  os.replace(tmp, os.path.join(train_dir, name))
  index_text = name
  if os.environ.get('ADVOC_EXPORT_TF_CKPT') == '1' and name.endswith('.pt'):
    # also as a TensorFlow tensor bundle + CheckpointState index, so that the reference (tf.train.Saver.restore /
    # tf.train.latest_checkpoint, train_evaluate.py:60-66,131) can load weights trained here
    prefix = export_tf_checkpoint(os.path.join(train_dir, name[:-3]), model, generator_only=generator_only)
    base = os.path.basename(prefix)
    index_text = 'model_checkpoint_path: "{0}"\nall_model_checkpoint_paths: "{0}"\n'.format(base)
  with open(os.path.join(train_dir, 'checkpoint.tmp'), 'w') as f:
    f.write(index_text)
  os.replace(os.path.join(train_dir, 'checkpoint.tmp'), os.path.join(train_dir, 'checkpoint'))
  stem = name[:name.rfind('-') + 1] if '-' in name else None
  if stem and max_to_keep:
    _prune(train_dir, stem + '*.pt', max_to_keep)
  return os.path.join(train_dir, name)


def export_tf_checkpoint(prefix, model, generator_only=False):
  """Writes the model as a TensorFlow tensor bundle `prefix`.index / .data-00000-of-00001 (advoc_amd.tf_checkpoint):
  variables under their TF names and layouts, `global_step` (int64), and -- unless generator_only -- the Adam slots as
  tf.train.AdamOptimizer names them (`<var>/Adam`, `<var>/Adam_1`, `beta1_power`, `beta2_power` for the optimizer
  built first, the generator's at advoc_model.py:250-255, `..._1` for the discriminator's).  The slot names are
  inferred from TF's conventions, not verified against a TF-written checkpoint (none is reachable here); minimize() order there: generator first.  Returns the
  prefix."""
  import numpy as np
  from advoc_amd import tf_checkpoint
  sd = model.state_dict()
  out = {}
  for k, v in sd.items():
    if k == 'global_step':
      continue
    if generator_only and not k.startswith('generator'):
      continue
    out[k] = v.detach().cpu().numpy().astype(np.float32)
  out['global_step'] = np.asarray(int(model.step), dtype=np.int64)
  if not generator_only and getattr(model, '_built', None):
    st = model._built
    for net, suffix in (('g', ''), ('d', '_1')):
      arena = st[net + '_arena']
      for slot, tf_slot in (('_m', 'Adam'), ('_v', 'Adam_1')):
        for k, v in arena.views(st[net + slot]).items():
          out['%s/%s' % (k, tf_slot)] = v.detach().cpu().numpy().astype(np.float32)
      t = st[net + '_t']
      out['beta1_power' + suffix] = np.asarray(model._beta1 ** t, dtype=np.float32)
      out['beta2_power' + suffix] = np.asarray(model._beta2 ** t, dtype=np.float32)
  tf_checkpoint.write_checkpoint(prefix, out)
  return prefix


def restore_checkpoint(fp, model, with_optimizer=True):
  """Restores a checkpoint written by save_checkpoint, or a TensorFlow checkpoint prefix
  (`model.ckpt-N` with `.index` / `.data-*` next to it, e.g. the reference's published vocoders):
  variables are matched by their TF names; TF Adam slots are not imported."""
  from advoc_amd import tf_checkpoint
  if not os.path.isfile(fp) and tf_checkpoint.is_tf_checkpoint(fp):
    model.build()
    loaded, missing, step = tf_checkpoint.load_into_model(fp, model)
    if missing:
      print('TF checkpoint {}: {} variables not found (left at their initial values): {}'.format(
          fp, len(missing), ', '.join(missing[:4]) + (' ...' if len(missing) > 4 else '')))
    model.step = step
    return model.step
  state = torch.load(fp, map_location='cpu')
  model.load_state_dict(state['model'])
  model.step = int(state.get('step', 0))
  if with_optimizer and 'optimizer' in state:
    dev = model._built['g_m'].device
    model.load_optimizer_state({k: v.to(dev) for k, v in state['optimizer'].items()},
                               tuple(state['optimizer_steps']))
  return model.step


def _loader(fps, args, model, batch_size, training, first_only=None):
  """train: train_evaluate.py:35-54; eval: :95-114 (slice_first_only from the datacfg); infer: :212-230
  (no slice_first_only argument there: the loader default False applies)."""
  from advoc_amd.loader import decode_extract_and_batch
  return decode_extract_and_batch(
      fps,
      batch_size=batch_size,
      slice_len=model.subseq_len,
      audio_fs=model.audio_fs,
      audio_mono=True,
      audio_normalize=args.data_normalize,
      decode_fastwav=args.data_fastwav,
      decode_parallel_calls=4,
      extract_type='magspec',
      extract_parallel_calls=8,
      repeat=training,
      shuffle=training,
      shuffle_buffer_size=512 if training else None,
      slice_first_only=args.data_slice_first_only if first_only is None else first_only,
      slice_randomize_offset=args.data_slice_randomize_offset if training else False,
      slice_overlap_ratio=args.data_slice_overlap_ratio if training else 0.,
      slice_pad_end=args.data_slice_pad_end if training else True,
      prefetch_size=batch_size * 8 if training else None,
      prefetch_gpu_num=0 if training else None)


def train(fps, args):
  from advoc_amd.model import Modes
  from advoc_amd.parallel import DataParallel
  from advoc_amd.spectral_util import SpectralUtil
  dp = DataParallel().init_from_env()
  model = _make_model(args, Modes.TRAIN)
  model.build(seed=0)
  dp.attach(model)

  pipe = _loader(dp.shard(sorted(fps)) if dp.enabled else fps, args, model, model.train_batch_size, True)
  spectral = SpectralUtil(n_mels=model.n_mels, fs=model.audio_fs)

  def feed():
    x_magspec, x_wav = pipe.next()
    x_melspec = spectral.mag_to_mel_linear_spec(x_magspec)
    x_inverted_magspec = spectral.mel_linear_to_mag_spec(x_melspec, transform='inverse')
    return x_inverted_magspec, x_magspec, x_wav, x_melspec
  model(feed)

  ckpt = latest_checkpoint(args.train_dir)
  if ckpt is not None:
    print('Restoring from {}'.format(ckpt))
    restore_checkpoint(ckpt, model)
  dp.broadcast_parameters(model)

  last_ckpt = last_summary = time.time()
  log = open(os.path.join(args.train_dir, 'summaries.jsonl'), 'a') if dp.rank == 0 else None
  from advoc_amd.tb_events import EventWriter
  events = EventWriter(args.train_dir) if dp.rank == 0 else None
  _step = model.step
  while _step < args.max_steps:
    _step = model.train_loop()
    now = time.time()
    if dp.rank == 0 and now - last_summary >= args.train_summary_every_nsecs:
      rec = dict(step=_step, time=now, **model.losses())
      refits = model.image_refits()
      if refits:       # tensors that jumped out of the one-pass scale window were re-imaged exactly (no clamped step)
        rec['operand_image_refits'] = refits
        rec['operand_image_saturations'] = model.image_saturations()
      log.write(json.dumps(rec) + '\n')
      log.flush()
      events.add_scalars(model.losses(), _step, wall_time=now)      # tags as advoc_model.py:272-275
      if True:        # image / audio summaries (advoc_model.py:258-281)
        images, audio = model.media_summaries()
        events.add_images(images, _step, wall_time=now)
        events.add_audio(audio, _step, model.audio_fs, wall_time=now)
      last_summary = now
    if dp.rank == 0 and now - last_ckpt >= args.train_ckpt_every_nsecs:
      save_checkpoint(args.train_dir, model)
      last_ckpt = now
  if dp.rank == 0:
    save_checkpoint(args.train_dir, model)
    log.write(json.dumps(dict(step=_step, time=time.time(), **model.losses())) + '\n')
    log.close()
  pipe.close()
  print('Done!')


def evaluate_checkpoint(fps, args, model, ckpt_fp):
  """Mean of the per-batch L1 between target and generated magnitude spectrograms over the
  whole dataset (reference train_evaluate.py:137,165-182)."""
  from advoc_amd.spectral_util import SpectralUtil
  restore_checkpoint(ckpt_fp, model, with_optimizer=False)
  spectral = SpectralUtil(n_mels=model.n_mels, fs=model.audio_fs)
  pipe = _loader(fps, args, model, model.eval_batch_size, False)
  all_l1 = []
  for x_magspec, _ in pipe.batches():
    x_melspec = spectral.mag_to_mel_linear_spec(x_magspec)
    x_inv = spectral.mel_linear_to_mag_spec(x_melspec, transform='inverse')
    l1, _ = model.l1_eval(x_inv, x_magspec)
    all_l1.append(l1)
  pipe.close()
  return float(np.mean(all_l1)) if all_l1 else float('nan'), len(all_l1)


def eval(fps, args, poll=True):   # noqa: A001  (name kept from the reference)
  from advoc_amd.model import Modes
  eval_dir = os.path.join(args.train_dir, 'eval_{}'.format(args.eval_dataset_name)
                          if args.eval_dataset_name is not None else 'eval_valid')
  os.makedirs(eval_dir, exist_ok=True)
  model = _make_model(args, Modes.EVAL)
  model.build(batch_size=model.eval_batch_size)
  ckpt_fp = None
  best = np.inf
  while True:
    latest = latest_checkpoint(args.train_dir)
    if latest is not None and latest != ckpt_fp:
      ckpt_fp = latest
      print('Evaluating {}'.format(ckpt_fp))
      l1, n = evaluate_checkpoint(fps, args, model, ckpt_fp)
      with open(os.path.join(eval_dir, 'summaries.jsonl'), 'a') as f:
        f.write(json.dumps(dict(step=model.step, gen_loss_L1=l1, batches=n)) + '\n')
      from advoc_amd.tb_events import EventWriter
      ew = EventWriter(eval_dir)
      ew.add_scalars({'gen_loss_L1': l1}, model.step)               # train_evaluate.py:143-145
      ew.close()
      if l1 < best:
        # the reference never updates its best value (train_evaluate.py:153,184-186), so it saves
        # every evaluated checkpoint; here "best" means best.
        best = l1
        save_checkpoint(eval_dir, model, generator_only=True, name='best_gen_loss_l1-%d.pt' % model.step,
                        max_to_keep=1)
        print('Saved best gen loss l1!')
      print('Done!')
    if not poll:
      return best
    time.sleep(1)


def infer(fps, args):
  """Runs the generator over the dataset from --infer_ckpt_path (or the latest checkpoint) and
  writes, under WORK_DIR/infer_*, the input / target / generated magnitude spectrograms (.npy) and
  the three audio clips the reference puts into its TensorBoard audio summaries
  (train_evaluate.py:262-281: the real waveform, the vocoded pseudo-inverse "heuristic" and the vocoded
  generator output) as PCM16 .wav files.  Phase comes from LWS as in the reference (advoc_model.py:262-281 ->
  spectral_util.py:45-50), the GPU restatement of advoc_amd.spectral."""
  from advoc_amd.model import Modes
  from advoc_amd.spectral_util import SpectralUtil
  infer_dir = os.path.join(args.train_dir, 'infer_{}'.format(args.infer_dataset_name)
                           if args.infer_dataset_name is not None else 'infer_valid')
  os.makedirs(infer_dir, exist_ok=True)
  model = _make_model(args, Modes.INFER)
  model.build(batch_size=args.infer_batch_size)
  ckpt_fp = args.infer_ckpt_path or latest_checkpoint(args.train_dir)
  if ckpt_fp is None:
    raise ValueError('no checkpoint to infer from')
  print('Infereing From {}'.format(ckpt_fp))
  restore_checkpoint(ckpt_fp, model, with_optimizer=False)
  spectral = SpectralUtil(n_mels=model.n_mels, fs=model.audio_fs)
  pipe = _loader(fps, args, model, args.infer_batch_size, False, first_only=False)
  from advoc_amd import spectral as S
  from advoc_amd.audioio import save_as_wav
  for i, (x_magspec, x_wav) in enumerate(pipe.batches()):
    x_melspec = spectral.mag_to_mel_linear_spec(x_magspec)
    x_inv = spectral.mel_linear_to_mag_spec(x_melspec, transform='inverse')
    gen = model.build_generator(x_inv)
    np.save(os.path.join(infer_dir, 'batch%06d_gen_magspec.npy' % i), gen.cpu().numpy())
    np.save(os.path.join(infer_dir, 'batch%06d_target_magspec.npy' % i), x_magspec.cpu().numpy())
    np.save(os.path.join(infer_dir, 'batch%06d_input_magspec.npy' % i), x_inv.cpu().numpy())
    both = torch.cat([x_inv[..., 0], gen[..., 0]], dim=0).abs().contiguous()       # [2b, T, 513]
    wav = S.lws_batch(both, spectral.NFFT, spectral.NHOP)
    b = x_inv.shape[0]
    for j in range(b):
      stem = os.path.join(infer_dir, 'batch%06d_clip%02d' % (i, j))
      save_as_wav(stem + '_real.wav', int(model.audio_fs), x_wav[j].cpu().numpy().reshape(-1, 1, 1))
      save_as_wav(stem + '_heuristic.wav', int(model.audio_fs), wav[j].cpu().numpy().reshape(-1, 1, 1))
      save_as_wav(stem + '_generated.wav', int(model.audio_fs), wav[b + j].cpu().numpy().reshape(-1, 1, 1))
  pipe.close()
  print('Done!')


def build_parser():
  from argparse import ArgumentParser
  parser = ArgumentParser()
  parser.add_argument('mode', type=str, choices=['train', 'eval', 'infer'])
  parser.add_argument('train_dir', type=str)
  parser.add_argument('--data_cfg', type=str, help='Path to dataset configuration')
  parser.add_argument('--model_type', type=str, choices=['regular', 'small'])
  parser.add_argument('--data_dir', type=str, required=True)
  parser.add_argument('--model_overrides', type=str)
  parser.add_argument('--train_ckpt_every_nsecs', type=int)
  parser.add_argument('--max_steps', type=int)
  parser.add_argument('--infer_batch_size', type=int)
  parser.add_argument('--train_summary_every_nsecs', type=int)
  parser.add_argument('--eval_dataset_name', type=str)
  parser.add_argument('--eval_wavenet_meta_fp', type=str)
  parser.add_argument('--eval_wavenet_ckpt_fp', type=str)
  parser.add_argument('--infer_dataset_name', type=str)
  parser.add_argument('--infer_ckpt_path', type=str)
  parser.set_defaults(
      mode=None, train_dir=None, model_type='regular', data_dir=None, model_overrides=None,
      train_ckpt_every_nsecs=360, train_summary_every_nsecs=60, max_steps=100000, infer_batch_size=1,
      eval_dataset_name=None, eval_wavenet_meta_fp=None, eval_wavenet_ckpt_fp=None,
      infer_dataset_name=None, infer_ckpt_path=None)
  return parser


def parse_data_cfg(path, args):
  """`key,value` lines -> args.data_<key>, int if it parses as int else float (:375-382).
  Blank lines and lines starting with `#` are ignored (an extension: the reference's files have neither)."""
  with open(path, 'r') as f:
    for line in f.read().strip().splitlines():
      line = line.strip()
      if not line or line.startswith('#'):
        continue
      k, v = line.split(',')
      try:
        v = int(v)
      except ValueError:
        v = float(v)
      setattr(args, 'data_' + k, v)
  return args


def main(argv=None):
  args = build_parser().parse_args(argv)
  parse_data_cfg(args.data_cfg, args)
  if not os.path.isdir(args.train_dir):
    os.makedirs(args.train_dir)
  fps = glob.glob(os.path.join(args.data_dir, '*'))
  print('Found {} audio files'.format(len(fps)))
  if args.mode == 'train':
    train(fps, args)
  elif args.mode == 'eval':
    eval(fps, args)
  elif args.mode == 'infer':
    infer(fps, args)
  else:
    raise NotImplementedError()


if __name__ == '__main__':
  main()
