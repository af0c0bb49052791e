"""CLI surface (argparse table, datacfg parsing, model overrides) -- CPU -- and an end-to-end
train / resume / eval / infer run on synthetic WAVs -- GPU (BASELINE.json configs[0] shape:
AdVoc-small on 8 wavs, 1+ steps)."""
import json
import os

import numpy as np
import pytest

gpu = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_parser_defaults_match_reference():
  from advoc_amd.train_evaluate import build_parser, parse_data_cfg
  a = build_parser().parse_args(['train', '/tmp/x', '--data_dir', '/d', '--data_cfg', 'c'])
  assert (a.mode, a.train_dir, a.model_type, a.max_steps) == ('train', '/tmp/x', 'regular', 100000)
  assert (a.train_ckpt_every_nsecs, a.train_summary_every_nsecs, a.infer_batch_size) == (360, 60, 1)
  assert a.model_overrides is None and a.infer_ckpt_path is None and a.eval_dataset_name is None
  with pytest.raises(SystemExit):
    build_parser().parse_args(['train', '/tmp/x'])          # --data_dir is required
  with pytest.raises(SystemExit):
    build_parser().parse_args(['finetune', '/tmp/x', '--data_dir', '/d'])
  parse_data_cfg(os.path.join(ROOT, 'datacfg', 'ljspeech.txt'), a)
  assert a.data_sample_rate == 22050 and isinstance(a.data_sample_rate, int)
  assert a.data_slice_overlap_ratio == 0.25 and a.data_fastwav == 1 and a.data_slice_pad_end == 1
  parse_data_cfg(os.path.join(ROOT, 'datacfg', 'sc09.txt'), a)
  assert a.data_sample_rate == 16000 and a.data_normalize == 1 and a.data_slice_first_only == 1
  assert a.data_slice_overlap_ratio == 0.0 and isinstance(a.data_slice_overlap_ratio, float)


def test_override_model_attrs():
  from advoc_amd.model import Advoc, AdvocSmall, Modes, override_model_attrs
  m, summary = override_model_attrs(AdvocSmall(Modes.TRAIN), 'train_batch_size=32,use_batchnorm=t,gan_weight=0.5')
  assert m.train_batch_size == 32 and m.use_batchnorm is True and m.gan_weight == 0.5
  lines = summary.split('\n')
  keys = [l.split(',')[0] for l in lines]
  assert keys == sorted(keys) and 'ngf,32' in lines and 'train_batch_size,32' in lines
  assert 'num_enc_layers,4' in lines and 'generator_type,pix2pix' in lines
  m2, s2 = override_model_attrs(Advoc(Modes.EVAL), None)
  assert 'ngf,64' in s2.split('\n') and m2.eval_batch_size == 1
  m3, _ = override_model_attrs(Advoc(Modes.EVAL), '   ')
  assert m3.ngf == 64
  with pytest.raises(AttributeError):
    override_model_attrs(Advoc(Modes.EVAL), 'nonexistent=1')


def test_unsupported_ablations_fail_loudly():
  from advoc_amd.model import Advoc, Modes
  from advoc_amd.model import override_model_attrs
  for ov in ('generator_type=linear', 'separable_conv=True', 'ngf=48', 'ndf=16'):
    m, _ = override_model_attrs(Advoc(Modes.TRAIN), ov)
    with pytest.raises(NotImplementedError):
      m._check_supported()
  # a clip length whose halvings and doublings do not retrace each other cannot be built in the
  # reference either (tf.concat shape error at advoc_model.py:137)
  for bad in ('subseq_len=100', 'subseq_len=16'):          # 16: the PatchGAN would have no output rows
    m, _ = override_model_attrs(Advoc(Modes.TRAIN), bad)
    with pytest.raises(ValueError):
      m._check_supported()
  # shorter power-of-two clips use the (1,2)-stride layers (advoc_model.py:109-116): supported
  for n, want in ((64, 2), (32, 3), (256, 0)):
    m, _ = override_model_attrs(Advoc(Modes.TRAIN), 'subseq_len=%d' % n)
    m._check_supported()
    assert sum(1 for st in m._encoder_strides() if st == (1, 2)) == want


@gpu
def test_train_resume_eval_infer_end_to_end(hip, tmp_path, capsys):
  from advoc_amd import train_evaluate as TE
  from advoc_amd.audioio import save_as_wav
  data = tmp_path / 'wavs'
  data.mkdir()
  rng = np.random.default_rng(0)
  for i in range(8):
    n = int(22050 * (3 + i % 3))
    t = np.arange(n) / 22050.
    x = 0.4 * np.sin(2 * np.pi * (150 + 60 * i) * t) * (1 + 0.3 * np.sin(2 * np.pi * 3 * t)) + 0.05 * rng.uniform(-1, 1, n)
    save_as_wav(str(data / ('c%d.wav' % i)), 22050, x.astype(np.float32)[:, None, None])
  work = str(tmp_path / 'work')
  common = ['--data_cfg', os.path.join(ROOT, 'datacfg', 'ljspeech.txt'), '--data_dir', str(data),
            '--model_type', 'small', '--model_overrides', 'train_batch_size=2,subseq_len=64']
  TE.main(['train', work] + common + ['--max_steps', '3', '--train_summary_every_nsecs', '0'])
  out = capsys.readouterr().out
  assert 'Found 8 audio files' in out and 'Done!' in out and ('-' * 80) in out and 'ngf,32' in out
  assert os.path.isfile(os.path.join(work, 'model.ckpt-3.pt'))
  recs = [json.loads(l) for l in open(os.path.join(work, 'summaries.jsonl'))]
  assert recs[-1]['step'] == 3 and all(np.isfinite(r['gen_loss_total']) for r in recs)
  # the same scalars as a TensorBoard event file (tags of advoc_model.py:263-266)
  import glob as _glob
  from advoc_amd.tb_events import read_events
  ev = read_events(_glob.glob(os.path.join(work, 'events.out.tfevents.*'))[0])
  assert ev[-1][0] == 3 and abs(ev[-1][1]['gen_loss_total'] - recs[-1]['gen_loss_total']) < 1e-4 * abs(recs[-1]['gen_loss_total'])
  # image / audio summaries of advoc_model.py:258-281 in the same file
  media = read_events(_glob.glob(os.path.join(work, 'events.out.tfevents.*'))[0], kinds=('image', 'audio'))
  tags = set(t for _, v in media for t in v)
  assert {'generated_magspec/image/0', 'target_magspec/image/0', 'input_magspec/image/0', 'input_melspec/image/0',
          'gen_audio/audio/0', 'input_audio/audio/0', 'target_audio/audio/0', 'target_x_wav/audio/0'} <= tags
  im = [v for _, v in media if 'input_melspec/image/0' in v][0]
  assert (im['input_melspec/image/0']['height'], im['input_magspec/image/0']['height']) == (80, 513)
  # resume: continues from step 3
  TE.main(['train', work] + common + ['--max_steps', '5'])
  assert 'Restoring from' in capsys.readouterr().out
  assert open(os.path.join(work, 'checkpoint')).read().strip() == 'model.ckpt-5.pt'
  # eval once
  args = TE.build_parser().parse_args(['eval', work] + common)
  TE.parse_data_cfg(args.data_cfg, args)
  import glob
  best = TE.eval(glob.glob(os.path.join(str(data), '*')), args, poll=False)
  assert np.isfinite(best) and best > 0
  assert any(f.startswith('best_gen_loss_l1-5') for f in os.listdir(os.path.join(work, 'eval_valid')))
  # infer
  TE.main(['infer', work] + common + ['--infer_batch_size', '2'])
  files = os.listdir(os.path.join(work, 'infer_valid'))
  assert any(f.endswith('gen_magspec.npy') for f in files)
  g = np.load(os.path.join(work, 'infer_valid', sorted(f for f in files if f.endswith('gen_magspec.npy'))[0]))
  assert g.shape == (2, 64, 513, 1) and np.isfinite(g).all()
  # the reference's three audio summaries, as WAV files
  from advoc_amd.audioio import decode_audio
  for kind in ('real', 'heuristic', 'generated'):
    fp = os.path.join(work, 'infer_valid', 'batch000000_clip00_%s.wav' % kind)
    fs, wav = decode_audio(fp, fastwav=True)
    assert fs == 22050 and wav.shape[0] in (64 * 256, 63 * 256 + 1024) and np.isfinite(wav).all()
