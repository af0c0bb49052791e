#!/usr/bin/env python
"""Headline benchmark: AdVoc G+D train step throughput in mel-frames/s (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (config.workload): BASELINE.json configs[1] -- AdVoc-small, LJSpeech geometry
(22.05 kHz, nfft 1024 / hop 256, 256-frame clips), batch 32 per GPU.  One "step" = one reference
train_loop (models/advoc/advoc_model.py:285-289): a discriminator update on one batch and a
generator update on the NEXT batch, each batch going waveform -> |STFT| -> mel -> pseudo-inverse
on the GPU first (advoc/loader.py:116-128 + models/advoc/train_evaluate.py:55-56).  Inputs are
synthetic waveforms (uniform noise + 3 sinusoids, seeded) already resident in HBM; weights are
N(0, 0.02) random init; dropout masks come from the on-device Philox stream.

Every 4th timed step carries the per-launch HIP-event instrumentation behind `roofline` and therefore runs
on one stream; the other steps run the weight-gradient kernels on a side stream as the trainer does by
default (advoc_amd.model._wgrad_ctx), about 3 % faster.  --no-launch-timing times uninstrumented steps only.

value = (global batch x 256 frames x steps) / wall time: the conservative accounting (the step
consumes TWO batches; only one is counted).  N > 1: batch sharded 32 per GPU (weak scaling), RCCL
all-reduce of the D and G gradient arenas.

Extra objects on the JSON line:
  roofline      the kernel instance with the largest total time in the timed region, timed per
                launch with HIP events on the launch stream (every 4th step of the timed region
                is instrumented): algorithmic flops / measured time against the roof of the
                pipe the kernel runs on: dense fp32 MFMA (157.3 TFLOP/s), or, for the split-bf16
                kernels, dense bf16 MFMA / 6 partial products (416.7 TFLOP/s algorithmic).
                `traffic` = L2-miss bytes per launch of that kernel from the committed rocprofv3
                counter passes of this same command (profiles/r01_traffic.json; FETCH_SIZE
                doubled per the gfx950 correction + WRITE_SIZE), null when not recorded.
  extractor     the HBM-bound leg: waveform -> |STFT| (stft1024_kernel) timed per launch with HIP
                events; algorithmic bytes = 790 528 per clip against 8 TB/s.
  inference     vocoded clips/s: mel -> pseudo-inverse -> generator forward on 256-frame chunks
                (scripts/spectrogram_advoc.py:80-94 semantics, batched; phase estimation not
                included), plus `joint_sc09`: z -> MelspecGAN -> AdVoc -> Griffin-Lim waveform
                (BASELINE configs[4] on one GPU).  Measured after the timed region; not part of `value`.
  cpu_baseline  the torch-CPU restatement of the reference graph (oracle/, "port") timed on this
                box's host cores on a bounded sample (rank 0, N = 1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
  sys.path.insert(0, ROOT)

FP32_MFMA_PEAK_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md, "Peak FP32 (matrix)"
# Kernels of the split-bf16 path (template flag `true>` at the end of the instance name) run every fp32
# product as SIX bf16 MFMA products (x = x0 + x1 + x2 exactly; the three smallest of the nine partial
# products are dropped): their roof in ALGORITHMIC fp32 flops is the dense bf16 peak / 6.
BF16_MFMA_PEAK_TFLOPS = 2500.0  # ibid., "BF16/F16 ~2.5 PF dense"
X6_PEAK_TFLOPS = BF16_MFMA_PEAK_TFLOPS / 6.0
HBM_PEAK_GBS = 8000.0           # ibid., HBM3E peak BW
CLIP_FRAMES = 256
CLIP_SAMPLES = (CLIP_FRAMES - 1) * 256 + 1024   # 66304


def synth_waveforms(batch, seed, device):
  """uniform(-0.5, 0.5) noise + 3 seeded sinusoids per clip."""
  import torch
  g = torch.Generator().manual_seed(seed)
  x = torch.rand(batch, CLIP_SAMPLES, generator=g) - 0.5
  t = torch.arange(CLIP_SAMPLES, dtype=torch.float32) / 22050.0
  for _ in range(3):
    f = 100.0 + 4000.0 * torch.rand(batch, 1, generator=g)
    a = 0.1 + 0.2 * torch.rand(batch, 1, generator=g)
    x = x + a * torch.sin(2 * 3.141592653589793 * f * t[None, :])
  return x.reshape(batch, CLIP_SAMPLES, 1, 1).to(device)


def cpu_baseline(model_small=True, budget_s=20.0):
  """Reference-equivalent CPU restatement (oracle/advoc_torch.py + oracle/spectral_np.py),
  one train_loop = D update + G update, batch 8 (reference default, advoc_model.py:18)."""
  import numpy as np
  import torch
  from oracle import advoc_torch as A
  from oracle import spectral_np as S
  B = 8
  cfg = A.Config(small=model_small)
  tr = A.Trainer(cfg, seed=0)
  W = S.create_mel_filterbank(22050, 1024, fmin=125, fmax=7600, n_mels=80).astype(np.float32)
  Wi = S.create_inverse_mel_filterbank(22050, 1024, fmin=125, fmax=7600, n_mels=80).astype(np.float32)
  rng = np.random.default_rng(0)

  def make_batch():
    wav = rng.uniform(-0.5, 0.5, size=(B, CLIP_SAMPLES, 1, 1)).astype(np.float32)
    mag = np.abs(S.stft_tf(wav, 1024, 256, pad_end=False)).astype(np.float32)
    mel = S.mag_to_mel_linear_spec(mag, W)
    inv = S.mel_linear_to_mag_spec(mel, Wi)
    return torch.from_numpy(inv), torch.from_numpy(mag)
  masks = A.make_dropout_masks(cfg, B, seed=1)
  tr.train_loop(make_batch(), make_batch(), masks, masks)      # warm-up (thread pools, allocator)
  n, t_total = 0, 0.0
  while n == 0 or t_total < budget_s:
    t0 = time.perf_counter()
    tr.train_loop(make_batch(), make_batch(), masks, masks)
    t_total += time.perf_counter() - t0
    n += 1
  return dict(value=B * CLIP_FRAMES * n / t_total, unit='mel-frames/s', cores=torch.get_num_threads(),
              kind='port',
              sample='%d train_loop iterations (1 D + 1 G update each) of AdVoc-small at batch %d, '
                     'STFT/mel in numpy, convs in torch-CPU fp32' % (n, B))


def extractor_leg(torch, spectral, wav, launches=30):
  """waveform -> |STFT| alone through the C ABI with a preallocated output: GB/s of algorithmic
  traffic (waveform read once + |X| written once).  Measured at the training batch
  (one launch = 32 x 256 frames: too small to fill 256 CUs) and at 512 clips per launch (what the
  loader's whole-file extraction looks like)."""
  from advoc_amd import _lib
  lib = _lib.load()
  win = spectral._device_window(1024, 256)
  tw = spectral._device_twiddle(1024)

  def run(clips):
    x = wav[:, :, 0, 0].repeat((clips + wav.shape[0] - 1) // wav.shape[0], 1)[:clips].contiguous()
    out = torch.empty(clips, CLIP_FRAMES, 513, dtype=torch.float32, device=x.device)
    call = lambda: _lib.check(lib.advoc_stft_mag_f32(_lib.ptr(x), clips, x.shape[1], _lib.ptr(win), _lib.ptr(tw),   # noqa: E731
                                                     1024, 256, CLIP_FRAMES, _lib.ptr(out), _lib.stream()), 'stft')
    for _ in range(3):
      call()
    evs = []
    for _ in range(launches):
      e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
      e0.record()
      call()
      e1.record()
      evs.append((e0, e1))
    torch.cuda.synchronize()
    ms = sum(a.elapsed_time(b) for a, b in evs) / launches
    nbytes = clips * (CLIP_SAMPLES * 4 + CLIP_FRAMES * 513 * 4)
    return ms, nbytes
  ms_b, bytes_b = run(wav.shape[0])
  ms_l, bytes_l = run(512)
  gbs = bytes_l / (ms_l * 1e-3) / 1e9
  return dict(kernel='stft1024_kernel<false>', bound='hbm', achieved=gbs, peak=HBM_PEAK_GBS, unit='GB/s',
              frac=gbs / HBM_PEAK_GBS, clips_per_launch=512, bytes_per_launch=bytes_l, avg_launch_ms=ms_l,
              frames_per_s=512 * CLIP_FRAMES / (ms_l * 1e-3),
              at_train_batch=dict(clips_per_launch=int(wav.shape[0]), avg_launch_ms=ms_b,
                                  achieved=bytes_b / (ms_b * 1e-3) / 1e9,
                                  frames_per_s=wav.shape[0] * CLIP_FRAMES / (ms_b * 1e-3)))


def inference_leg(torch, model_cls, Modes, su, mel, iters=10):
  """clips/s of mel -> magnitude through the generator (INFER mode, dropout active as in the
  reference), batch = the training batch."""
  m = model_cls(Modes.INFER)
  B = mel.shape[0]
  m.build(batch_size=B, seed=0)
  for _ in range(2):
    m.build_generator(su.mel_linear_to_mag_spec(mel))
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  for _ in range(iters):
    m.build_generator(su.mel_linear_to_mag_spec(mel))
  torch.cuda.synchronize()
  dt = time.perf_counter() - t0
  # waveform synthesis on top: Griffin-Lim, 60 iterations (advoc/spectral.py:294-311), all clips at once
  from advoc_amd import spectral
  mag = m.build_generator(su.mel_linear_to_mag_spec(mel))[..., 0].abs().contiguous()
  u = torch.rand(mag.shape, device=mag.device)
  spectral.griffin_lim_batch(mag, 1024, 256, 2, u)
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  spectral.griffin_lim_batch(mag, 1024, 256, 60, u)
  torch.cuda.synchronize()
  dt_gl = time.perf_counter() - t0
  return dict(value=B * iters / dt, unit='vocoded 256-frame clips/s', batch=B, ms_per_batch=dt / iters * 1e3,
              with_gl60_clips_per_s=B / (dt / iters + dt_gl), gl60_ms_per_batch=dt_gl * 1e3,
              note='value: mel -> pinv projection -> generator forward (magnitudes); with_gl60: plus 60 '
                   'Griffin-Lim iterations (iSTFT/STFT/projection kernels) to a waveform.  The reference '
                   'uses LWS for phase (third-party, not restated).')


def joint_leg(torch, n=64, iters=3):
  """BASELINE configs[4] on one GPU: z -> MelspecGAN generator -> mel [64 x 80] -> AdVoc (full model at
  subseq_len 64, its (1,2)-stride layers) -> Griffin-Lim (60 iterations) -> 16 kHz waveform; random
  weights (no checkpoints are reachable), synthetic z.  Samples per second, end to end on the GPU."""
  from advoc_amd.infer import vocode_batch
  from advoc_amd.melspecgan import MelspecGANGenerator
  from advoc_amd.model import Advoc, Modes
  G = MelspecGANGenerator(dim=64)
  voc = Advoc(Modes.INFER)
  voc.subseq_len = 64
  voc.audio_fs = 16000
  voc.build(batch_size=2 * n, seed=0)
  z = torch.randn(n, 100, generator=torch.Generator().manual_seed(0))

  def run():
    mel = G(z, denorm=True)
    return vocode_batch(voc, mel, phase_estimation='gl60', chunk_batch=2 * n)[1]
  run()
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  for _ in range(iters):
    wav = run()
  torch.cuda.synchronize()
  dt = (time.perf_counter() - t0) / iters
  return dict(value=n / dt, unit='generated 64-frame clips/s (z -> 16 kHz waveform, Griffin-Lim 60)', batch=n,
              ms_per_batch=dt * 1e3, samples_per_clip=int(wav.shape[1]),
              note='MelspecGAN G + AdVoc-full(subseq_len 64) + GL60, random weights; the reference uses LWS')


def recorded_traffic(kernel, model, batch):
  """L2-miss bytes per launch from the committed counter passes (tools/pmc_summary.py), if they
  were taken on this workload."""
  fp = os.path.join(ROOT, 'profiles', 'r01_traffic.json')
  if not os.path.exists(fp):
    return None
  rec = json.load(open(fp))
  meta = rec.get('_workload', {})
  if meta.get('model') != model or meta.get('batch') != batch:
    return None
  row = rec.get(kernel)
  return row['traffic_bytes'] if row else None


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--gpus', type=int, default=1)
  ap.add_argument('--steps', type=int, default=20)
  ap.add_argument('--warmup', type=int, default=3)
  ap.add_argument('--model', choices=['small', 'regular'], default='small')
  ap.add_argument('--batch', type=int, default=32, help='clips per GPU')
  ap.add_argument('--no-cpu-baseline', action='store_true')
  ap.add_argument('--train-only', action='store_true',
                  help='skip the extractor / inference legs that run after the timed region (profiling runs: '
                       'keeps the per-kernel statistics to the train step)')
  ap.add_argument('--no-launch-timing', action='store_true',
                  help='skip per-launch HIP events (roofline object becomes null)')
  args = ap.parse_args()

  import torch
  from advoc_amd import conv, spectral
  from advoc_amd.model import Advoc, AdvocSmall, Modes
  from advoc_amd.parallel import DataParallel
  from advoc_amd.spectral_util import SpectralUtil

  dp = DataParallel().init_from_env()
  if dp.world_size != args.gpus:
    raise SystemExit('--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run' % (args.gpus, dp.world_size))
  dev = torch.device('cuda', dp.local_rank)
  torch.cuda.set_device(dev)

  B = args.batch
  model = (AdvocSmall if args.model == 'small' else Advoc)(Modes.TRAIN)
  model.train_batch_size = B
  model.build(batch_size=B, seed=0)
  dp.attach(model)
  dp.broadcast_parameters(model)
  su = SpectralUtil(n_mels=model.n_mels, fs=model.audio_fs)

  pool = [synth_waveforms(B, 1234 + 17 * dp.rank + i, dev) for i in range(4)]
  state = {'i': 0}

  def feed():
    wav = pool[state['i'] % len(pool)]
    state['i'] += 1
    mag = spectral.stft_magnitude(wav, 1024, 256, pad_end=False)          # [B,256,513,1]
    mel = su.mag_to_mel_linear_spec(mag)
    inv = su.mel_linear_to_mag_spec(mel)
    return inv, mag, wav, mel
  model(feed)

  for _ in range(args.warmup):
    model.train_loop()
  torch.cuda.synchronize()
  dp.barrier()
  torch.cuda.synchronize()

  prof = None
  if not args.no_launch_timing:
    prof = conv.LaunchProfiler()
  # per-launch HIP events cost ~0.6 ms of host time per fully instrumented step (3 %): instrument
  # every 4th step of the timed region; average launch durations do not depend on which steps
  sampled = [i for i in range(args.steps) if i % 4 == 0] if prof is not None else []
  t0 = time.perf_counter()
  for i in range(args.steps):
    conv.Layer.profiler = prof if (prof is not None and i % 4 == 0) else None
    model.train_loop()
  torch.cuda.synchronize()
  dp.barrier()
  torch.cuda.synchronize()
  elapsed = time.perf_counter() - t0
  conv.Layer.profiler = None
  elapsed = dp.max_over_ranks(elapsed)

  frames = B * dp.world_size * CLIP_FRAMES * args.steps
  value = frames / elapsed

  roofline = None
  if prof is not None:
    rows = prof.rows()
    name, r = max(rows.items(), key=lambda kv: kv[1]['ms'])
    mfma = 'mfma' in name or 'gather_gemm' in name
    if mfma:
      achieved = r['flops'] / (r['ms'] * 1e-3) / 1e12
      x6 = name.endswith(', true>') and (name.startswith('gather_gemm_kernel<') or name.startswith('wgrad_mfma_kernel<'))
      peak = X6_PEAK_TFLOPS if x6 else FP32_MFMA_PEAK_TFLOPS
      roofline = dict(bound='mfma', kernel=name, achieved=achieved, peak=round(peak, 1),
                      unit='TFLOP/s', frac=achieved / peak,
                      pipe=('bf16 MFMA, 6 partial products per fp32 product (2500 / 6 TFLOP/s algorithmic)' if x6
                            else 'fp32 MFMA'),
                      vs_fp32_mfma_peak=achieved / FP32_MFMA_PEAK_TFLOPS,
                      traffic=recorded_traffic(name, args.model, B),
                      algorithmic_bytes_per_launch=r['bytes'] / r['launches'],
                      launches=r['launches'], avg_launch_ms=r['ms'] / r['launches'],
                      flops_per_launch=r['flops'] / r['launches'],
                      share_of_step=r['ms'] / (elapsed * 1e3 * len(sampled) / args.steps),
                      instrumented_steps=len(sampled))
    else:
      achieved = r['bytes'] / (r['ms'] * 1e-3) / 1e9
      roofline = dict(bound='hbm', kernel=name, achieved=achieved, peak=HBM_PEAK_GBS, unit='GB/s',
                      frac=achieved / HBM_PEAK_GBS, traffic=recorded_traffic(name, args.model, B),
                      launches=r['launches'],
                      avg_launch_ms=r['ms'] / r['launches'],
                      share_of_step=r['ms'] / (elapsed * 1e3 * len(sampled) / args.steps), instrumented_steps=len(sampled))
    if dp.rank == 0 and os.environ.get('ADVOC_BENCH_VERBOSE'):
      tot = sum(v['ms'] for v in rows.values())
      for k, v in sorted(rows.items(), key=lambda kv: -kv[1]['ms']):
        tf = v['flops'] / max(v['ms'], 1e-9) / 1e9
        print('  %-44s launches %5d  %9.2f ms (%5.1f%%)  %7.2f TFLOP/s  %7.1f GB/s alg' % (
            k, v['launches'], v['ms'], 100 * v['ms'] / tot, tf, v['bytes'] / max(v['ms'], 1e-9) / 1e6),
            file=sys.stderr)
      print('  conv-stack launches total %.2f ms in %d instrumented steps; %.2f ms wall for %d steps' % (
          tot, len(sampled), elapsed * 1e3, args.steps), file=sys.stderr)

  extractor = inference = None
  if dp.rank == 0 and not args.train_only:
    extractor = extractor_leg(torch, spectral, pool[0])
    mel0 = su.mag_to_mel_linear_spec(spectral.stft_magnitude(pool[0], 1024, 256, pad_end=False))
    inference = inference_leg(torch, AdvocSmall if args.model == 'small' else Advoc, Modes, su, mel0)
    inference['joint_sc09'] = joint_leg(torch)
  dp.barrier()

  cpu = None
  if dp.rank == 0 and args.gpus == 1 and not args.no_cpu_baseline:
    cpu = cpu_baseline(model_small=(args.model == 'small'))

  if dp.rank == 0:
    losses = model.losses()
    out = {
        'metric': 'mel-frames/sec (AdVoc G+D train step)',
        'value': value,
        'unit': 'mel-frames/s',
        'n_gpus': args.gpus,
        'steps': args.steps,
        'warmup': args.warmup,
        'ms_per_step': elapsed * 1e3 / args.steps,
        'higher_is_better': True,
        'scaling': 'weak',
        'vs_baseline': None,
        'dtype': 'f32',
        'dtype_note': 'fp32 tensors and fp32 accumulation everywhere; the large conv contractions run on the '
                      'bf16 matrix cores with each fp32 operand split exactly into three bf16 terms (six of the '
                      'nine partial products): error vs float64 2.5e-7, below the fp32 MFMA path (4e-7)',
        'data': 'synthetic',
        'config': {
            'workload': 'AdVoc-%s train_loop (1 D update + 1 G update on fresh batches), LJSpeech '
                        'geometry 22.05 kHz nfft 1024 hop 256, %d clips x 256 frames per GPU, HIP '
                        'STFT/mel/pinv extractor in the loop' % (args.model, B),
            'global_batch': B * dp.world_size,
            'frames_per_clip': CLIP_FRAMES,
            'parallelism': 'dp%d' % dp.world_size,
            'frames_counted_per_step': 'global_batch*256 (the step consumes 2 batches; 1 is counted)',
        },
        'per_gpu_value': value / dp.world_size,
        'losses': losses,
        'roofline': roofline,
        'extractor': extractor,
        'inference': inference,
        'cpu_baseline': cpu,
    }
    print(json.dumps(out))


if __name__ == '__main__':
  main()
