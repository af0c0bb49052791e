#!/usr/bin/env python
"""Headline benchmark: AdVoc G+D train step throughput in mel-frames/s (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment re-launches itself under
torch.distributed.run (one rank per GPU, rendezvous on 127.0.0.1).

Workload (config.workload): BASELINE.json configs[2] -- AdVoc (full: ngf = ndf = 64, 8 encoders), LJSpeech
geometry (22.05 kHz, nfft 1024 / hop 256, 256-frame clips), batch 64 per GPU, the HIP STFT / mel / pinv
extractor in the loop; N > 1 is configs[3] (64 clips per GPU, RCCL all-reduce of the D and G gradient arenas,
weak scaling).  One "step" = one reference train_loop (models/advoc/advoc_model.py:285-289): a discriminator
update on one batch and a generator update on the NEXT batch, each batch going waveform -> |STFT| -> mel ->
pseudo-inverse on the GPU first (advoc/loader.py:116-128 + models/advoc/train_evaluate.py:55-56).  Inputs are
synthetic waveforms (uniform noise + 3 sinusoids, seeded) already resident in HBM; weights are N(0, 0.02)
random init; dropout masks come from the on-device Philox stream.

value = (global batch x 256 frames x steps) / wall time of K uniform, uninstrumented steps: the conservative
accounting (the step consumes TWO batches; only one is counted).

Extra objects on the JSON line:
  roofline      the conv-stack kernel instance with the largest total time, timed per launch with HIP events on
                the launch stream over --prof-steps extra steps that follow the timed region in the same process
                (per-launch events serialise the weight-gradient side stream, so they are kept out of `value`):
                algorithmic flops / measured time against the roof of the pipe the kernel runs on: dense fp32
                MFMA (157.3 TFLOP/s), or, for the split-bf16 kernels, dense bf16 MFMA / 6 partial products
                (416.7 TFLOP/s algorithmic).  `traffic` = L2-miss bytes per launch of that kernel from the
                committed rocprofv3 counter passes of this same command (the newest profiles/rNN_traffic.json; FETCH_SIZE
                doubled per the gfx950 correction + WRITE_SIZE), null when not recorded for this workload.  The operand-image
                passes (amax_kernel + pair_image_kernel) that precede an image-based GEMM are launched and timed on their
                own during the instrumented steps (`operand_images(...)` in `kernels`); the small weight-image kernels
                stay inside the GEMM's call.
                `kernels` lists every instance with its share, so the HBM-bound ones can be read too.
  extractor     the HBM-bound leg: waveform -> |STFT| -> mel -> pseudo-inverse, timed per launch with HIP
                events; algorithmic bytes per clip from SURVEY.md §8d against 8 TB/s.
  inference     vocoded clips/s: mel -> pseudo-inverse -> generator forward on 256-frame chunks
                (scripts/spectrogram_advoc.py:80-94 semantics, batched), plus phase reconstruction legs and
                `joint_sc09`: z -> MelspecGAN -> AdVoc -> waveform (BASELINE configs[4] on one GPU).
  small         BASELINE configs[1] (AdVoc-small, 32 clips) train step, a short secondary run.
  loader        WAV directory -> decode_extract_and_batch -> batches: mel-frames/s of the real input pipeline.
  cpu_baseline  the torch-CPU restatement of the reference graph (oracle/, "port") timed on this box's host
                cores on a bounded sample (rank 0, N = 1 only): 8 / 16 / 32 / 64 / all threads, 3 timed iterations
                each, headline = the best setting, the sweep listed.
  dist          (N > 1) what the collectives ran on: backend, world size, every rank's device, and the measured time of
                one all-reduce of the generator's gradient arena -- evidence that RCCL saw N ranks.
"""
import argparse
import ctypes
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
  sys.path.insert(0, ROOT)

FP32_MFMA_PEAK_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md, "Peak FP32 (matrix)"
# Kernels of the split paths run every fp32 product as several 16-bit MFMA products with fp32 accumulation: their roof in
# ALGORITHMIC fp32 flops is the dense bf16 / f16 peak divided by the products per fp32 product --
#   register-split kernels (x6.h: x = x0 + x1 + x2 in bf16, six of nine partial products)          2500 / 6
#   operand-image kernels (igemm_h3.hip: x 2^s = h0 + h1 in fp16, three of four partial products)  2500 / 3
BF16_MFMA_PEAK_TFLOPS = 2500.0  # ibid., "BF16/F16 ~2.5 PF dense"
X6_PEAK_TFLOPS = BF16_MFMA_PEAK_TFLOPS / 6.0
H3_PEAK_TFLOPS = BF16_MFMA_PEAK_TFLOPS / 3.0
HBM_PEAK_GBS = 8000.0           # ibid., HBM3E peak BW
CLIP_FRAMES = 256
CLIP_SAMPLES = (CLIP_FRAMES - 1) * 256 + 1024   # 66304
def _latest_traffic_json():
  """profiles/rNN_traffic.json of the highest round (tools/run_gpu_prof_r04.sh writes it next to the other summaries)."""
  import glob
  found = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r[0-9][0-9]_traffic.json')))
  return found[-1] if found else os.path.join(ROOT, 'profiles', 'r03_traffic.json')


TRAFFIC_JSON = _latest_traffic_json()


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


def mfma_pipe(name):
  """(peak TFLOP/s in algorithmic fp32 flops, description) of the matrix pipe a kernel instance runs on."""
  if '_h3_kernel' in name or '_h3_256_kernel' in name or '_h3_flat_kernel' in name or '_h3_256_flat_kernel' in name:
    return H3_PEAK_TFLOPS, 'f16 MFMA, 3 partial products per fp32 product (2500 / 3 TFLOP/s algorithmic)'
  if name.endswith(', true>') and (name.startswith('gather_gemm_kernel<') or name.startswith('wgrad_mfma_kernel<')):
    return X6_PEAK_TFLOPS, 'bf16 MFMA, 6 partial products per fp32 product (2500 / 6 TFLOP/s algorithmic)'
  return FP32_MFMA_PEAK_TFLOPS, 'fp32 MFMA'


def is_split_bf16(name):
  return mfma_pipe(name)[0] != FP32_MFMA_PEAK_TFLOPS


def cpu_baseline(model_small, threads, batch, iters=3, give_up_s=None):
  """Reference-equivalent CPU restatement (oracle/advoc_torch.py + oracle/spectral_np.py), one train_loop = D update + G
  update, at `threads` intra-op threads: one untimed warm-up iteration (thread pool, allocator), then `iters` timed ones.
  give_up_s: when the warm-up iteration alone takes longer than this, the setting is reported from that single iteration
  (an oversubscribed all-core run of this small batch was measured 100 x slower than 16 threads: 388 s for 3 iterations)."""
  import numpy as np
  import torch
  from oracle import advoc_torch as A
  from oracle import spectral_np as S
  torch.set_num_threads(threads)
  B = batch
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
  what = 'AdVoc-%s at batch %d, STFT/mel in numpy, convs in torch-CPU fp32' % ('small' if model_small else 'full', B)
  t0 = time.perf_counter()
  tr.train_loop(make_batch(), make_batch(), masks, masks)
  t_warm = time.perf_counter() - t0
  if give_up_s is not None and t_warm > give_up_s:
    return dict(value=B * CLIP_FRAMES / t_warm, unit='mel-frames/s', cores=threads, kind='port', seconds=t_warm, iterations=1,
                best_iteration_value=B * CLIP_FRAMES / t_warm,
                sample='ONE train_loop iteration (the first, no warm-up: it alone took %.1f s, more than %.1f s = 3 x the best '
                       'setting\'s iteration; not repeated) of %s' % (t_warm, give_up_s, what))
  times = []
  for _ in range(iters):
    t0 = time.perf_counter()
    tr.train_loop(make_batch(), make_batch(), masks, masks)
    times.append(time.perf_counter() - t0)
  t_total = sum(times)
  return dict(value=B * CLIP_FRAMES * iters / t_total, unit='mel-frames/s', cores=threads, kind='port', seconds=t_total,
              iterations=iters, best_iteration_value=B * CLIP_FRAMES / min(times),
              sample='%d timed train_loop iterations (1 D + 1 G update each; 1 warm-up before) of %s' % (iters, what))


def cpu_baseline_legs(model_small, threads):
  """BASELINE.md section 3, the legs beside the headline sweep (same restatement, `threads` intra-op threads, synthetic
  clips of the GPU legs' shape): (a) the train_loop at the reference's default batch 8 (advoc_model.py:18) for the benched
  model, (b) the OTHER model at batch 8, (c) the extractor alone -- waveform -> |STFT| -> mel -> pseudo-inverse
  (advoc/spectral.py:60-83, spectral_util.py:29-43) in numpy, (d) generator-only inference on 256-frame chunks
  (scripts/spectrogram_advoc.py:80-94: mel -> pinv -> G).  Each: one warm-up, then a few timed repetitions (~20 s of CPU
  work in total)."""
  import numpy as np
  import torch
  from oracle import advoc_torch as A
  from oracle import spectral_np as S
  legs = {}
  if not model_small:      # (the headline is batch 8 since r6; the r1-r5 headline setting kept as a leg)
    r = cpu_baseline(model_small, threads, 4, iters=2)
    legs['train_batch4'] = dict(value=r['value'], unit=r['unit'], seconds=r['seconds'], iterations=r['iterations'], cores=threads,
                                sample=r['sample'] + ' (rounds 1-5 quoted this setting as the headline)')
  r = cpu_baseline(not model_small, threads, 8, iters=2 if not model_small else 1)
  legs['other_model_batch8'] = dict(value=r['value'], unit=r['unit'], seconds=r['seconds'], iterations=r['iterations'],
                                    cores=threads, sample=r['sample'])
  torch.set_num_threads(threads)
  W = S.create_mel_filterbank(22050, 1024, fmin=125, fmax=7600, n_mels=80).astype(np.float32)
  Wi = S.create_inverse_mel_filterbank(22050, 1024, fmin=125, fmax=7600, n_mels=80).astype(np.float32)
  rng = np.random.default_rng(0)
  B = 8
  wav = rng.uniform(-0.5, 0.5, size=(B, CLIP_SAMPLES, 1, 1)).astype(np.float32)

  def extract():
    mag = np.abs(S.stft_tf(wav, 1024, 256, pad_end=False)).astype(np.float32)
    mel = S.mag_to_mel_linear_spec(mag, W)
    return mag, mel, S.mel_linear_to_mag_spec(mel, Wi)
  extract()
  reps, t0 = 5, time.perf_counter()
  for _ in range(reps):
    mag, mel, inv = extract()
  dt = (time.perf_counter() - t0) / reps
  legs['extractor'] = dict(value=B / dt, unit='clips/s', us_per_clip=dt / B * 1e6, algorithmic_GBps=1397760.0 * B / dt / 1e9,
                           cores='numpy (its own BLAS threads for the two projections)', seconds=dt * reps,
                           sample='%d timed passes over %d clips of %d samples: waveform -> |STFT| (numpy rfft) -> mel -> '
                                  'pseudo-inverse; 1 397 760 algorithmic bytes per clip (SURVEY.md section 8d)' % (reps, B, CLIP_SAMPLES))
  cfg = A.Config(small=model_small)
  P = A.init_params(cfg, seed=0)
  masks = A.make_dropout_masks(cfg, B, seed=1)
  x = torch.from_numpy(inv)
  with torch.no_grad():
    A.build_generator(P, x, cfg, masks)
    reps, t0 = 3, time.perf_counter()
    for _ in range(reps):
      A.build_generator(P, x, cfg, masks)
    dt = (time.perf_counter() - t0) / reps
  legs['generator_inference'] = dict(value=B / dt, unit='256-frame clips/s', cores=threads, seconds=dt * reps,
                                     sample='%d timed generator forward passes (AdVoc-%s, torch-CPU fp32) over %d chunks of 256 '
                                            'frames, magnitudes out (no phase reconstruction)' % (reps, 'small' if model_small else 'full', B))
  return legs


def cpu_baseline_sweep(model_small):
  """The reference's CPU path beside the GPU number: the port timed at 8 / 16 / 32 / 64 / all host threads (8 is the
  reference's own extract_parallel_calls, train_evaluate.py:41), >= 3 timed iterations each; the headline is the BEST
  setting (an all-core run of a batch-4 conv stack is oversubscribed), the whole sweep is listed.  A setting whose first
  iteration is already 3 x slower than the best setting's iterations is reported from that one iteration."""
  ncores = os.cpu_count() or 8
  settings = sorted(set(t for t in (8, 16, 32, 64, ncores) if 0 < t <= ncores))
  batch = 8          # (r6: the reference's default train_batch_size, advoc_model.py:18, as BASELINE.md section 3 names it; r1-r5: 4 for the full model)
  sweep, best_iter, skipped = [], None, []
  for t in settings:
    if sweep and sweep[-1]['value'] < 0.6 * max(r['value'] for r in sweep):
      # throughput is already collapsing with more threads (oversubscription): the next, larger setting took 124 s for
      # ONE iteration on a 256-thread host -- it is listed as skipped instead of run
      skipped.append(dict(cores=t, skipped='%d threads already ran at %.2f x the best setting' % (
          sweep[-1]['cores'], sweep[-1]['value'] / max(r['value'] for r in sweep))))
      continue
    r = cpu_baseline(model_small, t, batch, iters=3, give_up_s=None if best_iter is None else 3.0 * best_iter)
    sweep.append(r)
    it = r['seconds'] / r['iterations']
    best_iter = it if best_iter is None else min(best_iter, it)
  best = max([r for r in sweep if r['iterations'] >= 3] or sweep, key=lambda r: r['value'])
  out = dict(best)
  out['host_cpus'] = ncores
  # the other legs BASELINE.md section 3 lists, each a bounded sample at the best thread count of the sweep
  try:
    out['legs'] = cpu_baseline_legs(model_small, best['cores'])
  except Exception as e:     # never take the headline line down
    out['legs'] = dict(error=repr(e))
  out['sweep'] = [dict(cores=r['cores'], value=r['value'], seconds=r['seconds'], iterations=r['iterations']) for r in sweep] + skipped
  out['note'] = 'headline = best of the thread sweep; a reported baseline, not the optimisation target'
  return out


def event_timed(torch, call, launches, warm=5):
  """Average ms per call over a TIMED REGION of `launches` back-to-back calls: ONE HIP-event pair around the region (the
  gaps between the launches are inside it, so this is an upper bound of the average kernel duration).  An event pair
  per launch -- what this function did until r4 -- puts a marker packet in front of and behind every kernel and reads
  3-10 % high on 30-120 us kernels (tools/micro/event_overhead.py: 108-124 us per launch against 104-107 us for the
  same 30 launches under one pair)."""
  # warm-up by TIME, not by count: the first launches of a leg run before the clocks have come up (a cold first repetition of
  # the 110 us STFT launch reads 120-127 us, the next ones 105-108); at least `warm` launches and 25 ms of them
  t0 = time.perf_counter()
  done = 0
  while done < warm or time.perf_counter() - t0 < 0.025:
    for _ in range(warm):
      call()
    torch.cuda.synchronize()
    done += warm
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(launches):
    call()
  e1.record()
  torch.cuda.synchronize()
  return e0.elapsed_time(e1) / launches


def extractor_leg(torch, spectral, su, wav, launches=30):
  """The feature extractor through the C ABI with preallocated outputs, GB/s of algorithmic traffic
  (SURVEY.md §8d).  `stft`: waveform -> |STFT| alone (790 528 B per clip); `triple`: waveform -> (|X|, mel,
  pseudo-inverse) = what a training batch needs (1 397 760 B per clip), fused when the library has the fused
  entry point.  Measured at 512 clips per launch (whole-file extraction in the loader) and at the training
  step's feed (one launch for the two batches of a train_loop)."""
  from advoc_amd import _lib
  lib = _lib.load()
  win = spectral._device_window(1024, 256)
  tw = spectral._device_twiddle(1024)

  def clips_of(n):
    return wav[:, :, 0, 0].repeat((n + wav.shape[0] - 1) // wav.shape[0], 1)[:n].contiguous()

  # (r6) Every timed region ROTATES over enough (input, output) buffer sets that their sum is beyond three Infinity Caches
  # (256 MiB): a hot loop over ONE set of the train feed's size -- 34 MB in, 145 MB out -- is served by that cache and read
  # 64 us where the same launch takes 82 us inside the train step.  `sets` is reported per leg.
  def n_sets(set_bytes):
    return max(1, -(-3 * 256 * 1024 * 1024 // int(set_bytes)))

  def rotating(calls):
    state = {'i': 0}

    def call():
      calls[state['i']]()
      state['i'] = (state['i'] + 1) % len(calls)
    return call

  def run_stft(clips):
    nbytes = clips * (CLIP_SAMPLES * 4 + CLIP_FRAMES * 513 * 4)
    sets = n_sets(nbytes)
    calls = []
    for _ in range(sets):
      x = clips_of(clips).clone()
      out = torch.empty(clips, CLIP_FRAMES, 513, dtype=torch.float32, device=x.device)
      calls.append(lambda x=x, out=out: _lib.check(lib.advoc_stft_mag_f32(
          _lib.ptr(x), clips, x.shape[1], _lib.ptr(win), _lib.ptr(tw), 1024, 256, CLIP_FRAMES, _lib.ptr(out), _lib.stream()), 'stft'))
    return event_timed(torch, rotating(calls), launches), nbytes, sets

  def run_triple(clips, fused):
    # waveform -> (|X|, mel, pinv(mel)) through the C ABI with preallocated outputs, like run_stft (the Python wrapper's
    # per-call allocations are part of the train step's time, not of the kernels'): the ONE launch of
    # SpectralUtil.extract_training_triple (csrc/extract.hip), or the two launches it replaced (stft + mel_pinv)
    nbytes = clips * (CLIP_SAMPLES * 4 + CLIP_FRAMES * (80 + 2 * 513) * 4)
    sets = n_sets(nbytes)
    runs, wp, inv_t = su._const('packed')
    tab, unscale = su._const('pairs')
    calls = []
    for _ in range(sets):
      x = clips_of(clips).clone()
      mag = torch.empty(clips, CLIP_FRAMES, 513, dtype=torch.float32, device=x.device)
      mel = torch.empty(clips, CLIP_FRAMES, 80, dtype=torch.float32, device=x.device)
      inv = torch.empty(clips, CLIP_FRAMES, 513, dtype=torch.float32, device=x.device)

      def call_fused(x=x, mag=mag, mel=mel, inv=inv):
        _lib.check(lib.advoc_stft_mel_pinv_f32(_lib.ptr(x), clips, x.shape[1], _lib.ptr(win), _lib.ptr(tw), 1024, 256,
                                               CLIP_FRAMES, _lib.ptr(wp), _lib.ptr(runs), int(wp.numel()), 513, 80,
                                               _lib.ptr(tab), _lib.ptr(unscale), _lib.ptr(mag), _lib.ptr(mel), _lib.ptr(inv),
                                               _lib.stream()), 'stft_mel_pinv')

      def call_two(x=x, mag=mag, mel=mel, inv=inv):
        _lib.check(lib.advoc_stft_mag_f32(_lib.ptr(x), clips, x.shape[1], _lib.ptr(win), _lib.ptr(tw), 1024, 256,
                                          CLIP_FRAMES, _lib.ptr(mag), _lib.stream()), 'stft')
        _lib.check(lib.advoc_mel_pinv_f32(_lib.ptr(mag), _lib.ptr(wp), _lib.ptr(runs), _lib.ptr(inv_t), _lib.ptr(mel),
                                          _lib.ptr(inv), clips * CLIP_FRAMES, 513, 80, int(wp.numel()), _lib.stream()),
                   'mel_pinv')
      calls.append(call_fused if fused else call_two)
    return event_timed(torch, rotating(calls), launches), nbytes, sets

  nb = 2 * wav.shape[0]
  ms_l, bytes_l, sets_l = run_stft(512)
  ms_b, bytes_b, sets_b = run_stft(nb)
  gbs = bytes_l / (ms_l * 1e-3) / 1e9
  out = dict(kernel='stft1024_hop256_kernel', bound='hbm', achieved=gbs, peak=HBM_PEAK_GBS, unit='GB/s',
             frac=gbs / HBM_PEAK_GBS, clips_per_launch=512, bytes_per_launch=bytes_l, avg_launch_ms=ms_l, buffer_sets_rotated=sets_l,
             frames_per_s=512 * CLIP_FRAMES / (ms_l * 1e-3),
             at_train_feed=dict(clips_per_launch=nb, avg_launch_ms=ms_b, achieved=bytes_b / (ms_b * 1e-3) / 1e9, buffer_sets_rotated=sets_b,
                                frac=bytes_b / (ms_b * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                frames_per_s=nb * CLIP_FRAMES / (ms_b * 1e-3)))
  fused_max = getattr(su, 'FUSED_MAX_FRAMES', 0)
  res = {}
  for tag, clips in (('bulk', 512), ('feed', nb)):
    ms_f, by, sets = run_triple(clips, True)
    ms_2, _, _ = run_triple(clips, False)
    uses_fused = clips * CLIP_FRAMES <= fused_max            # what SpectralUtil.extract_training_triple launches at this size
    ms = ms_f if uses_fused else ms_2
    res[tag] = dict(clips_per_launch=clips, avg_ms=ms, achieved=by / (ms * 1e-3) / 1e9, frac=by / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                    buffer_sets_rotated=sets,
                    path='one launch (stft_mel_pinv_kernel)' if uses_fused else 'two launches (stft1024_hop256_kernel + mel_pinv_kernel)',
                    one_launch_ms=ms_f, two_launches_ms=ms_2)
  out['triple'] = dict(what='waveform -> |X|, mel, pinv(mel) (1 397 760 B per clip) as SpectralUtil.extract_training_triple '
                            'launches it: ONE launch (csrc/extract.hip) up to %d frames per call, two above' % fused_max,
                       **res['bulk'])
  out['triple']['at_train_feed'] = res['feed']
  return out


def inference_leg(torch, model_cls, Modes, su, mel, iters=50, warm=5):
  """clips/s of mel -> magnitude through the generator (INFER mode, dropout active as in the
  reference), batch = the training batch.  Wall clock over `iters` batches after `warm` warm-ups, with
  the HIP-event time of the same region beside it."""
  from advoc_amd import spectral
  m = model_cls(Modes.INFER)
  B = mel.shape[0]
  m.build(batch_size=B, seed=0)
  for _ in range(warm):
    m.build_generator(su.mel_linear_to_mag_spec(mel))
  torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  t0 = time.perf_counter()
  e0.record()
  for _ in range(iters):
    m.build_generator(su.mel_linear_to_mag_spec(mel))
  e1.record()
  torch.cuda.synchronize()
  dt = time.perf_counter() - t0
  dt_ev = e0.elapsed_time(e1) * 1e-3
  out = dict(value=B * iters / dt, unit='vocoded 256-frame clips/s', batch=B, iters=iters, warmup=warm,
             ms_per_batch=dt / iters * 1e3, ms_per_batch_hip_events=dt_ev / iters * 1e3,
             note='value: mel -> pinv projection -> generator forward (magnitudes), wall clock')
  # waveform synthesis on top, all clips of the batch at once
  mag = m.build_generator(su.mel_linear_to_mag_spec(mel))[..., 0].abs().contiguous()
  u = torch.rand(mag.shape, device=mag.device)
  spectral.griffin_lim_batch(mag, 1024, 256, 2, u)
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  for _ in range(3):
    spectral.griffin_lim_batch(mag, 1024, 256, 60, u)
  torch.cuda.synchronize()
  dt_gl = (time.perf_counter() - t0) / 3
  out.update(with_gl60_clips_per_s=B / (dt / iters + dt_gl), gl60_ms_per_batch=dt_gl * 1e3)
  if hasattr(spectral, 'lws_batch'):
    spectral.lws_batch(mag, 1024, 256)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
      spectral.lws_batch(mag, 1024, 256)
    torch.cuda.synchronize()
    dt_lws = (time.perf_counter() - t0) / 3
    out.update(with_lws_clips_per_s=B / (dt / iters + dt_lws), lws_ms_per_batch=dt_lws * 1e3,
               lws_note='LWS (the reference default phase_estimation): restated from the published algorithm, '
                        'parity unpinned (lws 1.2 is third-party and absent)')
    # LWS's time-ordered pass runs one CU per clip (3 072 dependent Jacobi steps per clip): 64 clips fill a quarter of the
    # chip, so a vocoding service hands it the output of FOUR generator batches at once (advoc_amd.infer.vocode_batch does
    # the same: every chunk through the generator in chunk_batch pieces, phase reconstruction over all samples at once)
    group = 4
    mags = torch.cat([mag] * group, dim=0).contiguous()
    spectral.lws_batch(mags, 1024, 256)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(2):
      spectral.lws_batch(mags, 1024, 256)
    torch.cuda.synchronize()
    dt_lws_g = (time.perf_counter() - t0) / 2
    out.update(with_lws_grouped_clips_per_s=group * B / (group * dt / iters + dt_lws_g), lws_group_batches=group,
               lws_ms_per_group=dt_lws_g * 1e3)
    # the reference's vocode path ends in a WAVEFORM through its default phase estimator (scripts/spectrogram_advoc.py:95,
    # advoc/spectral.py:314-326): that is the headline; the magnitude-only rate stays beside it
    out['magnitudes_only_clips_per_s'] = out['value']
    out['value'] = out['with_lws_grouped_clips_per_s']
    out['unit'] = 'vocoded 256-frame clips/s (mel -> pinv -> generator -> LWS waveform)'
    out['note'] = ('value: mel -> pinv projection -> generator forward (batches of %d) -> LWS phase reconstruction (the reference '
                   'default) over %d batches per call, waveform out; with_lws_clips_per_s: LWS per single batch; '
                   'magnitudes_only_clips_per_s stops at the generator output (wall clock over the same batches)' % (B, group))
  return out


def joint_leg(torch, n=64, iters=5):
  """BASELINE configs[4] on one GPU: z -> MelspecGAN generator -> mel [64 x 80] -> AdVoc (full model at
  subseq_len 64, its (1,2)-stride layers) -> LWS (the reference default; Griffin-Lim 60 beside it) -> 16 kHz waveform; random
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

  def run(phase):
    mel = G(z, denorm=True)
    return vocode_batch(voc, mel, phase_estimation=phase, chunk_batch=2 * n)[1]
  res = {}
  for phase in ('lws', 'gl60'):
    for _ in range(2):
      run(phase)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
      wav = run(phase)
    torch.cuda.synchronize()
    res[phase] = (time.perf_counter() - t0) / iters
  dt = res['lws']
  return dict(value=n / dt, unit='generated 64-frame clips/s (z -> 16 kHz waveform, LWS: the reference default)', batch=n,
              ms_per_batch=dt * 1e3, samples_per_clip=int(wav.shape[1]),
              with_gl60_clips_per_s=n / res['gl60'], gl60_ms_per_batch=res['gl60'] * 1e3,
              note='MelspecGAN G + AdVoc-full(subseq_len 64) + phase reconstruction, random weights; LWS parity unpinned')


def loader_leg(torch, seconds=6.0, n_files=48, batch=64):
  """The real input pipeline: a directory of synthetic PCM16 WAVs with LJSpeech-like lengths (1.1-10.1 s at
  22.05 kHz) -> advoc_amd.loader.decode_extract_and_batch (decode threads, device STFT, slicing, shuffle buffer,
  batching) -> [B,256,513,1] batches; mel-frames/s delivered, beside what one GPU's train step consumes."""
  import shutil
  import tempfile
  import numpy as np
  from advoc_amd import audioio, loader
  d = tempfile.mkdtemp(prefix='advoc_bench_wav_')
  try:
    rng = np.random.default_rng(0)
    fps = []
    for i in range(n_files):
      n = int(22050 * rng.uniform(1.1, 10.1))
      x = (rng.uniform(-0.5, 0.5, size=(n, 1, 1)) * 0.5).astype(np.float32)
      fp = os.path.join(d, '%04d.wav' % i)
      audioio.save_as_wav(fp, 22050, x)
      fps.append(fp)
    out = {}
    for workers in (4, 16):
      pipe = loader.BatchPipeline(
          fps, batch, 256, audio_fs=22050, audio_mono=True, audio_normalize=True, decode_fastwav=True,
          decode_parallel_calls=workers, extract_type='magspec', extract_nfft=1024, extract_nhop=256,
          repeat=True, shuffle=True, shuffle_buffer_size=512,
          slice_first_only=False, slice_randomize_offset=True, slice_overlap_ratio=0.25, slice_pad_end=False,
          prefetch_size=8, seed=0)
      for _ in range(3):
        pipe.next()
      torch.cuda.synchronize()
      t0 = time.perf_counter()
      nb = 0
      while time.perf_counter() - t0 < seconds / 2:
        pipe.next()
        nb += 1
      torch.cuda.synchronize()
      dt = time.perf_counter() - t0
      pipe.close()
      out['workers_%d' % workers] = dict(frames_per_s=nb * batch * CLIP_FRAMES / dt, batches=nb, seconds=dt)
    best = max(v['frames_per_s'] for v in out.values())
    return dict(value=best, unit='mel-frames/s delivered (WAV files -> [B,256,513,1] batches)', batch=batch,
                files=n_files, detail=out,
                note='synthetic PCM16 WAVs, LJSpeech-like lengths; decode on host threads, STFT / slicing / '
                     'shuffle / batching on the device (advoc/loader.py:66-214 semantics)')
  finally:
    shutil.rmtree(d, ignore_errors=True)


def kernel_source_sha16():
  """sha256[:16] of the HIP sources the library is built from (csrc/*.hip, *.h, include/*.h): what a counter pass was taken
  ON.  (The GPU box has no .git: a commit id is not available where bench.py runs; the sources are.)"""
  import glob
  import hashlib
  h = hashlib.sha256()
  files = sorted(glob.glob(os.path.join(ROOT, 'advoc_amd', 'csrc', '*.hip')) + glob.glob(os.path.join(ROOT, 'advoc_amd', 'csrc', '*.h'))
                 + glob.glob(os.path.join(ROOT, 'include', '*.h')))
  for f in files:
    h.update(os.path.basename(f).encode())
    h.update(open(f, 'rb').read())
  return h.hexdigest()[:16]


def recorded_traffic(kernel, model, batch):
  """(L2-miss bytes per launch, provenance) from the committed counter passes (tools/pmc_summary.py) -- None with the
  reason when they were taken on another workload or on OTHER KERNEL SOURCES than the ones this run was built from (r5: the
  file used to be trusted whatever its age)."""
  rel = os.path.relpath(TRAFFIC_JSON, ROOT)
  if not os.path.exists(TRAFFIC_JSON):
    return None, rel + ': not found'
  rec = json.load(open(TRAFFIC_JSON))
  meta = rec.get('_workload', {})
  how = ' (rocprofv3 FETCH_SIZE x 2 + WRITE_SIZE passes of `%s`, committed; not re-measured by this run)' % meta.get('command', '?')
  if meta.get('model') != model or meta.get('batch') != batch:
    return None, rel + ': taken on another workload' + how
  now, then = kernel_source_sha16(), meta.get('kernel_source_sha16')
  if then != now:
    return None, '%s: STALE -- taken on kernel sources %s, this build is %s; traffic withheld%s' % (rel, then, now, how)
  row = rec.get(kernel)
  return (row['traffic_bytes'] if row else None), '%s: kernel sources %s = this build%s' % (rel, then, how)


def roofline_from(prof, model, batch, ms_per_step, prof_steps, verbose):
  rows = prof.rows()
  tot = sum(v['ms'] for v in rows.values())
  kernels = []
  for k, v in sorted(rows.items(), key=lambda kv: -kv[1]['ms']):
    mfma = 'mfma' in k or 'gather_gemm' in k or 'wgrad_h3' in k or 'patch_gemm' in k
    entry = dict(kernel=k, launches_per_step=v['launches'] / prof_steps, share_of_conv_stack=v['ms'] / tot,
                 avg_launch_ms=v['ms'] / v['launches'])
    # the roof that bounds a kernel is the one that needs MORE time for its algorithmic work: flops at the peak of the
    # matrix pipe it runs on, or bytes at the HBM peak (the 1- and 2-channel edge layers and the 16-column first stage of
    # the two-stage path run MFMA instructions but move ~8 flop per byte)
    t_hbm = v['bytes'] / (HBM_PEAK_GBS * 1e9)
    t_mfma = v['flops'] / (mfma_pipe(k)[0] * 1e12) if (mfma and v['flops'] > 0) else 0.0
    if t_mfma >= t_hbm:
      peak = mfma_pipe(k)[0]
      entry.update(bound='mfma', achieved=v['flops'] / (v['ms'] * 1e-3) / 1e12, peak=round(peak, 1), unit='TFLOP/s')
    else:
      entry.update(bound='hbm', achieved=v['bytes'] / (v['ms'] * 1e-3) / 1e9, peak=HBM_PEAK_GBS, unit='GB/s')
    entry['frac'] = entry['achieved'] / entry['peak']
    kernels.append(entry)
  name, r = max(rows.items(), key=lambda kv: kv[1]['ms'])
  top = kernels[0]
  traffic, traffic_source = recorded_traffic(name, model, batch)
  roofline = dict(bound=top['bound'], kernel=name, achieved=top['achieved'], peak=top['peak'], unit=top['unit'],
                  frac=top['frac'], traffic=traffic, traffic_source=traffic_source,
                  algorithmic_bytes_per_launch=r['bytes'] / r['launches'],
                  traffic_population='(r6) `traffic` and `algorithmic_bytes_per_launch` average the SAME launches: the C calls '
                                     'that run `%s` (the `_flat` instances of the short-row layers are rows of their own in `kernels`)' % name,
                  launches=r['launches'], avg_launch_ms=r['ms'] / r['launches'],
                  share_of_step=r['ms'] / prof_steps / ms_per_step, instrumented_steps=prof_steps,
                  measured='HIP events per launch on %d serial steps right after the timed region' % prof_steps,
                  launch_covers=('wgrad_h3_256_kernel + the wgrad_reduce_kernel that sums its K slices in order (one C call; '
                                 'rocprofv3 lists the two separately)') if name == 'wgrad_h3_256_kernel' else name,
                  kernels=kernels[:24])
  cp = getattr(prof, 'clock_probe', None)
  if cp and cp[1] > 0:
    # (r6) the shader clock wgrad_h3_256_kernel ran at in these steps (its first workgroup's s_memtime / s_memrealtime,
    # advoc_clock_probe_read): the dense peak the fraction is priced against assumes 2.4 GHz
    ghz = 0.1 * cp[0] / cp[1]
    roofline.update(clock_ghz=ghz, clock_launches_probed=cp[2], peak_clock_ghz=2.4,
                    frac_of_peak_at_this_clock=(top['frac'] * 2.4 / ghz) if (name == 'wgrad_h3_256_kernel' and ghz > 0) else None,
                    clock_note='clock_ghz: wgrad_h3_256_kernel, workgroup 0, every launch of the instrumented steps; '
                               'frac_of_peak_at_this_clock = frac x 2.4 / clock_ghz = the schedule\'s share of the matrix '
                               'pipe, the rest of the gap to 1.0 is the sustained clock under this kernel\'s power draw')
  if top['bound'] == 'mfma':
    roofline.update(pipe=mfma_pipe(name)[1],
                    vs_fp32_mfma_peak=top['achieved'] / FP32_MFMA_PEAK_TFLOPS,
                    flops_per_launch=r['flops'] / r['launches'])
  if verbose:
    for k, v in sorted(rows.items(), key=lambda kv: -kv[1]['ms']):
      tf = v['flops'] / max(v['ms'], 1e-9) / 1e9
      print('  %-52s launches %5d  %9.2f ms (%5.1f%%)  %7.2f TFLOP/s  %7.1f GB/s alg' % (
          k, v['launches'], v['ms'], 100 * v['ms'] / tot, tf, v['bytes'] / max(v['ms'], 1e-9) / 1e6),
          file=sys.stderr)
    print('  conv-stack launches total %.2f ms in %d instrumented steps; %.2f ms per timed step' % (
        tot, prof_steps, ms_per_step), file=sys.stderr)
  return roofline


def train_leg(torch, model_name, B, steps, warmup, dp, dev, prof_steps):
  """Returns (elapsed seconds of `steps` uninstrumented train_loops [max over ranks], model, su, pool, profiler)."""
  from advoc_amd import conv, spectral
  from advoc_amd.model import Advoc, AdvocSmall, Modes
  from advoc_amd.spectral_util import SpectralUtil
  model = (AdvocSmall if model_name == 'small' else Advoc)(Modes.TRAIN)
  model.train_batch_size = B
  model.build(batch_size=B, seed=0)
  dp.attach(model)
  dp.broadcast_parameters(model)
  su = SpectralUtil(n_mels=model.n_mels, fs=model.audio_fs)
  # a train_loop consumes two batches (D update, then G update): both are extracted by ONE set of launches over
  # 2B clips (a launch over B clips alone leaves most of the 256 CUs idle), then handed out half by half
  pool = [synth_waveforms(2 * B, 1234 + 17 * dp.rank + i, dev) for i in range(2)]
  state = {'i': 0, 'pending': None}

  def feed():
    if state['pending'] is not None:
      out, state['pending'] = state['pending'], None
      return out
    wav = pool[state['i'] % len(pool)]
    state['i'] += 1
    if state.get('timed') is not None:       # (instrumented steps: the extractor's launch(es) between two HIP events)
      e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
      e0.record()
      mag, mel, inv = su.extract_training_triple(wav)
      e1.record()
      state['timed'].append((e0, e1))
    else:
      mag, mel, inv = su.extract_training_triple(wav)
    state['pending'] = (inv[B:], mag[B:], wav[B:], mel[B:])
    return inv[:B], mag[:B], wav[:B], mel[:B]
  model(feed)

  for _ in range(warmup):
    model.train_loop()
  torch.cuda.synchronize()
  dp.barrier()
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  for _ in range(steps):
    model.train_loop()
  torch.cuda.synchronize()
  dp.barrier()
  torch.cuda.synchronize()
  elapsed = dp.max_over_ranks(time.perf_counter() - t0)

  prof = None
  if prof_steps > 0:
    prof = conv.LaunchProfiler()
    conv.Layer.profiler = prof
    state['timed'] = []
    from advoc_amd import _lib as _l
    clk = (ctypes.c_uint64 * 3)()
    _l.check(_l.load().advoc_clock_probe_read(clk, 1), 'advoc_clock_probe_read')       # reset: count the instrumented steps only
    for _ in range(prof_steps):
      model.train_loop()
    torch.cuda.synchronize()
    conv.Layer.profiler = None
    _l.check(_l.load().advoc_clock_probe_read(clk, 0), 'advoc_clock_probe_read')
    prof.clock_probe = [int(v) for v in clk]
    prof.extract_ms = [e0.elapsed_time(e1) for e0, e1 in state['timed']]
    prof.extract_clips = 2 * B
    state['timed'] = None
  return elapsed, model, su, pool, prof


def dist_report(torch, dp, model):
  """N > 1: backend, world size, per-rank devices (gathered) and the time of one all-reduce of the G gradient arena
  (bucketed as in training), max over ranks.  Called by every rank; rank 0 gets the dict."""
  if not dp.enabled:
    return None
  import torch.distributed as dist
  props = torch.cuda.get_device_properties(dp.local_rank)
  import hashlib
  st = model._built
  model._flush_d_adam()
  torch.cuda.synchronize()
  digest = hashlib.sha1(st['g_param'].cpu().numpy().tobytes() + st['d_param'].cpu().numpy().tobytes()).hexdigest()[:16]
  mine = dict(rank=dp.rank, local_rank=dp.local_rank, device='cuda:%d' % dp.local_rank, name=props.name,
              gcn_arch=getattr(props, 'gcnArchName', ''), host=socket.gethostname(), pid=os.getpid(),
              # every rank's parameters after the timed steps: data parallel is right iff these are all equal
              param_sha1_16=digest)
  ranks = [None] * dp.world_size
  dist.all_gather_object(ranks, mine)
  flat = model._built['g_grad']
  scratch = torch.zeros_like(flat)
  times = []
  for i in range(4):
    torch.cuda.synchronize()
    dp.barrier()
    t0 = time.perf_counter()
    dp.allreduce_(scratch)
    torch.cuda.synchronize()
    times.append(time.perf_counter() - t0)
  ms = dp.max_over_ranks(min(times[1:])) * 1e3
  nbytes = flat.numel() * 4
  return dict(backend=dp.backend, world_size=dp.world_size, ranks=ranks, bucket_bytes=dp.bucket_elems * 4,
              reserve_cus=int(os.environ.get('ADVOC_RESERVE_CUS', '0') or 0),
              reserve_cus_source=('ADVOC_RESERVE_CUS' if 'ADVOC_RESERVE_CUS' in os.environ and dp.reserve_source == 'user'
                                  else 'ADVOC_DP_RESERVE_CUS (bench.py default 8 for N > 1)' if dp.reserve_source == 'dp' else 'none'),
              params_equal_across_ranks=len(set(r['param_sha1_16'] for r in ranks)) == 1,
              g_arena_bytes=nbytes, g_arena_allreduce_ms=ms,
              g_arena_allreduce_busbw_gbs=nbytes * 2 * (dp.world_size - 1) / dp.world_size / (ms * 1e-3) / 1e9,
              rccl='torch.distributed backend "nccl" = RCCL on ROCm' if dp.backend == 'nccl' else 'host-staged (wiring check)')


def dist_step_check(torch, dp, dev):
  """(r6) Correctness evidence beside the throughput of an N-GPU run: ONE AdVoc-small train_loop (32 frames, 2 clips per rank)
  on a global batch sharded over the ranks -- real RCCL, the asynchronous bucketed all-reduce, the deferred discriminator
  update -- against the same loop on the whole global batch in ONE process on rank 0 (tests/test_hip_parallel.py's contract:
  the arena of a rank holds the SUM over ranks, Adam applies 1/N).  Called by every rank; rank 0 gets the dict."""
  from advoc_amd.model import AdvocSmall, Modes
  T, per = 32, 2
  GB = per * dp.world_size

  def make(batch):
    m = AdvocSmall(Modes.TRAIN)
    m.subseq_len, m.train_batch_size = T, batch
    m.build(batch_size=batch, seed=13)
    return m
  g = torch.Generator().manual_seed(21)
  batches = []
  for _ in range(2):
    target = torch.rand(GB, T, 513, 1, generator=g) * 2
    batches.append((target * (0.5 + torch.rand(GB, T, 513, 1, generator=g)) - 0.1, target))

  def run(model, lo, hi):
    it = iter(batches)

    def feed():
      x, t = next(it)
      return x[lo:hi].to(dev), t[lo:hi].to(dev)
    model(feed)
    model.train_loop()
    torch.cuda.synchronize()
    st = model._built
    return {k: v.detach().double().cpu() for name in ('d_G', 'g_G') for k, v in st[name].items()}
  m = make(per)
  dp.attach(m)
  dp.broadcast_parameters(m)
  mine = run(m, dp.rank * per, (dp.rank + 1) * per)
  out = None
  if dp.rank == 0:
    from advoc_amd.parallel import DataParallel as _DP
    ref_model = _DP().attach(make(GB))           # (a DataParallel that is not enabled: the plain single-process step)
    ref = run(ref_model, 0, GB)
    worst = {'discriminator': 0.0, 'generator': 0.0}
    for k, v in ref.items():
      e = float((mine[k] / dp.world_size - v).norm() / v.norm().clamp_min(1e-30))
      net = 'discriminator' if k.startswith('discriminator') else 'generator'
      worst[net] = max(worst[net], e)
    # bars of tests/test_hip_parallel.py: D gradients at the initial weights 2e-5; G gradients behind D's first Adam step 5e-3
    ok = worst['discriminator'] < 2e-5 and worst['generator'] < 5e-3
    out = dict(step_equals_single_gpu=bool(ok), worst_rel_l2_first_step_gradients=worst,
               what='AdVoc-small, 32 frames, %d clips sharded over %d ranks vs one process on rank 0' % (GB, dp.world_size))
  dp.barrier()
  return out


def free_port():
  s = socket.socket()
  s.bind(('127.0.0.1', 0))
  port = s.getsockname()[1]
  s.close()
  return port


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--gpus', type=int, default=1)
  ap.add_argument('--steps', type=int, default=40)
  ap.add_argument('--warmup', type=int, default=3)
  ap.add_argument('--model', choices=['small', 'regular'], default='regular')
  ap.add_argument('--batch', type=int, default=0, help='clips per GPU (default 64 regular / 32 small)')
  ap.add_argument('--prof-steps', type=int, default=4,
                  help='instrumented (per-launch HIP events, serial) steps after the timed region; 0: no roofline')
  ap.add_argument('--no-cpu-baseline', action='store_true')
  ap.add_argument('--train-only', action='store_true',
                  help='skip the extractor / inference / small / loader legs that run after the timed region '
                       '(profiling runs: keeps the per-kernel statistics to the train step)')
  ap.add_argument('--no-launch-timing', action='store_true', help='same as --prof-steps 0')
  args = ap.parse_args()

  if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
    # plain `python bench.py --gpus N`: become the launcher (one rank per GPU, RCCL over xGMI)
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus),
           '--master-addr', '127.0.0.1', '--master-port', str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    raise SystemExit(subprocess.call(cmd, env=env))

  import torch
  from advoc_amd import spectral
  from advoc_amd.model import Advoc, AdvocSmall, Modes
  from advoc_amd.parallel import DataParallel

  # N > 1: the persistent launches leave 8 CUs (one per XCD) to RCCL's kernels unless the caller chose (r5; recorded in
  # `dist.reserve_cus`; NOTEBOOK.md section 5).  An explicit ADVOC_RESERVE_CUS / ADVOC_DP_RESERVE_CUS is never overridden.
  if int(os.environ.get('WORLD_SIZE', '1')) > 1 and 'ADVOC_DP_RESERVE_CUS' not in os.environ \
      and 'ADVOC_RESERVE_CUS' not in os.environ:
    os.environ['ADVOC_DP_RESERVE_CUS'] = '8'
  dp = DataParallel().init_from_env()
  if dp.world_size != args.gpus:
    raise SystemExit('--gpus %d but WORLD_SIZE=%d' % (args.gpus, dp.world_size))
  dev = torch.device('cuda', dp.local_rank)
  torch.cuda.set_device(dev)

  B = args.batch or (32 if args.model == 'small' else 64)
  prof_steps = 0 if args.no_launch_timing else args.prof_steps
  elapsed, model, su, pool, prof = train_leg(torch, args.model, B, args.steps, args.warmup, dp, dev, prof_steps)
  frames = B * dp.world_size * CLIP_FRAMES * args.steps
  value = frames / elapsed
  ms_per_step = elapsed * 1e3 / args.steps

  roofline = None
  if prof is not None and dp.rank == 0:
    roofline = roofline_from(prof, args.model, B, ms_per_step, prof_steps, bool(os.environ.get('ADVOC_BENCH_VERBOSE')))
  losses = model.losses() if dp.rank == 0 else None
  delayed = ({'exact_refits': int(model.image_refits()), 'values_out_of_window': int(model.image_saturations())}
             if dp.rank == 0 else None)
  dist_info = dist_report(torch, dp, model)
  if dp.enabled:
    try:
      chk = dist_step_check(torch, dp, dev)
    except Exception as e:     # never take the headline line down; the failure is in the line
      chk = dict(step_equals_single_gpu=False, error=repr(e))
    if dist_info is not None and chk is not None:
      dist_info.update(chk)

  extractor = inference = small = loader_res = None
  if dp.rank == 0 and not args.train_only:
    extractor = extractor_leg(torch, spectral, su, pool[0][:B])
    if prof is not None and getattr(prof, 'extract_ms', None):
      # (r6) the same launch INSIDE the train step (instrumented steps: one HIP-event pair around the extractor call of every
      # train_loop): operands cold, the previous step's Adam pass just behind it -- the figure the roofline claim is about
      ms_in = sum(prof.extract_ms) / len(prof.extract_ms)
      by = prof.extract_clips * (CLIP_SAMPLES * 4 + CLIP_FRAMES * (80 + 2 * 513) * 4)
      extractor['triple']['in_step'] = dict(us=ms_in * 1e3, clips_per_launch=prof.extract_clips, calls=len(prof.extract_ms),
                                            achieved=by / (ms_in * 1e-3) / 1e9, frac=by / (ms_in * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                            what='SpectralUtil.extract_training_triple as the train step calls it (its launch and '
                                                 'its output allocation), HIP events on the launch stream')
      extractor['triple']['in_step_us'] = ms_in * 1e3
    mel0 = su.mag_to_mel_linear_spec(spectral.stft_magnitude(pool[0][:B], 1024, 256, pad_end=False))
    inference = inference_leg(torch, AdvocSmall if args.model == 'small' else Advoc, Modes, su, mel0)
    inference['joint_sc09'] = joint_leg(torch)
    try:
      loader_res = loader_leg(torch)
    except Exception as e:   # the loader leg must never take the headline line down
      loader_res = dict(error=repr(e))
  dp.barrier()
  if not args.train_only and args.model == 'regular' and dp.world_size == 1:
    # secondary: BASELINE configs[1] (AdVoc-small, 32 clips), short
    del model
    torch.cuda.empty_cache()
    from advoc_amd.parallel import DataParallel as _DP
    el_s, m_s, _, _, _ = train_leg(torch, 'small', 32, 20, 3, _DP(), dev, 0)
    small = dict(workload='AdVoc-small train_loop, 32 clips x 256 frames (BASELINE configs[1])', steps=20,
                 ms_per_step=el_s * 1e3 / 20, value=32 * CLIP_FRAMES * 20 / el_s, unit='mel-frames/s')
    del m_s

  cpu = None
  if dp.rank == 0 and args.gpus == 1 and not args.no_cpu_baseline:
    cpu = cpu_baseline_sweep(args.model == 'small')

  if dp.rank == 0:
    out = {
        'metric': 'mel-frames/sec (AdVoc G+D train step)',
        'value': value,
        'unit': 'mel-frames/s',
        'n_gpus': args.gpus,
        'steps': args.steps,
        'warmup': args.warmup,
        'ms_per_step': ms_per_step,
        'higher_is_better': True,
        'scaling': 'weak',
        'vs_baseline': None,
        'dtype': 'f32',
        'dtype_note': 'fp32 tensors and fp32 accumulation everywhere; the large conv contractions run on the 16-bit '
                      'matrix cores with each fp32 operand written as a sum of 16-bit terms -- forward, backward-data '
                      'and weight gradients: two fp16 terms of the operand scaled by one power of two per tensor, three '
                      'of the four partial products (representation and dropped term <= 2^-22 relative); shapes outside '
                      'those kernels: three bf16 terms, six of nine partial products, or the fp32 MFMA (1-2 channel edge '
                      'layers) -- measured against float64 at or below the error of the fp32 MFMA chain on the same '
                      'layers (tests/test_hip_conv.py, tools/micro/h3_numerics.py)',
        'data': 'synthetic',
        'config': {
            'workload': 'AdVoc-%s train_loop (1 D update + 1 G update on fresh batches), LJSpeech '
                        'geometry 22.05 kHz nfft 1024 hop 256, %d clips x 256 frames per GPU, HIP '
                        'STFT/mel/pinv extractor in the loop (BASELINE configs[%d])' % (
                            'full' if args.model == 'regular' else 'small', B,
                            (3 if dp.world_size > 1 else 2) if args.model == 'regular' else 1),
            'global_batch': B * dp.world_size,
            'frames_per_clip': CLIP_FRAMES,
            'parallelism': 'dp%d' % dp.world_size,
            'frames_counted_per_step': 'global_batch*256 (the step consumes 2 batches; 1 is counted)',
        },
        'per_gpu_value': value / dp.world_size,
        'target_frames_per_s_per_gpu': 50000,
        'losses': losses,
        'delayed_scaling': {**delayed,
                            'note': 'operand images re-built on the device with the exact scale since the model was built '
                                    '(warm-up + timed + instrumented steps); nothing clamped is ever consumed'},
        'roofline': roofline,
        'extractor': extractor,
        'inference': inference,
        'small': small,
        'loader': loader_res,
        'cpu_baseline': cpu,
        'dist': dist_info,
    }
    print(json.dumps(out))


if __name__ == '__main__':
  main()
