"""Thin Python binding of the convolution-stack C ABI (advoc_conv_* in include/advoc_hip.h).

One ``Layer`` object = one conv / transposed-conv of the reference graph
(/root/reference/models/advoc/advoc_model.py:25-69) with its fused input
activation, skip concat, width trim, bias and dropout.  All tensors are
float32 NHWC torch tensors resident in HBM; this module only fills
``advoc_conv_layer`` structs and calls into libadvoc_hip.so -- no arithmetic
happens in Python/torch.
"""
import ctypes
import os

import torch

from advoc_amd import _lib

CONV = 0
DECONV = 1
ACT_NONE = 0
ACT_LRELU = 1
ACT_RELU = 2


def same_pad(n, k, s):
  """TF 'SAME' padding: (before, after) for input size n (tf.layers.conv2d, advoc_model.py:46-51)."""
  out = -(-n // s)
  total = max((out - 1) * s + k - n, 0)
  return total // 2, total - total // 2


def _t4(t, w=None):
  """advoc_tensor4 view of a contiguous NHWC tensor; w = logical width (<= physical)."""
  if t is None:
    return _lib.Tensor4(None, 0, 0, 0, 0, 0)
  _lib.require_device(t)
  if t.dtype != torch.float32 or t.dim() != 4:
    raise _lib.AdvocHipError('expected float32 NHWC tensor, got {} {}'.format(t.dtype, tuple(t.shape)))
  n, h, wp, c = t.shape
  w = wp if w is None else w
  if not 0 < w <= wp:
    raise _lib.AdvocHipError('logical width {} outside physical width {}'.format(w, wp))
  return _lib.Tensor4(t.data_ptr(), n, h, w, c, wp)


class LaunchProfiler(object):
  """Optional per-launch timing with HIP events recorded on the launch stream (bench.py).
  Nothing is synchronised until `rows()` is called."""

  def __init__(self):
    self.records = []

  def timed(self, name, flops, nbytes, fn):
    e0 = torch.cuda.Event(enable_timing=True)
    e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    fn()
    e1.record()
    self.records.append((name, flops, nbytes, e0, e1))

  def rows(self):
    """{kernel name: dict(launches, ms, flops, bytes)}; synchronises."""
    torch.cuda.synchronize()
    out = {}
    for name, flops, nbytes, e0, e1 in self.records:
      r = out.setdefault(name, dict(launches=0, ms=0.0, flops=0.0, bytes=0.0))
      r['launches'] += 1
      r['ms'] += e0.elapsed_time(e1)
      r['flops'] += flops
      r['bytes'] += nbytes
    return out


class Layer(object):
  """A bound conv layer: holds the ctypes struct and keeps every tensor it points at alive."""

  profiler = None   # set to a LaunchProfiler to time every launch
  _workspaces = {}
  _wgrad_scratches = {}
  # True: the caller guarantees that between a forward() and the backward_weight() that follows it the layer's inputs
  # are unchanged, and that backward_data(dy) / backward_weight(dy) of one step see the same dy contents (the train
  # step of advoc_amd.model does): the operand images forward / backward_data leave behind are then read again by
  # backward_weight instead of being rebuilt.  False (default, stand-alone use): every call builds what it reads.
  reuse_images = False
  # True: consecutive calls on this layer see data of the same scale (the train step): after the first (exact, two-pass)
  # image of a buffer, later images are built in ONE pass with the scale taken from the previous image's largest
  # magnitude (ADVOC_IMG_*_DELAYED, 2^6 of head room; a tensor that leaves it, or shrinks by more than 2^6, is re-imaged
  # exactly on the device in the same call; csrc/image.hip).  False (default): always the exact two-pass form.
  delayed_scale = False
  # True (Layer.emit_images = False turns it off): a layer whose consumers were registered with add_image_consumer writes THEIR
  # operand images from its own forward epilogue (advoc_conv_layer.y_img, csrc/image_emit.h) once the consumer's header
  # holds a previous magnitude -- the consumer's image pass (a read and a write of the whole tensor) disappears
  emit_images = True
  # Layer.emit_dx = True turns it on: a backward_data call given `grad_consumer` -- the layer below, whose output gradient this
  # call's dx0 is -- writes THAT layer's output-gradient image (and its bias column sums) from its own epilogue
  # (advoc_conv_layer.dx_img): the image pass of the layer below disappears.  Off by default: measured on the one producer
  # that has it (discriminator layer_5 -> layer_4, the largest image pass of the step) the passes lose 0.31 ms per step and
  # the producer, an issue-bound kernel, gains 0.18 ms; the step does not move (NOTEBOOK.md section 7, profiles/r04_i_*)
  emit_dx = False
  # (r5) ADVOC_DX_BOUNDED=0 turns it off: where the backward-data call of a layer runs on a patch kernel and the layer below
  # reads its output gradient only as an operand image, that image is written by this call's epilogue under a scale derived
  # from an a-priori bound of |dx| (max|dy| max|w| taps K: nothing can leave the fp16 range, so no history, no refit, no fp32
  # tensor to refit from) and the fp32 dx0 is NOT written: the image pass of the layer below (a read and a write of the
  # tensor) disappears at no extra store in the epilogue.  Needs reuse_images (the train step's guarantees).
  dx_bounded = os.environ.get('ADVOC_DX_BOUNDED', '1') == '1'
  # Layer.dx_accum = True: ... also from calls that ACCUMULATE into dx0 (the generator's encoder chain: a decoder's skip gradient
  # arrives first, with its largest magnitude recorded -- dx1_amax / bound_add).  Built, value-checked
  # (tests/test_hip_conv.py: enc_accum*) and OFF by default: same box, alternating, 38.56 / 38.50 / 38.66 ms without against
  # 38.48 / 38.54 / 38.73 with it -- the accumulating epilogue pays in loads and image arithmetic what the seven passes cost.
  dx_accum = False
  # (r5) ADVOC_Y_IMAGE_ONLY=0 turns it off: a layer whose output has ONE reader (add_image_consumer(..., exclusive=True)) that
  # reads it as an operand image and gates its backward-data pass on that image's signs (a patch kernel) writes the IMAGE
  # ONLY, under a scale from an a-priori bound of |y| (max|x| max|w| taps K + max|b|): the fp32 tensor -- half the bytes the
  # forward epilogue stores, twice the bytes the consumer's gating reads -- is never written.  Where the producer's kernel
  # can (advoc_conv_emits_images() == 2: the <= 2-input-channel matrix kernel, i.e. the discriminator's layer_1).
  y_image_only = os.environ.get('ADVOC_Y_IMAGE_ONLY', '1') == '1'

  @staticmethod
  def _workspace_for(device, nbytes):
    ws = Layer._workspaces.get(device)
    if ws is None or ws.numel() * 4 < nbytes:
      ws = torch.empty((nbytes + 3) // 4, dtype=torch.float32, device=device)
      Layer._workspaces[device] = ws
    return ws

  @staticmethod
  def _wgrad_scratch_for(device, nbytes):
    """Scratch for the K slices of the image weight gradient (advoc_conv_layer.wgrad_ws): one buffer per device, shared by
    all layers and grown on demand -- their backward_weight calls must therefore be ordered on ONE stream (the model runs
    every weight gradient on its side stream, or everything on one)."""
    ws = Layer._wgrad_scratches.get(device)
    if ws is None or ws.numel() * 4 < nbytes:
      ws = torch.empty((nbytes + 3) // 4, dtype=torch.float32, device=device)
      Layer._wgrad_scratches[device] = ws
    return ws

  def __init__(self, kind, x0, y, weight, bias=None, x1=None, in_w=None, out_w=None, stride=(2, 2),
               pad=(1, 1), in_act=ACT_NONE, drop_mask=None, drop_scale=0., in_scale=None,
               in_shift=None, in_mask=None, in_mask_scale=0., workspace=True, w_amax=None):
    kh, kw = int(weight.shape[0]), int(weight.shape[1])
    cin = x0.shape[3] + (x1.shape[3] if x1 is not None else 0)
    cout = y.shape[3]
    want = (kh, kw, cin, cout) if kind == CONV else (kh, kw, cout, cin)
    if tuple(weight.shape) != want:
      raise _lib.AdvocHipError('kernel shape {} != {}'.format(tuple(weight.shape), want))
    if bias is not None and tuple(bias.shape) != (cout,):
      raise _lib.AdvocHipError('bias shape {} != ({},)'.format(tuple(bias.shape), cout))
    for t in (weight, bias, drop_mask, in_scale, in_shift, in_mask, w_amax):
      if t is not None:
        _lib.require_device(t)
    if w_amax is not None and (w_amax.dtype != torch.int32 or w_amax.numel() != 1):
      raise _lib.AdvocHipError('w_amax must be one int32 (float bits of max |kernel|, advoc_segmented_amax_f32)')
    if in_mask is not None and (in_mask.dtype != torch.uint8 or tuple(in_mask.shape) != tuple(x0.shape)):
      raise _lib.AdvocHipError('input mask must be uint8 with the shape of x0')
    if in_scale is not None and (tuple(in_scale.shape) != (cin,) or tuple(in_shift.shape) != (cin,)):
      raise _lib.AdvocHipError('in_scale / in_shift must have one entry per input channel')
    if drop_mask is not None and (drop_mask.dtype != torch.uint8 or tuple(drop_mask.shape) != tuple(y.shape)):
      raise _lib.AdvocHipError('dropout mask must be uint8 with the shape of y')
    self.kind = kind
    self.tensors = (x0, x1, y, weight, bias, drop_mask, in_scale, in_shift, in_mask, w_amax)
    self.x0, self.x1, self.y, self.weight, self.bias = x0, x1, y, weight, bias
    s = _lib.ConvLayer()
    s.kind = kind
    s.kh, s.kw = kh, kw
    s.sh, s.sw = stride
    s.pad_t, s.pad_l = pad
    s.in_act = in_act
    s.x0 = _t4(x0, in_w)
    s.x1 = _t4(x1, in_w)
    s.in_scale = _lib.ptr(in_scale)
    s.in_shift = _lib.ptr(in_shift)
    s.y = _t4(y, out_w)
    s.w = _lib.ptr(weight)
    s.b = _lib.ptr(bias)
    s.w_amax = _lib.ptr(w_amax)
    s.drop_mask = _lib.ptr(drop_mask)
    s.drop_scale = float(drop_scale)
    s.in_mask = _lib.ptr(in_mask)
    s.in_mask_scale = float(in_mask_scale)
    s.workspace = None
    s.workspace_bytes = 0
    self.struct = s
    # scratch for the two-stage path of the 1-2 channel layers: one buffer per device, shared by
    # all layers (launches are stream-ordered), grown on demand
    need = max(_lib.load().advoc_conv_workspace_bytes(ctypes.byref(s), d) for d in (0, 1, 2))
    if need > 0 and workspace:
      ws = Layer._workspace_for(x0.device, need)
      s.workspace = ws.data_ptr()
      s.workspace_bytes = ws.numel() * 4
      self.tensors = self.tensors + (ws,)
    # persistent operand images (advoc_conv_layer.x_img / dy_img): only where the image-based weight gradient applies
    self._img = []
    self._x_current = False
    self._dy_current_ptr = None
    self._x_built = False          # x_img / dy_img have held an image before (their headers carry its magnitude)
    self._dy_built = False
    if workspace:
      for which, img_field, hdr_field in ((0, 'x_img', 'x_hdr'), (1, 'dy_img', 'dy_hdr')):
        nbytes = _lib.load().advoc_conv_image_bytes(ctypes.byref(s), which)
        if nbytes > 0:
          # (zeros, once: a producer that writes this image from its epilogue -- dx_img -- writes the LOGICAL pixels only; a
          # trimmed column of the tensor must read as the zero gradient it is, as it does behind an image pass)
          img = torch.zeros(nbytes // 2, dtype=torch.int16, device=x0.device)
          hdr = torch.zeros(8, dtype=torch.int32, device=x0.device)
          setattr(s, img_field, img.data_ptr())
          setattr(s, hdr_field, hdr.data_ptr())
          self._img += [img, hdr]
      self.tensors = self.tensors + tuple(self._img)
      need = _lib.load().advoc_conv_wgrad_ws_bytes(ctypes.byref(s))
      if need > 0:
        wws = Layer._wgrad_scratch_for(x0.device, need)
        s.wgrad_ws = wws.data_ptr()
        s.wgrad_ws_bytes = wws.numel() * 4
        self.tensors = self.tensors + (wws,)
    s.img_flags = 0
    # headers of dy_img per caller ROLE (set_dy_role): one layer object can see output gradients of different losses in
    # one train step (the discriminator's fake pass: D-loss gradients in the D step, G-loss gradients in the G step);
    # each role keeps its own magnitude history so that delayed scaling compares like with like
    self._dy_roles = {}
    self._dy_role = None
    self._names = {}
    self._consumers = []           # [(consumer layer, source index)] whose x_img this layer's forward can write
    self._exclusive = False        # the one consumer is the only reader of y (add_image_consumer)
    self._x_final = False          # x_img was written under an a-priori scale since the last forward: no refit check
    self._x_gates = False          # ... and the fp32 input was not written: backward_data gates on the image
    self._gates_ok = None
    self._w_l1 = 0                 # ADVOC_IMG_W_L1 when set_weight_image was given 32-word headers
    self._emits = None             # advoc_conv_emits_images(), asked once
    self._x_emitted = set()        # sources of x_img written by their producers since the last forward
    self._db_done_for = None
    self._emits_dx = None          # advoc_conv_emits_dx_image(), asked once
    self._emits_dx_acc = None      # ... as an accumulating call
    self._dx_table = None          # replica table of the column sums that ride in the dx image emission
    self._dy_emitted_for = None    # (dy pointer, db pointer or None): dy_img was written by the layer above's backward_data
    self._dy_image_only = None     # dy pointer whose fp32 tensor was never written (the image in dy_img is its only copy)
    self._bias_fusable = bool(_lib.load().advoc_conv_bias_fusable(ctypes.byref(s)))
    self._thin_bias = False
    if kind == CONV and cin <= 2 and workspace:
      # the replica table of the bias sums that ride in the thin weight-gradient kernel: owned by the layer, because
      # weight gradients run on a side stream next to other layers' users of the shared workspace (advoc_amd.model)
      table = torch.empty(_lib.WGRAD_TABLE_BYTES // 4, dtype=torch.float32, device=x0.device)
      s.wgrad_table = table.data_ptr()
      self.tensors = self.tensors + (table,)
      try:
        self._thin_bias = 'thin_wgrad_kernel' in self.kernel_name(2)
      except _lib.AdvocHipError:
        pass
    lw = s.x0.w
    grid = (y.shape[1] * s.y.w) if kind == CONV else (x0.shape[1] * lw)
    self.flops = 2.0 * x0.shape[0] * grid * kh * kw * cin * cout
    # algorithmic HBM bytes of one pass: every input element once, every output element once, weights
    self.bytes_fwd = 4.0 * (x0.shape[0] * x0.shape[1] * lw * cin + y.shape[0] * y.shape[1] * s.y.w * cout
                            + kh * kw * cin * cout)
    # backward-data through an input activation reads the pre-activation input once more (act'(x) gates dx: the
    # reference's ReluGrad / LeakyReluGrad op reads it too); forward and the weight gradient touch x, y / dy, w once
    self.bytes_dir = (self.bytes_fwd,
                      self.bytes_fwd + (4.0 * x0.shape[0] * x0.shape[1] * lw * cin if in_act != ACT_NONE else 0.0),
                      self.bytes_fwd)

  def weight_image_desc(self, direction):
    """(taps, n_total, ktot, b_kn, bytes) of the weight image the forward (0) / backward-data (1) call reads, or None when
    that direction does not run on the image kernels (advoc_conv_weight_image_desc)."""
    out = (ctypes.c_int64 * 5)()
    _lib.check(_lib.load().advoc_conv_weight_image_desc(ctypes.byref(self.struct), direction, out),
               'advoc_conv_weight_image_desc')
    return tuple(int(v) for v in out) if out[4] > 0 else None

  def set_weight_image(self, direction, img_ptr, hdr_ptr, l1=False):
    """Persistent weight image of this direction (advoc_weight_images_f32 keeps it current; None: per-call images).
    l1=True: the headers are the 32-word ones of advoc_weight_images_l1_f32 (per-tap row-L1 maxima: ADVOC_IMG_W_L1)."""
    self.struct.w_img[direction] = img_ptr
    self.struct.w_img_hdr[direction] = hdr_ptr
    self._w_l1 = 512 if l1 else 0

  def kernel_name(self, direction):
    """Kernel template instance this layer launches for direction 0 fwd / 1 bwd-data / 2 bwd-weight."""
    if direction not in self._names:
      buf = ctypes.create_string_buffer(128)
      _lib.check(_lib.load().advoc_conv_kernel_name(ctypes.byref(self.struct), direction, buf, 128),
                 'advoc_conv_kernel_name')
      self._names[direction] = buf.value.decode()
    return self._names[direction]

  def _run(self, direction, fn, extra_bytes=0.0):
    prof = Layer.profiler
    if prof is None:
      fn()
    else:
      prof.timed(self.kernel_name(direction), self.flops, self.bytes_dir[direction] + extra_bytes, fn)

  def _timed_image(self, which, dy=None):
    """With a launch profiler attached, the operand-image passes of an image-based call are launched (and timed) on
    their own, so that the GEMM's own duration is what its kernel name is charged with -- the same split rocprofv3
    shows.  Returns the img_flags bit to pass to the call that follows."""
    prof = Layer.profiler
    if prof is None or not (self.struct.x_img if which == 0 else self.struct.dy_img):
      return 0
    if 'h3' not in self.kernel_name(0 if which == 0 else 1):
      return 0
    src = [t for t in (self.x0, self.x1) if t is not None] if which == 0 else [dy]
    self.struct.img_flags = self._delayed_bits()
    one_pass = bool(self.struct.img_flags & (4 if which == 0 else 8)) and (self._x_built if which == 0 else self._dy_built)
    # exact scaling reads the source twice (magnitude pass, image pass); delayed scaling once; 4 B written either way
    nbytes = sum((8.0 if one_pass else 12.0) * t.numel() for t in src)
    prof.timed('operand_images(amax_kernel + pair_image_kernel)', 0.0, nbytes, lambda: _lib.check(
        _lib.load().advoc_conv_make_image(ctypes.byref(self.struct), which, _lib.ptr(dy), _lib.stream()),
        'advoc_conv_make_image'))
    if which == 0:
      self._x_built = True
    else:
      self._dy_built = True
    return 1 if which == 0 else 2

  def set_dy_role(self, role):
    """Selects the magnitude history (header) the next backward_data / backward_weight calls use for the image of dy.
    The image buffer itself is shared; only the 32-byte header (and its 'an image was built before' flag) is per role."""
    if not self.struct.dy_img or role == self._dy_role:
      return
    if self._dy_role is None:                       # the header allocated with the buffer becomes the first role's
      self._dy_roles[role] = [self._img[-1], self._dy_built] if role not in self._dy_roles else self._dy_roles[role]
    else:
      self._dy_roles[self._dy_role][1] = self._dy_built
      if role not in self._dy_roles:
        hdr = torch.zeros(8, dtype=torch.int32, device=self.x0.device)
        self._dy_roles[role] = [hdr, False]
        self.tensors = self.tensors + (hdr,)
    self._dy_role = role
    hdr, built = self._dy_roles[role]
    self.struct.dy_hdr = hdr.data_ptr()
    self._dy_built = built
    self._dy_current_ptr = None

  def image_headers(self):
    """Every operand-image header of this layer (torch int32[8] each)."""
    out = list(self._img[1::2])
    for hdr, _ in self._dy_roles.values():
      if all(hdr.data_ptr() != h.data_ptr() for h in out):
        out.append(hdr)
    return out

  def add_image_consumer(self, consumer, source, exclusive=False):
    """`consumer` (a Layer) reads this layer's output y as its input `source` (0: x0, 1: x1): from the second step on this
    layer's forward writes the consumer's operand image itself.  exclusive=True: the caller guarantees that NOTHING else
    reads y (no other layer, no summary, no batch norm): where the kernels allow, y then exists as that image only
    (Layer.y_image_only)."""
    want = consumer.x0 if source == 0 else consumer.x1
    if want is None or want.data_ptr() != self.y.data_ptr() or tuple(want.shape) != tuple(self.y.shape):
      raise _lib.AdvocHipError('the consumer\'s input {} is not this layer\'s output'.format(source))
    if len(self._consumers) >= 2:
      raise _lib.AdvocHipError('at most two image consumers per layer')
    self._consumers.append((consumer, source))
    self._exclusive = bool(exclusive) and len(self._consumers) == 1

  def _emit_targets(self):
    """[(slot, consumer, source)] this forward call writes images for."""
    if not (Layer.emit_images and self.delayed_scale and self._consumers):
      return []
    if self._emits is None:
      self._emits = int(_lib.load().advoc_conv_emits_images(ctypes.byref(self.struct)))
    if not self._emits:
      return []
    if self._image_only_consumer() is not None:
      return [(0,) + self._consumers[0]]        # under the a-priori scale: no magnitude history needed
    out = []
    for k, (c, src) in enumerate(self._consumers):
      cs = c.struct
      if not (c.delayed_scale and c._x_built and cs.x_img and 'h3' in c.kernel_name(0)):
        continue
      if cs.in_scale or cs.in_mask:          # batch-norm affine / input dropout: not known when the producer runs
        continue
      out.append((k, c, src))
    return out

  def tracks_dx1_amax(self):
    """True when this layer's backward-data kernel raises dx1_amax (the image kernels: patch and per-tap)."""
    return self.x1 is not None and '_h3_kernel' in self.kernel_name(1)

  def _image_only_consumer(self):
    """The one consumer this layer's output exists for as an image only (Layer.y_image_only), or None."""
    if not (Layer.y_image_only and self._exclusive and self._emits == 2 and self.reuse_images and len(self._consumers) == 1):
      return None
    c, src = self._consumers[0]
    cs = c.struct
    if src != 0 or c.x1 is not None or not (c.reuse_images and cs.x_img) or cs.in_scale or cs.in_mask:
      return None
    if not c.kernel_name(1).startswith('patch_gemm_h3_kernel') or 'h3' not in c.kernel_name(0):
      return None
    #  ... and its WEIGHT GRADIENT must read x_img too: the fp32 kernels it would otherwise fall back to read the
    # tensor this layer never writes
    if 'h3' not in c.kernel_name(2):
      return None
    if c._gates_ok is None:      # the consumer's backward-data launch must be one that can gate on the image: asked once
      c._gates_ok = bool(_lib.load().advoc_conv_gates_on_image(ctypes.byref(cs)))
    return c if c._gates_ok else None

  def _delayed_bits(self):
    if not self.delayed_scale:
      return 0
    return (4 if self._x_built else 0) | (8 if self._dy_built else 0)

  def forward(self):
    nsrc = 2 if self.x1 is not None else 1
    if len(self._x_emitted) == nsrc and self.struct.x_img:
      # ADVOC_IMG_X_CURRENT | ADVOC_IMG_X_EMITTED (refit check instead of an image pass) or | ADVOC_IMG_X_BOUNDED (final)
      flags = 1 | (128 if self._x_final else 16)
    else:
      flags = self._timed_image(0)
      self._x_final = self._x_gates = False
    self._x_emitted = set()
    targets = self._emit_targets()
    only = self._image_only_consumer() if targets else None
    for k in (0, 1):
      self.struct.y_img[k].img = None
      self.struct.y_img[k].mode = 0
    for k, c, src in targets:
      off = 0 if src == 0 else (4 * c.x0.numel() + 255) // 256 * 256
      self.struct.y_img[k].img = c.struct.x_img + off
      self.struct.y_img[k].hdr = c.struct.x_hdr
      self.struct.y_img[k].act = c.struct.in_act
      self.struct.y_img[k].mode = 3 if only is not None else 0      # ADVOC_Y_BOUNDED | ADVOC_Y_IMAGE_ONLY
    self.struct.img_flags = flags | self._delayed_bits() | self._w_l1
    try:
      # (the consumers' operand images this launch writes: 2 fp16 terms = 4 bytes per element and consumer)
      self._run(0, lambda: _lib.check(
          _lib.load().advoc_conv_forward(ctypes.byref(self.struct), _lib.stream()), 'advoc_conv_forward'),
                extra_bytes=4.0 * self.y.numel() * len(targets))
    finally:
      self.struct.img_flags = 0
      for k in (0, 1):
        self.struct.y_img[k].img = None
        self.struct.y_img[k].mode = 0
    for k, c, src in targets:
      c._x_emitted.add(src)
      c._x_final = c._x_gates = only is not None
    if self.struct.x_img and 'h3' in self.kernel_name(0):
      self._x_built = True
    # the image-based forward kernel has just left the input image in x_img
    self._x_current = bool(self.struct.x_img) and 'h3' in self.kernel_name(0)
    return self.y

  def _dx_target(self, dx0, dx1, accum0, accum1, consumer, consumer_db, bound_add=None):
    """(the layer below (`consumer`) whose output-gradient image this backward_data call can write, kind) or (None, 0);
    kind 2: the thin matrix kernel under the one-pass scale (fp32 dx0 written too), 3: a patch kernel under the a-priori
    scale, image only."""
    if consumer is None or dx0 is None or accum1 or (accum0 and (bound_add is None or not Layer.dx_accum)):
      return None, 0
    cs = consumer.struct
    if not (cs.dy_img and cs.dy_hdr):
      return None, 0
    if cs.drop_mask or 'h3' not in consumer.kernel_name(1) or 'h3' not in consumer.kernel_name(2):
      return None, 0
    if consumer_db is not None and not consumer._bias_fusable:
      return None, 0
    if accum0:
      # an accumulating call (dx0 already holds a skip gradient whose largest magnitude is in bound_add): asked separately
      if self._emits_dx_acc is None:
        self.struct.dx_img.bound_add = bound_add.data_ptr()
        try:
          self._emits_dx_acc = int(_lib.load().advoc_conv_emits_dx_image(ctypes.byref(self.struct)))
        finally:
          self.struct.dx_img.bound_add = None
      kind = self._emits_dx_acc if self._emits_dx_acc == 3 else 0
    else:
      if self._emits_dx is None:
        self._emits_dx = int(_lib.load().advoc_conv_emits_dx_image(ctypes.byref(self.struct)))
      kind = self._emits_dx
    # 3 / 4: a patch kernel / the thin matrix kernel under the a-priori scale (a launch that reports 2 runs as 4 when max |w|
    # is on the device); 2: the thin kernel under the one-pass scale (r4, off by default)
    if kind in (3, 4) or (kind == 2 and self.struct.w_amax):
      bounded_ok = Layer.dx_bounded and self.reuse_images and consumer.reuse_images
      if bounded_ok:
        kind = 3
      elif kind != 2:
        return None, 0
    if kind == 2:
      if not (Layer.emit_dx and dx1 is None and self.delayed_scale and consumer.delayed_scale and consumer._dy_built):
        return None, 0
    elif kind != 3:
      return None, 0
    if self.x0.data_ptr() != consumer.y.data_ptr() or tuple(dx0.shape) != tuple(consumer.y.shape):
      raise _lib.AdvocHipError('this layer\'s input is not the consumer layer\'s output')
    return consumer, kind

  def backward_data(self, dy, dx0=None, dx1=None, accum0=False, accum1=False, db=None, db_accumulate=True,
                    grad_consumer=None, consumer_db=None, consumer_db_accumulate=True, dx1_amax=None, bound_add=None):
    """dx0 / dx1 <- gradient w.r.t. the inputs.  db (optional): the bias gradient buffer of this layer -- where the call
    builds the image of dy (image kernels, advoc_conv_bias_fusable) the per-channel sums are taken in the same pass and
    the backward_weight call that follows with the same dy skips its bias kernel.
    grad_consumer (optional): the Layer whose output y this layer reads as x0, i.e. whose output GRADIENT dx0 is; where the
    kernels allow (advoc_conv_emits_dx_image) this call writes that layer's output-gradient image itself, and with
    consumer_db its bias gradient (per-channel sums of dx0): its backward_data / backward_weight calls that follow with
    dx0 (and consumer_db) then build neither.
    dx1_amax (optional, one int32 on the device, zeroed by the caller): receives the float bits of the largest |value| this
    call writes to dx1 (image kernels only -- check tracks_dx1_amax()); bound_add: such a word for the tensor dx0 ACCUMULATES
    into (accum0): with it an accumulating call can write the layer below's image too (r5: the encoder chain)."""
    _lib.require_device(dy)
    if tuple(dy.shape) != tuple(self.y.shape):
      raise _lib.AdvocHipError('dy shape {} != y shape {}'.format(tuple(dy.shape), tuple(self.y.shape)))
    for d, x in ((dx0, self.x0), (dx1, self.x1)):
      if d is not None:
        _lib.require_device(d)
        if x is None or tuple(d.shape) != tuple(x.shape):
          raise _lib.AdvocHipError('dx must have the shape of the matching input')
    self._db_done_for = None
    # the layer above wrote dy_img (and possibly this layer's bias gradient) from its backward_data epilogue
    emitted, self._dy_emitted_for = self._dy_emitted_for, None
    if emitted is not None and not (emitted[0] == dy.data_ptr() and 'h3' in self.kernel_name(1)):
      emitted = None
    db_by_producer = emitted is not None and db is not None and emitted[1] == db.data_ptr()
    self._dy_image_only = dy.data_ptr() if (emitted is not None and emitted[2]) else None
    fuse_db = db is not None and self._bias_fusable and 'h3' in self.kernel_name(1) and emitted is None
    if fuse_db:
      _lib.require_device(db)
      if not db_accumulate:
        db.zero_()
      self.struct.db_fused = _lib.ptr(db)
    target, dx_kind = self._dx_target(dx0, dx1, accum0, accum1, grad_consumer, consumer_db, bound_add)
    self.struct.dx1_amax = dx1_amax.data_ptr() if (dx1_amax is not None and dx1 is not None and self.tracks_dx1_amax()) else None
    if target is not None:
      self.struct.dx_img.mode = 3 if dx_kind == 3 else 0        # ADVOC_DX_BOUNDED | ADVOC_DX_IMAGE_ONLY
      self.struct.dx_img.bound_add = bound_add.data_ptr() if (accum0 and bound_add is not None) else None
      if consumer_db is not None:
        _lib.require_device(consumer_db)
        if not consumer_db_accumulate:
          consumer_db.zero_()
        if self._dx_table is None:
          self._dx_table = torch.empty(_lib.WGRAD_TABLE_BYTES // 4, dtype=torch.float32, device=dy.device)
        self.struct.dx_img.colsum = _lib.ptr(consumer_db)
        self.struct.dx_img.table = self._dx_table.data_ptr()
      self.struct.dx_img.img = target.struct.dy_img
      self.struct.dx_img.hdr = target.struct.dy_hdr
    try:
      if emitted is not None:
        # ADVOC_IMG_DY_CURRENT | ADVOC_IMG_DY_EMITTED (refit check instead of an image pass) or | ADVOC_IMG_DY_BOUNDED (final)
        flags = 2 | (64 if emitted[2] else 32)
      else:
        flags = self._timed_image(1, dy)
      # (ADVOC_IMG_X_GATES: the fp32 input was never written, its producer left the image only)
      self.struct.img_flags = flags | self._delayed_bits() | (256 if self._x_gates else 0) | self._w_l1
      self._run(1, lambda: _lib.check(_lib.load().advoc_conv_backward_data(
          ctypes.byref(self.struct), _lib.ptr(dy), _lib.ptr(dx0), _lib.ptr(dx1), int(accum0),
          int(accum1), _lib.stream()), 'advoc_conv_backward_data'),
                extra_bytes=4.0 * dx0.numel() if target is not None else 0.0)
    finally:                     # a failed call must not leave a stale bias pointer / flags behind
      self.struct.img_flags = 0
      self.struct.db_fused = None
      self.struct.dx_img.img = None
      self.struct.dx_img.hdr = None
      self.struct.dx_img.colsum = None
      self.struct.dx_img.table = None
      self.struct.dx_img.mode = 0
      self.struct.dx_img.bound_add = None
      self.struct.dx1_amax = None
    if target is not None:
      target._dy_emitted_for = (dx0.data_ptr(), consumer_db.data_ptr() if consumer_db is not None else None, dx_kind == 3)
    if fuse_db or db_by_producer:
      self._db_done_for = (dy.data_ptr(), db.data_ptr())
    if self.struct.dy_img and 'h3' in self.kernel_name(1):
      self._dy_built = True
    self._dy_current_ptr = dy.data_ptr() if (self.struct.dy_img and 'h3' in self.kernel_name(1)) else None

  _wgrad_scratch_use = {}        # device -> stream of the last launch that used the shared K-slice scratch

  def backward_weight(self, dy, dw, db=None, accumulate=False):
    """dw (+ db) <- gradients of the kernel (and bias).  Layers that sum their K slices in order share ONE scratch buffer
    per device (advoc_conv_layer.wgrad_ws).  Calls on one stream are ordered by the stream (the model runs every weight
    gradient on its side stream, or everything on one); a call that arrives on ANOTHER stream first makes that stream wait
    for everything queued on the last user's stream (which includes the last launch that used the scratch), so two models,
    or a profiled and an unprofiled step, never have two launches writing the same partial tiles.  (r5: nothing is recorded
    per call any more -- the per-call event record put a marker packet, ~6 us of idle queue, behind every weight-gradient
    launch of the step: profiles/r05_b_step_sequence.txt.)"""
    _lib.require_device(dy)
    scratch_user = bool(self.struct.wgrad_ws)
    if scratch_user:
      cur = torch.cuda.current_stream(dy.device)
      last = Layer._wgrad_scratch_use.get(dy.device)
      if last is not None and last.cuda_stream != cur.cuda_stream:
        cur.wait_stream(last)
      Layer._wgrad_scratch_use[dy.device] = cur
    _lib.require_device(dw)
    if tuple(dw.shape) != tuple(self.weight.shape):
      raise _lib.AdvocHipError('dw shape mismatch')
    flags = 0
    if self.reuse_images:
      flags = (1 if self._x_current else 0) | (2 if self._dy_current_ptr == dy.data_ptr() else 0)
    #  operands that exist as images ONLY can never be rebuilt from their fp32 tensors (never written): a call
    # that would do so is refused instead of overwriting the one valid copy with an image of garbage
    if self._x_gates and not (flags & 1):
      raise _lib.AdvocHipError('backward_weight: the input exists as an operand image only and that image is not current '
                               '(reuse_images off, or no forward since the inputs changed)')
    pend = self._dy_emitted_for
    if ((self._dy_image_only == dy.data_ptr() or (pend is not None and pend[2] and pend[0] == dy.data_ptr()))
        and not (flags & 2)):
      raise _lib.AdvocHipError('backward_weight: this output gradient exists as an operand image only; call backward_data '
                               '(which adopts the image) first, with reuse_images on')
    dy_only = self._dy_image_only == dy.data_ptr()
    # (ADVOC_IMG_X_GATES / ADVOC_IMG_DY_BOUNDED tell the library which operands have no fp32 tensor behind them: it refuses
    # every path that would read one)
    self.struct.img_flags = flags | self._delayed_bits() | (256 if self._x_gates else 0) | (64 if dy_only else 0)
    # (the bias gradient counts as done only for the SAME dy and the SAME db buffer the backward_data call summed into)
    db_done = db is not None and self._db_done_for == (dy.data_ptr(), db.data_ptr())
    # 1-2 channel inputs (encoder_1, layer_1): the weight-gradient kernel reads every dy element exactly once and takes the
    # bias gradient on the way (advoc_conv_backward_weight's db argument)
    db_rides = db is not None and not db_done and self._thin_bias
    if db_rides:
      _lib.require_device(db)
    try:
      self._run(2, lambda: _lib.check(_lib.load().advoc_conv_backward_weight(
          ctypes.byref(self.struct), _lib.ptr(dy), _lib.ptr(dw), _lib.ptr(db) if db_rides else None, int(accumulate),
          _lib.stream()), 'advoc_conv_backward_weight'))
    finally:
      self.struct.img_flags = 0
    # one use per forward: the caller may rewrite the inputs before the next call -- EXCEPT an input that exists as the
    # image only (nothing to rewrite it from: it stays current until the next forward; gradient accumulation, profilers)
    if not self._x_gates:
      self._x_current = False
    if 'h3' in self.kernel_name(2):        # the image-based weight gradient has (re)built whatever was not current
      self._x_built = self._x_built or bool(self.struct.x_img)
      self._dy_built = self._dy_built or bool(self.struct.dy_img)
    # one use per backward_data (the next step's dy lives at the same address) -- except an image-only dy, see above
    if self._dy_image_only is None or self._dy_image_only != self._dy_current_ptr:
      self._dy_current_ptr = None
    self._db_done_for = None
    if db is not None and not db_done and not db_rides:
      if dy_only:
        raise _lib.AdvocHipError('backward_weight: the bias gradient of an image-only output gradient comes from its producer '
                                 '(consumer_db of the layer above); it cannot be summed from a tensor that was never written')
      _lib.require_device(db)
      call = lambda: _lib.check(_lib.load().advoc_conv_backward_bias(      # noqa: E731
          ctypes.byref(self.struct), _lib.ptr(dy), _lib.ptr(db), int(accumulate), _lib.stream()),
          'advoc_conv_backward_bias')
      prof = Layer.profiler
      if prof is None:
        call()
      else:
        prof.timed('bias_grad_kernel', 0.0, 4.0 * dy.numel(), call)
