"""ctypes binding of libadvoc_hip.so (the C ABI declared in include/advoc_hip.h).

There is deliberately NO fallback: if the shared library is missing or a call
returns an error code, this module raises.  The product never computes the hot
path on the CPU.
"""
import ctypes
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
# ADVOC_HIP_LIB: another build of the same library (A/B timing of kernel variants, tools/micro); never a fallback
LIB_PATH = os.environ.get('ADVOC_HIP_LIB') or os.path.join(_HERE, 'csrc', 'libadvoc_hip.so')
HEADER_PATH = os.path.join(os.path.dirname(_HERE), 'include', 'advoc_hip.h')

_p = ctypes.c_void_p
_i32 = ctypes.c_int32
_i64 = ctypes.c_int64
_f32 = ctypes.c_float


class AdvocHipError(RuntimeError):
  pass


class Tensor4(ctypes.Structure):
  """struct advoc_tensor4 (include/advoc_hip.h)."""
  _fields_ = [('p', _p), ('n', _i32), ('h', _i32), ('w', _i32), ('c', _i32), ('w_pitch', _i32)]


class ImageOut(ctypes.Structure):
  """advoc_conv_layer.y_img[k] (include/advoc_hip.h)."""
  _fields_ = [('img', _p), ('hdr', _p), ('act', _i32), ('mode', _i32)]


class DxImage(ctypes.Structure):
  """advoc_conv_layer.dx_img (include/advoc_hip.h)."""
  _fields_ = [('img', _p), ('hdr', _p), ('colsum', _p), ('table', _p), ('mode', _i32), ('reserved', _i32),
              ('bound_add', _p)]


class ConvLayer(ctypes.Structure):
  """struct advoc_conv_layer (include/advoc_hip.h)."""
  _fields_ = [
      ('kind', _i32), ('kh', _i32), ('kw', _i32), ('sh', _i32), ('sw', _i32),
      ('pad_t', _i32), ('pad_l', _i32), ('in_act', _i32),
      ('x0', Tensor4), ('x1', Tensor4),
      ('in_scale', _p), ('in_shift', _p),
      ('y', Tensor4),
      ('w', _p), ('b', _p),
      ('drop_mask', _p), ('drop_scale', _f32),
      ('in_mask', _p), ('in_mask_scale', _f32),
      ('workspace', _p), ('workspace_bytes', _i64),
      ('x_img', _p), ('x_hdr', _p), ('dy_img', _p), ('dy_hdr', _p), ('img_flags', _i32),
      ('db_fused', _p), ('w_amax', _p), ('w_img', _p * 2), ('w_img_hdr', _p * 2),
      ('wgrad_table', _p),
      ('y_img', ImageOut * 2),
      ('wgrad_ws', _p), ('wgrad_ws_bytes', _i64),
      ('dx_img', DxImage),
      ('dx1_amax', _p),
  ]

WGRAD_TABLE_BYTES = 262144      # ADVOC_WGRAD_TABLE_BYTES


# name -> (restype, argtypes).  Must list every symbol include/advoc_hip.h declares
# (tests/test_abi.py checks the two against each other and against `nm -D`).
PROTOTYPES = {
    'advoc_abi_version': (ctypes.c_int, []),
    'advoc_error_string': (ctypes.c_char_p, [ctypes.c_int]),
    'advoc_target_arch': (ctypes.c_char_p, []),
    'advoc_last_hip_error': (ctypes.c_char_p, []),
    'advoc_tuning_reload': (None, []),
    'advoc_clock_probe_read': (ctypes.c_int, [ctypes.POINTER(ctypes.c_uint64), ctypes.c_int32]),
    'advoc_stft_mag_f32': (ctypes.c_int, [_p, _i64, _i64, _p, _p, _i32, _i32, _i64, _p, _p]),
    'advoc_stft_c64': (ctypes.c_int, [_p, _i64, _i64, _p, _p, _i32, _i32, _i64, _p, _p]),
    'advoc_stft_twiddle_host': (ctypes.c_int, [_p, _i32]),
    'advoc_istft_f32': (ctypes.c_int, [_p, _i64, _i64, _p, _p, _i32, _i32, _p, _p, _p]),
    'advoc_istft_project_f32': (ctypes.c_int, [_p, _p, _i64, _i64, _p, _p, _i32, _i32, _p, _p, _p]),
    'advoc_phase_project_c64': (ctypes.c_int, [_p, _p, _i64, _p]),
    'advoc_cabs_f32': (ctypes.c_int, [_p, _p, _i64, _p]),
    'advoc_polar_c64': (ctypes.c_int, [_p, _p, _p, _i64, _p]),
    'advoc_lws_mean_mag_f32': (ctypes.c_int, [_p, _i64, _i64, _p, _p]),
    'advoc_lws_causal_c64': (ctypes.c_int, [_p, _p, _p, _i64, _i64, _i32, _i32, _p, _i32, _i32, _i32, _p, _i32, _i32, _f32, _f32, _i32, _p]),
    'advoc_lws_batch_c64': (ctypes.c_int, [_p, _p, _p, _p, _i64, _i64, _i32, _i32, _p, _i32, _i32, _f32, _p]),
    'advoc_lws_batch_sweeps_c64': (ctypes.c_int, [_p, _p, _p, _p, _i64, _i64, _i32, _i32, _p, _i32, _i32, _p, _i32, _p, _p]),
    'advoc_matmul_nt_f32': (ctypes.c_int, [_p, _p, _p, _i64, _i32, _i32, _p]),
    'advoc_mel_pinv_f32': (ctypes.c_int, [_p, _p, _p, _p, _p, _p, _i64, _i32, _i32, _i32, _p]),
    'advoc_stft_mel_pinv_f32': (ctypes.c_int, [_p, _i64, _i64, _p, _p, _i32, _i32, _i64, _p, _p, _i32, _i32, _i32, _p, _p, _p, _p, _p, _p]),
    'advoc_tanh_affine_f32': (ctypes.c_int, [_p, _p, _i64, _f32, _f32, _p]),
    'advoc_mel_dbnorm_f32': (ctypes.c_int, [_p, _i64, _f32, _f32, _f32, _p]),
    'advoc_conv_workspace_bytes': (_i64, [_p, _i32]),
    'advoc_conv_wgrad_ws_bytes': (_i64, [_p]),
    'advoc_conv_image_bytes': (_i64, [_p, _i32]),
    'advoc_conv_bias_fusable': (ctypes.c_int, [_p]),
    'advoc_conv_emits_images': (ctypes.c_int, [_p]),
    'advoc_conv_emits_dx_image': (ctypes.c_int, [_p]),
    'advoc_conv_gates_on_image': (ctypes.c_int, [_p]),
    'advoc_segmented_amax_f32': (ctypes.c_int, [_p, _p, _p, _i32, _p, _p]),
    'advoc_conv_weight_image_desc': (ctypes.c_int, [_p, _i32, _p]),
    'advoc_weight_images_f32': (ctypes.c_int, [_p, _p, _p, _i32, _p, _p, _p]),
    'advoc_weight_images_l1_f32': (ctypes.c_int, [_p, _p, _p, _i32, _p, _p, _p]),
    'advoc_conv_make_image': (ctypes.c_int, [_p, _i32, _p, _p]),
    'advoc_conv_forward': (ctypes.c_int, [_p, _p]),
    'advoc_conv_backward_data': (ctypes.c_int, [_p, _p, _p, _p, _i32, _i32, _p]),
    'advoc_conv_backward_weight': (ctypes.c_int, [_p, _p, _p, _p, _i32, _p]),
    'advoc_conv_backward_bias': (ctypes.c_int, [_p, _p, _p, _i32, _p]),
    'advoc_bn_forward': (ctypes.c_int, [_p, _i64, _i32, _p, _p, _f32, _p, _p, _p, _p, _p, _p]),
    'advoc_bn_backward': (ctypes.c_int, [_p, _p, _i64, _i32, _p, _p, _p, _p, _p, _i32, _p, _p]),
    'advoc_bn_forward_stats': (ctypes.c_int, [_p, _i64, _i32, _p, _p]),
    'advoc_bn_forward_finalize': (ctypes.c_int, [_p, _i64, _i32, _p, _p, _f32, _p, _p, _p, _p, _p]),
    'advoc_bn_backward_stats': (ctypes.c_int, [_p, _p, _i64, _i32, _p, _p, _p, _p, _i32, _p, _p]),
    'advoc_bn_backward_apply': (ctypes.c_int, [_p, _p, _i64, _i32, _p, _p, _p, _p, _i64, _p]),
    'advoc_conv_kernel_name': (ctypes.c_int, [_p, _i32, ctypes.c_char_p, _i32]),
    'advoc_gan_d_loss': (ctypes.c_int, [_p, _p, _i64, _p, _p, _p, _p]),
    'advoc_gan_g_loss': (ctypes.c_int, [_p, _i64, _p, _p, _i64, _f32, _f32, _p, _p, _i32, _p, _p]),
    'advoc_adam_tf_f32': (ctypes.c_int, [_p, _p, _p, _p, _i64, _f32, _f32, _f32, _f32, _f32, _p]),
    'advoc_sigmoid_f32': (ctypes.c_int, [_p, _p, _i64, _p]),
    'advoc_zero_f32': (ctypes.c_int, [_p, _i64, _p]),
    'advoc_dropout_mask_u8': (ctypes.c_int, [_p, _i64, ctypes.c_uint64, ctypes.c_uint64, _f32, _p]),
}

_lock = threading.Lock()
_lib = None


def load():
  """Loads libadvoc_hip.so once; raises AdvocHipError if it has not been built."""
  global _lib
  if _lib is not None:
    return _lib
  with _lock:
    if _lib is not None:
      return _lib
    if not os.path.isfile(LIB_PATH):
      raise AdvocHipError(
          'libadvoc_hip.so not found at {} -- build it first '
          '(python -c "import __graft_entry__ as g; g.build()" or make -C advoc_amd/csrc). '
          'advoc_amd has no CPU fallback for the hot path.'.format(LIB_PATH))
    # torch ships its own libamdhip64.so.7; import it FIRST so the process holds exactly one
    # HIP runtime (the one that owns torch's device memory and streams) and our library's
    # NEEDED libamdhip64.so.7 binds to it.  Loading /opt/rocm's copy first leaves two
    # runtimes in the process and ours reports hipErrorNoDevice.
    import torch  # noqa: F401
    try:
      lib = ctypes.CDLL(LIB_PATH)
    except OSError as e:
      raise AdvocHipError('cannot load {}: {}'.format(LIB_PATH, e))
    for name, (res, args) in PROTOTYPES.items():
      try:
        fn = getattr(lib, name)
      except AttributeError:
        raise AdvocHipError('{} does not export {} (stale build?)'.format(LIB_PATH, name))
      fn.restype = res
      fn.argtypes = args
    if lib.advoc_abi_version() != 1:
      raise AdvocHipError('ABI version mismatch: library reports {}'.format(lib.advoc_abi_version()))
    _lib = lib
  return _lib


def reload_env():
  """Makes the library re-read its ADVOC_* diagnostic switches from os.environ (it caches them on first use)."""
  load().advoc_tuning_reload()


ERR_UNSUPPORTED = -2       # ADVOC_ERR_UNSUPPORTED (include/advoc_hip.h)


def check(rc, what=''):
  if rc != 0:
    msg = load().advoc_error_string(rc).decode()
    if rc == -3:
      msg += ': ' + load().advoc_last_hip_error().decode()
    raise AdvocHipError('{} failed: {} ({})'.format(what or 'libadvoc_hip call', msg, rc))


def require_device(t, name='tensor'):
  """Every tensor handed to the C ABI must be a contiguous float32 tensor in HBM."""
  import torch
  if not isinstance(t, torch.Tensor) or not t.is_cuda:
    raise AdvocHipError('{} must be a torch tensor on a HIP device (no CPU path exists)'.format(name))
  if not t.is_contiguous():
    raise AdvocHipError('{} must be contiguous'.format(name))
  return t


def ptr(t):
  """Raw address of a tensor for the C ABI (NULL for None).  The kernels address dense row-major
  buffers: a strided view would be read as if it were contiguous, so it is refused here.  The returned
  address does NOT keep `t` alive: pass named tensors, never temporaries."""
  if t is None:
    return ctypes.c_void_p(0)
  if not t.is_contiguous():
    raise AdvocHipError('non-contiguous tensor (shape {}, strides {}) passed to the HIP library'.format(
        tuple(t.shape), tuple(t.stride())))
  return ctypes.c_void_p(t.data_ptr())


def stream():
  import torch
  return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def device():
  """The HIP device everything runs on; raises if there is none (no CPU fallback)."""
  import torch
  if not torch.cuda.is_available():
    raise AdvocHipError('no HIP device visible: advoc_amd runs its hot path only on MI355X (gfx950)')
  return torch.device('cuda', torch.cuda.current_device())
