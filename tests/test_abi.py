"""The C-ABI library builds, loads and exports exactly what include/advoc_hip.h declares.
CPU only: no kernel is launched."""
import os
import re
import subprocess

import pytest

from advoc_amd import _lib


def _header_symbols():
  src = open(_lib.HEADER_PATH).read()
  src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
  return sorted(set(re.findall(r'\b(advoc_[a-z0-9_]+)\s*\(', src)))


@pytest.fixture(scope='module')
def lib():
  if not os.path.isfile(_lib.LIB_PATH):
    import __graft_entry__ as g
    g.build()
  return _lib.load()


def test_header_and_binding_agree(lib):
  syms = _header_symbols()
  assert len(syms) >= 7
  assert sorted(_lib.PROTOTYPES) == syms, 'PROTOTYPES in _lib.py must list every header symbol'


def test_layer_struct_layout_matches_the_header(tmp_path):
  """The ctypes mirror of struct advoc_conv_layer (advoc_amd/_lib.py:ConvLayer) against the header compiled as C: total
  size and the offset of every member the binding names -- a member added to one side only would otherwise show up as
  wrong numbers on the GPU, not as an error."""
  import ctypes
  fields = [f[0] for f in _lib.ConvLayer._fields_]
  src = ['#include <stdio.h>', '#include <stddef.h>', '#include "advoc_hip.h"', 'int main(void) {',
         '  printf("sizeof %zu\\n", sizeof(advoc_conv_layer));']
  src += ['  printf("%s %%zu\\n", offsetof(advoc_conv_layer, %s));' % (f, f) for f in fields]
  src += ['  printf("image_out %zu\\n", sizeof(((advoc_conv_layer*)0)->y_img[0]));', '  return 0;', '}']
  c = tmp_path / 'layout.c'
  c.write_text('\n'.join(src))
  exe = tmp_path / 'layout'
  subprocess.run(['gcc', '-std=c99', '-I', os.path.dirname(_lib.HEADER_PATH), str(c), '-o', str(exe)], check=True)
  out = dict(line.split() for line in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.splitlines())
  assert int(out['sizeof']) == ctypes.sizeof(_lib.ConvLayer)
  assert int(out['image_out']) == ctypes.sizeof(_lib.ImageOut)
  for f in fields:
    assert int(out[f]) == getattr(_lib.ConvLayer, f).offset, f


def test_every_header_symbol_is_exported(lib):
  out = subprocess.check_output(['nm', '-D', '--defined-only', _lib.LIB_PATH]).decode()
  exported = set(l.split()[-1] for l in out.splitlines() if ' T ' in l)
  for s in _header_symbols():
    assert s in exported, s
    assert hasattr(lib, s)


def test_version_and_errors(lib):
  assert lib.advoc_abi_version() == 1
  assert lib.advoc_target_arch() == b'gfx950'
  assert lib.advoc_error_string(0) == b'ok'
  for code in (-1, -2, -3, -4, -99):
    assert len(lib.advoc_error_string(code)) > 0


def test_code_object_is_gfx950_only():
  out = subprocess.check_output(['strings', '-a', _lib.LIB_PATH]).decode()
  targets = set(re.findall(r'amdgcn-amd-amdhsa--(gfx[0-9a-f]+)', out))
  assert targets == {'gfx950'}, targets


def test_null_and_shape_errors_need_no_gpu(lib):
  # argument validation happens before any HIP call
  assert lib.advoc_stft_mag_f32(None, 1, 1024, None, None, 1024, 256, 4, None, None) == -4
  assert lib.advoc_stft_mag_f32(None, 1, 700, None, None, 1024, 256, 0, None, None) == 0
  assert lib.advoc_matmul_nt_f32(None, None, None, 1, 1, 1, None) == -4
  one = 1  # a non-null dummy address; never dereferenced on these paths
  assert lib.advoc_stft_mag_f32(one, 1, 1024, one, one, 1000, 250, 4, one, None) == -2
  assert lib.advoc_stft_mag_f32(one, -1, 1024, one, one, 1024, 256, 4, one, None) == -1
  assert lib.advoc_stft_mag_f32(one, 0, 1024, one, one, 1024, 256, 4, one, None) == 0
  import ctypes
  import numpy as np
  tw = (ctypes.c_float * 2048)()
  assert lib.advoc_stft_twiddle_host(tw, 1024) == 0
  t = np.frombuffer(tw, dtype=np.float32).reshape(1024, 2)
  e = np.arange(1024)
  assert np.array_equal(t[:, 0], np.cos(2 * np.pi * e / 1024).astype(np.float32))
  assert np.array_equal(t[:, 1], np.sin(2 * np.pi * e / 1024).astype(np.float32))
  assert lib.advoc_stft_twiddle_host(tw, 512) == -2
  assert lib.advoc_matmul_nt_f32(one, one, one, 4, 0, 3, None) == -1


def test_product_refuses_cpu_tensors():
  import torch
  from advoc_amd import spectral
  if torch.cuda.is_available():
    pytest.skip('only meaningful without a HIP device')
  with pytest.raises(_lib.AdvocHipError):
    spectral.stft_tf(torch.zeros(1, 2048, 1, 1), 1024, 256)
