"""Pins the MelspecGAN generator and the repo's protobuf reader (§8b-6 / §8f-4) to the one
TensorFlow-written artefact the reference holds: models/melspecgan/infer.meta, decoded into
tests/golden/melspecgan_graph.json by tests/golden/make_melspecgan_graph.py (reference: models/melspecgan/conv2d.py:82-150,
train.py:156-177, infer.py)."""
import json
import os

import numpy as np
import pytest
import torch

from advoc_amd import tf_checkpoint as T
from oracle import melspecgan_torch as O

import tf_graph_interp as interp

HERE = os.path.dirname(os.path.abspath(__file__))
REF_META = '/root/reference/models/melspecgan/infer.meta'


@pytest.fixture(scope='module')
def graph():
  with open(os.path.join(HERE, 'golden', 'melspecgan_graph.json')) as f:
    return json.load(f)


def _nodes(graph):
  return dict((n['name'], n) for n in graph['nodes'])


def test_golden_is_what_the_reader_decodes_from_the_reference_file(graph):
  """In the build container: the committed JSON is exactly what read_meta_graph decodes from the real TF bytes."""
  if not os.path.isfile(REF_META):
    pytest.skip('reference not present (GPU box)')
  import importlib.util
  spec = importlib.util.spec_from_file_location('make_graph', os.path.join(HERE, 'golden', 'make_melspecgan_graph.py'))
  mod = importlib.util.module_from_spec(spec)
  spec.loader.exec_module(mod)
  fresh = json.loads(json.dumps(mod.build(REF_META), sort_keys=True))
  assert fresh == graph
  assert graph['tf_version'] == '1.12.0' and len(graph['nodes']) == 246


def test_reader_on_real_tensorflow_bytes(graph):
  """NodeDef / SaverDef byte strings cut from the TF-written file decode to the attributes TF's own tools print."""
  raw = graph['raw_nodes']
  n = T._parse_proto(bytes.fromhex(raw['G/upconv_1/conv2d_transpose']))
  assert n[1][0] == b'G/upconv_1/conv2d_transpose' and n[2][0] == b'Conv2DBackpropInput'
  attrs = dict((T._parse_proto(a)[1][0].decode(), T._parse_attr(T._parse_proto(a)[2][0])) for a in n[5])
  assert attrs['strides'] == [1, 2, 2, 1] and attrs['padding'] == 'SAME' and attrs['data_format'] == 'NHWC'
  assert attrs['dilations'] == [1, 1, 1, 1] and attrs['T'] == ('dtype', 1)
  assert attrs['_output_shapes'] == [('shape', [-1, 8, 10, 256])]            # negative (unknown) dims: 10-byte varints
  n = T._parse_proto(bytes.fromhex(raw['G/batch_normalization/FusedBatchNorm']))
  attrs = dict((T._parse_proto(a)[1][0].decode(), T._parse_attr(T._parse_proto(a)[2][0])) for a in n[5])
  assert attrs['is_training'] is False and abs(attrs['epsilon'] - 1e-3) < 1e-9
  n = T._parse_proto(bytes.fromhex(raw['G/Reshape/shape']))
  attrs = dict((T._parse_proto(a)[1][0].decode(), T._parse_attr(T._parse_proto(a)[2][0])) for a in n[5])
  assert attrs['value'].dtype == np.int32 and attrs['value'].tolist() == [-1, 4, 5, 512]   # tensor_content bytes
  n = T._parse_proto(bytes.fromhex(raw['save/SaveV2/tensor_names']))
  attrs = dict((T._parse_proto(a)[1][0].decode(), T._parse_attr(T._parse_proto(a)[2][0])) for a in n[5])
  assert len(attrs['value']) == 27 and attrs['value'][0] == b'G/batch_normalization/beta' and attrs['value'][-1] == b'global_step'
  sd = T._parse_proto(bytes.fromhex(graph['raw_saver_def']))
  assert sd[1][0] == b'save/Const:0' and sd[3][0] == b'save/restore_all' and sd[4][0] == 5 and sd[7][0] == 2


def test_read_meta_graph_api_on_the_reference_file():
  if not os.path.isfile(REF_META):
    pytest.skip('reference not present (GPU box)')
  meta = T.read_meta_graph(REF_META)
  v = T.meta_graph_variables(meta)
  assert list(v)[:2] == ['G/z_proj/W', 'G/z_proj/b'] and v['G/z_proj/W'] == (np.dtype('float32'), (100, 10240))
  assert v['global_step'] == (np.dtype('int64'), ())
  assert meta['collections']['trainable_variables'][0] == 'G/z_proj/W:0' and len(meta['collections']['variables']) == 27
  from advoc_amd.melspecgan import MelspecGANGenerator
  T.check_model_against_meta_graph(MelspecGANGenerator().variable_specs(), meta, scope='G/')
  with pytest.raises(ValueError, match='shape'):
    T.check_model_against_meta_graph(MelspecGANGenerator(dim=32).variable_specs(), meta, scope='G/')
  with pytest.raises(ValueError, match='no counterpart'):
    T.check_model_against_meta_graph(MelspecGANGenerator(batchnorm=False).variable_specs(), meta, scope='G/')


def test_variable_table_matches_oracle_and_product(graph):
  """Names, shapes AND creation order of the 26 generator variables TensorFlow created (tf.train.Saver restores by name)."""
  tfv = [(v['name'], tuple(v['shape'])) for v in graph['variables'] if v['name'].startswith('G/')]
  assert len(tfv) == 26 and all(v['dtype'] == 'float32' for v in graph['variables'] if v['name'].startswith('G/'))
  assert [(n, tuple(s)) for n, s in O.variable_specs(64, 5, True)] == tfv
  from advoc_amd.melspecgan import MelspecGANGenerator
  assert [(n, tuple(s)) for n, s in MelspecGANGenerator().variable_specs()] == tfv
  # what the Saver of the meta graph stores: the same names + global_step
  nodes = _nodes(graph)
  saved = set(nodes['save/SaveV2/tensor_names']['attrs']['value'])
  assert saved == set(n for n, _ in tfv) | {'global_step'}
  trainable = [s[:-2] for s in graph['collections']['trainable_variables']]
  assert trainable == [n for n, _ in tfv if 'moving_' not in n]


def test_hyper_parameters_match_oracle_and_product(graph):
  nodes = _nodes(graph)
  from advoc_amd import melspecgan as prod
  assert nodes['z']['attrs']['shape'] == ['shape', [-1, O.Z_DIM]] and prod.Z_DIM == O.Z_DIM == 100
  assert nodes['G/Reshape/shape']['attrs']['value']['value'] == [-1, 4, 5, 512]
  for i in range(4):
    bn = nodes['G/batch_normalization%s/FusedBatchNorm' % ('' if i == 0 else '_%d' % i)]['attrs']
    assert bn['is_training'] is False and bn['data_format'] == 'NHWC'
    assert abs(bn['epsilon'] - O.BN_EPS) < 1e-9 and abs(bn['epsilon'] - prod.BN_EPS) < 1e-9
  h, w = 4, 5
  for i, c in enumerate((256, 128, 64, 1)):
    cv = nodes['G/upconv_%d/conv2d_transpose' % (i + 1)]
    assert cv['op'] == 'Conv2DBackpropInput'
    a = cv['attrs']
    assert a['strides'] == [1, 2, 2, 1] and a['padding'] == 'SAME' and a['data_format'] == 'NHWC' and a['dilations'] == [1, 1, 1, 1]
    shp = [nodes['G/upconv_%d/conv2d_transpose/output_shape/%d' % (i + 1, k)]['attrs']['value']['value'] for k in (1, 2, 3)]
    h, w = 2 * h, 2 * w
    assert shp == [h, w, c]                             # output = stride x input, as the oracle / product assume
    for init in ('mean', 'stddev'):
      v = nodes['G/upconv_%d/W/Initializer/random_normal/%s' % (i + 1, init)]['attrs']['value']['value']
      assert abs(v - (0.02 if init == 'stddev' else 0.0)) < 1e-9
  assert (h, w) == (64, 80)
  # tail: tanh -> (x + 1) * 0.5 -> identity 'G_z'   (util.feats_denorm, train.py:163-164)
  assert nodes['G_z']['inputs'] == ['mul'] and nodes['mul']['inputs'] == ['add', 'mul/y'] and nodes['add']['inputs'] == ['G/Tanh', 'add/y']
  assert nodes['add/y']['attrs']['value']['value'] == 1.0 and nodes['mul/y']['attrs']['value']['value'] == 0.5
  assert nodes['samp_z']['op'] == 'Add' and nodes['samp_z_n']['op'] == 'Placeholder'


@pytest.mark.parametrize('batch', [1, 3])
def test_oracle_equals_the_tensorflow_graph(graph, batch):
  """oracle.melspecgan_torch.generator == the TF-written graph evaluated op by op (float64, random parameters with
  non-trivial BN statistics): pins layer order, BN placement, the transposed-conv SAME rule and the output mapping."""
  P = O.init_params(dim=64, seed=3, dtype=torch.float64)
  z = torch.randn(batch, 100, generator=torch.Generator().manual_seed(7), dtype=torch.float64)
  want = interp.run(graph, 'G_z', {'z': z}, P)
  got = O.generator(P, z, dim=64)
  assert tuple(want.shape) == tuple(got.shape) == (batch, 64, 80, 1)
  assert float((got - want).abs().max()) < 1e-10     # float64 round-off of two summation orders
  assert float(want.std()) > 1e-3                      # not a degenerate comparison
