"""A tiny evaluator for the forward part of a TF1 GraphDef as decoded into tests/golden/melspecgan_graph.json
(test infrastructure).  It walks the node list TensorFlow WROTE -- not this repo's transcription of the model -- and
executes each op by its TensorFlow definition on torch-CPU tensors:

  Conv2DBackpropInput  by definition "the gradient of conv2d w.r.t. its input": evaluated literally as the autograd
                       gradient of a SAME-padded (TF rule: total = max((out-1) s + k - in, 0), before = total // 2)
                       strided torch conv2d -- independent of the conv_transpose2d + crop form the oracle uses
  FusedBatchNorm       is_training=False: (x - mean) * gamma * rsqrt(var + epsilon) + beta
"""
import torch
import torch.nn.functional as F


def _val(a):
  return torch.tensor(a['value'], dtype=getattr(torch, a['dtype'])).reshape(a['shape'])


def _same_pad(n_in, n_out, k, s):
  total = max((n_out - 1) * s + k - n_in, 0)
  return total // 2, total - total // 2


def conv2d_backprop_input(sizes, w, dy, strides, padding, data_format):
  assert data_format == 'NHWC' and padding == 'SAME' and strides[0] == 1 and strides[3] == 1
  b, h, wd, c = [int(v) for v in sizes]
  kh, kw, cin, cout = w.shape                     # HWIO of the FORWARD conv: I = channels of `sizes`
  assert cin == c and cout == dy.shape[3]
  x = torch.zeros(b, c, h, wd, dtype=dy.dtype, requires_grad=True)
  pt, pb = _same_pad(h, dy.shape[1], kh, strides[1])
  pl, pr = _same_pad(wd, dy.shape[2], kw, strides[2])
  y = F.conv2d(F.pad(x, (pl, pr, pt, pb)), w.permute(3, 2, 0, 1).contiguous(), None, stride=(strides[1], strides[2]))
  assert tuple(y.shape) == (b, cout, dy.shape[1], dy.shape[2]), (y.shape, dy.shape)
  g, = torch.autograd.grad(y, x, dy.permute(0, 3, 1, 2).contiguous())
  return g.permute(0, 2, 3, 1).contiguous()


def run(graph, fetch, feeds, variables):
  """graph: the decoded JSON; fetch: node name; feeds: {placeholder: tensor}; variables: {name: tensor}."""
  nodes = dict((n['name'], n) for n in graph['nodes'])
  cache = {}

  def ev(ref):
    name = ref.split(':')[0].lstrip('^')
    if name in cache:
      return cache[name]
    n = nodes[name]
    op, a = n['op'], n['attrs']
    i = [ev(r) for r in n['inputs'] if not r.startswith('^')] if op not in ('VariableV2',) else []
    if op == 'Placeholder':
      out = feeds[name]
    elif op == 'VariableV2':
      out = variables[name]
    elif op == 'Identity':
      out = i[0]
    elif op == 'Const':
      out = _val(a['value'])
    elif op == 'MatMul':
      x, y = i
      out = (x.t() if a['transpose_a'] else x) @ (y.t() if a['transpose_b'] else y)
    elif op == 'BiasAdd':
      assert a['data_format'] == 'NHWC'
      out = i[0] + i[1]
    elif op == 'Reshape':
      out = i[0].reshape([int(v) for v in i[1]])
    elif op == 'FusedBatchNorm':
      assert a['is_training'] is False and a['data_format'] == 'NHWC'
      x, gamma, beta, mean, var = i
      out = (x - mean) * (gamma * torch.rsqrt(var + a['epsilon'])) + beta
    elif op == 'Relu':
      out = torch.relu(i[0])
    elif op == 'Tanh':
      out = torch.tanh(i[0])
    elif op == 'Add':
      out = i[0] + i[1].to(i[0].dtype)
    elif op == 'Mul':
      out = i[0] * i[1].to(i[0].dtype)
    elif op == 'Shape':
      out = torch.tensor(list(i[0].shape), dtype=torch.int32)
    elif op == 'StridedSlice':
      x, b, e, s = i
      assert a['shrink_axis_mask'] == 1 and a['begin_mask'] == 0 and a['end_mask'] == 0 and int(s[0]) == 1
      out = x[int(b[0])]
    elif op == 'Pack':
      out = torch.stack([torch.as_tensor(v).reshape(()) for v in i])
    elif op == 'Conv2DBackpropInput':
      out = conv2d_backprop_input(i[0], i[1], i[2], a['strides'], a['padding'], a['data_format'])
    else:
      raise NotImplementedError(op)
    cache[name] = out
    return out
  return ev(fetch)
