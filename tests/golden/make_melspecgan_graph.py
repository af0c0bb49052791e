"""Generates tests/golden/melspecgan_graph.json from the ONLY TensorFlow-written artefact the reference holds:
/root/reference/models/melspecgan/infer.meta (a MetaGraphDef exported by models/melspecgan/infer.py / train.py:156-177
with TensorFlow 1.12).  Run in the build container (the reference is not on the GPU box):

    python tests/golden/make_melspecgan_graph.py

The JSON is DATA decoded from that file by the repo's own protobuf reader (advoc_amd.tf_checkpoint.read_meta_graph):
every node's op / inputs / attributes (constants as nested lists), the variable table, the collections, the saver
definition, plus a few raw NodeDef byte strings (hex) so that the reader is also exercised on real TF bytes where the
reference is absent.  No reference source text is stored.
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from advoc_amd import tf_checkpoint as T   # noqa: E402

SRC = '/root/reference/models/melspecgan/infer.meta'
OUT = os.path.join(HERE, 'melspecgan_graph.json')
RAW_NODES = ('G/upconv_1/conv2d_transpose', 'G/batch_normalization/FusedBatchNorm', 'G/Reshape/shape',
             'G/z_proj/MatMul', 'save/SaveV2/tensor_names', 'z')


def plain(v):
  if isinstance(v, np.ndarray):
    return dict(dtype=str(v.dtype), shape=list(v.shape), value=v.tolist())
  if isinstance(v, bytes):
    return v.decode('utf-8', 'replace')
  if isinstance(v, tuple):
    return [plain(x) for x in v]
  if isinstance(v, list):
    return [plain(x) for x in v]
  if isinstance(v, (np.floating, np.integer)):
    return v.item()
  return v


def build(src=SRC):
  meta = T.read_meta_graph(src)
  nodes = []
  for name, n in meta['nodes'].items():
    nodes.append(dict(name=name, op=n['op'], inputs=n['inputs'],
                      attrs=dict((k, plain(v)) for k, v in sorted(n['attrs'].items()))))
  variables = [dict(name=k, dtype=str(dt), shape=list(shape)) for k, (dt, shape) in T.meta_graph_variables(meta).items()]
  # raw bytes of a few NodeDefs (GraphDef field 1 entries), for reader tests that travel without the reference
  with open(src, 'rb') as f:
    mg = T._parse_proto(f.read())
  raw = {}
  for nb in T._parse_proto(mg[2][0])[1]:
    nm = T._parse_proto(nb)[1][0].decode()
    if nm in RAW_NODES:
      raw[nm] = nb.hex()
  return dict(source='models/melspecgan/infer.meta (TensorFlow %s MetaGraphDef, %d bytes)' % (meta['tf_version'], os.path.getsize(src)),
              tf_version=meta['tf_version'], nodes=nodes, variables=variables, collections=meta['collections'],
              saver=meta['saver'], raw_saver_def=mg[3][0].hex(), raw_nodes=raw)


if __name__ == '__main__':
  with open(OUT, 'w') as f:
    json.dump(build(), f, indent=0, sort_keys=True)
    f.write('\n')
  print('wrote', OUT, os.path.getsize(OUT), 'bytes')
