"""The library's environment switches: every ADVOC_* variable the product reads is listed there AND
set by some test -- a switch nobody exercises is a dispatch path that rots unseen.  CPU only."""
import glob
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DIAG_ONLY = {'ADVOC_H3_ABLATE', 'ADVOC_H3_LOG', 'ADVOC_H3_PATCH_ABLATE', 'ADVOC_H3_SKIP_PREP'}
READ = re.compile(r'getenv\("(ADVOC_[A-Z0-9_]+)"|env_int\("(ADVOC_[A-Z0-9_]+)"|environ\.get\(\'(ADVOC_[A-Z0-9_]+)\'|'
                  r'environ\[\'(ADVOC_[A-Z0-9_]+)\'\]|\'(ADVOC_[A-Z0-9_]+)\' in os\.environ')


def product_switches():
  out = set()
  for f in glob.glob(os.path.join(ROOT, 'advoc_amd', '*.py')) + glob.glob(os.path.join(ROOT, 'advoc_amd', 'csrc', '*.hip')):
    for m in READ.finditer(open(f).read()):
      out.add(next(g for g in m.groups() if g))
  return out


def test_every_switch_is_documented_and_exercised():
  switches = product_switches()
  doc = open(os.path.join(ROOT, 'INTEGRATION.md')).read()
  tests = ''.join(open(f).read() for f in glob.glob(os.path.join(ROOT, 'tests', 'test_*.py')))
  missing_doc = sorted(s for s in switches if '`%s`' % s not in doc)
  assert not missing_doc, 'switches read by the product but not listed in INTEGRATION.md section 4: %s' % missing_doc
  untested = sorted(s for s in switches - DIAG_ONLY if not re.search(r'\b%s\b' % s, tests))
  assert not untested, 'switches no test sets: %s' % untested
  assert len(switches - DIAG_ONLY) <= 38, 'the list only shrinks: %d' % len(switches - DIAG_ONLY)


def test_library_path_override(tmp_path):
  """ADVOC_HIP_LIB: another build of the same library; a path that does not exist is an error, never a fallback."""
  from advoc_amd import _lib
  copy = tmp_path / 'libadvoc_hip_copy.so'
  copy.write_bytes(open(_lib.LIB_PATH, 'rb').read())
  code = 'from advoc_amd import _lib; print(_lib.LIB_PATH); print(_lib.load().advoc_target_arch().decode())'
  env = dict(os.environ, ADVOC_HIP_LIB=str(copy), PYTHONPATH=ROOT)
  out = subprocess.run([sys.executable, '-c', code], env=env, capture_output=True, text=True, cwd=ROOT)
  assert out.returncode == 0 and out.stdout.split() == [str(copy), 'gfx950'], (out.stdout, out.stderr[-400:])
  env['ADVOC_HIP_LIB'] = str(tmp_path / 'nope.so')
  bad = subprocess.run([sys.executable, '-c', code], env=env, capture_output=True, text=True, cwd=ROOT)
  assert bad.returncode != 0


def _free_port():
  import socket
  s = socket.socket()
  s.bind(('127.0.0.1', 0))
  p = s.getsockname()[1]
  s.close()
  return p


def test_dp_reserve_becomes_the_library_switch():
  """ADVOC_DP_RESERVE_CUS=k is honoured by DataParallel only when there is more than one rank (or ADVOC_DP_FORCE): it becomes
  ADVOC_RESERVE_CUS for the persistent launches; an explicit ADVOC_RESERVE_CUS wins."""
  code = ('import os\nfrom advoc_amd.parallel import DataParallel\n'
          'dp = DataParallel().init_from_env(backend="gloo")\n'
          'print(dp.reserve_source, os.environ.get("ADVOC_RESERVE_CUS"))\n')
  base = dict(os.environ, PYTHONPATH=ROOT, RANK='0', WORLD_SIZE='1', LOCAL_RANK='0', MASTER_ADDR='127.0.0.1',
              MASTER_PORT=str(_free_port()), ADVOC_DP_FORCE='1', ADVOC_DP_RESERVE_CUS='16')
  base.pop('ADVOC_RESERVE_CUS', None)
  out = subprocess.run([sys.executable, '-c', code], env=base, capture_output=True, text=True, cwd=ROOT)
  assert out.returncode == 0 and out.stdout.split()[-2:] == ['dp', '16'], (out.stdout, out.stderr[-400:])
  out = subprocess.run([sys.executable, '-c', code], env=dict(base, ADVOC_RESERVE_CUS='24', MASTER_PORT=str(_free_port())),
                       capture_output=True, text=True, cwd=ROOT)
  assert out.returncode == 0 and out.stdout.split()[-2:] == ['user', '24'], (out.stdout, out.stderr[-400:])
