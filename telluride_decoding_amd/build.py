"""Builds libtd_hotpath.so (hand-written HIP for gfx950) in-tree with hipcc."""
import glob
import os
import shutil
import subprocess

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, 'csrc')
LIB = os.path.join(PKG, 'libtd_hotpath.so')
ARCH = 'gfx950'


def _hipcc():
  for cand in (shutil.which('hipcc'), '/opt/rocm/bin/hipcc'):
    if cand and os.path.exists(cand):
      return cand
  raise RuntimeError('hipcc not found: cannot build the MI355X hot-path library')


def sources():
  return sorted(glob.glob(os.path.join(CSRC, '*.hip')))


def is_stale():
  if not os.path.exists(LIB):
    return True
  t = os.path.getmtime(LIB)
  deps = sources() + glob.glob(os.path.join(CSRC, '*.h')) + glob.glob(
      os.path.join(ROOT, 'include', '*.h'))
  return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
  """Compiles every HIP source for gfx950 into one shared library."""
  if not force and not is_stale():
    return LIB
  objs = []
  obj_dir = os.path.join(PKG, 'csrc', '_obj')
  os.makedirs(obj_dir, exist_ok=True)
  # -fno-slp-vectorize: hipcc's SLP pass packs adjacent f32 FMAs into v_pk_fma_f32, which
  # issues slower than the two v_fma_f32 it replaces on gfx950 (measured on the FIR kernel)
  flags = ['--offload-arch=' + ARCH, '-O3', '-fPIC', '-std=c++17', '-fno-slp-vectorize',
           '-Wno-unused-result', '-I' + os.path.join(ROOT, 'include'), '-I' + CSRC]
  flags += os.environ.get('TD_EXTRA_HIPCC_FLAGS', '').split()   # development: ablation macros
  procs = []
  for src in sources():
    obj = os.path.join(obj_dir, os.path.basename(src) + '.o')
    objs.append(obj)
    cmd = [_hipcc()] + flags + ['-c', src, '-o', obj]
    if verbose:
      print(' '.join(cmd))
    procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE,
                                        stderr=subprocess.STDOUT)))
  for src, proc in procs:
    out, _ = proc.communicate()
    if proc.returncode != 0:
      raise RuntimeError('hipcc failed on %s:\n%s' % (src, out.decode()))
  cmd = [_hipcc(), '--offload-arch=' + ARCH, '-shared', '-fPIC', '-o', LIB] + objs
  subprocess.run(cmd, check=True, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
  return LIB


if __name__ == '__main__':
  print(build(force=True, verbose=True))
