"""Compact per-kernel resource table: python tools/kres.py <file.hip> [name-filter]."""
import re, subprocess, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1]; filt = sys.argv[2] if len(sys.argv) > 2 else ''
out = subprocess.run(['hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fno-slp-vectorize', '-I' + ROOT + '/include',
                      '-I' + ROOT + '/telluride_decoding_amd/csrc', '-c', src, '-o', '/tmp/kres.o',
                      '-Rpass-analysis=kernel-resource-usage'], stderr=subprocess.PIPE, text=True).stderr
cur = None; rows = {}
for line in out.splitlines():
  m = re.search(r'remark:\s+(.*?): (.*?) \[-Rpass', line)
  if not m:
    if 'error' in line: print(line)
    continue
  k, v = m.group(1).strip(), m.group(2).strip()
  if k == 'Function Name':
    cur = subprocess.run(['c++filt', v], stdout=subprocess.PIPE, text=True).stdout.strip()
    cur = re.sub(r'\(anonymous namespace\)::', '', cur).split('(')[0]
    rows[cur] = {}
  elif cur: rows[cur][k] = v
print('%-48s %5s %5s %4s %7s %6s %6s' % ('kernel', 'VGPR', 'AGPR', 'occ', 'LDS', 'vspill', 'scratch'))
for n, r in rows.items():
  if filt in n:
    print('%-48s %5s %5s %4s %7s %6s %6s' % (n[-48:], r.get('VGPRs'), r.get('AGPRs'), r.get('Occupancy [waves/SIMD]'),
          r.get('LDS Size [bytes/block]'), r.get('VGPRs Spill'), r.get('ScratchSize [bytes/lane]')))
