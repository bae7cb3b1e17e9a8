"""profiles/<tag>_lagcov_pmc.json (read by bench.py for roofline.traffic) from the FETCH_SIZE /
WRITE_SIZE passes of tools/make_profiles.sh:  python tools/make_pmc_json.py r01"""
import json, os, re, sys
tag = sys.argv[1]
root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'profiles')

def counter(path, kernel, name):
  lines = open(path).read().split('\n')
  for i, ln in enumerate(lines):
    if ln.startswith(kernel):
      for nxt in lines[i + 1:i + 12]:
        if not nxt.startswith(' '):
          break
        m = re.match(r'\s+%s\s+n=(\d+)\s+avg=(\S+)' % name, nxt)
        if m:
          return ln.strip(), float(m.group(2))
  raise SystemExit('no %s for %s in %s' % (name, kernel, path))

out = {}
for key, kernel, algo, note in (
    ('lagcov', 'lagcov_split_kernel<true, 83, true, false, false, false>', 256000000,
     'reads the 256 MB of input (the four lag-group workgroups of a time slab share one XCD L2); '
     'writes 33 MB = 64 float32 partial slabs of 512 KB (one workgroup per CU and lag group walks '
     'three <= 8192-sample slabs and leaves one partial slab), summed in float64 by '
     'stats_finalize_kernel'),
    ('targets', 'lagcov_targets_mfma_kernel<true, false>', 260000000,
     'the second read of x (+ y): y^T x~, column sums and the channel maxima of the float16 kernel'),
    ('gram', 'gram_bf16x3_kernel<5>', 288000000,
     'C3 one-pass CCA moments: every input byte read once; 512 partial slabs of 15 KB'),
    ('project', 'cca_project_stream_kernel<3>', 328000000,
     'C3 transform: x and x2 read once, 40 MB of outputs written'),
    ('fir', 'fir_stream_kernel<true, 2, 2, false, false>', 312000000,
     'C4 decode, the FIR prediction: 307.2 MB of EEG read once (+ the 31 halo rows of every strip, '
     'mostly L2 hits) and 4.8 MB of predictions written')):
  name, fetch = counter(os.path.join(root, tag + '_hotkernels_pmc1.txt'), kernel, 'FETCH_SIZE')
  _, write = counter(os.path.join(root, tag + '_hotkernels_pmc2.txt'), kernel, 'WRITE_SIZE')
  out[key] = {
      'kernel': name,
      'fetch_size_kb_per_launch': fetch, 'write_size_kb_per_launch': write,
      'hbm_bytes_per_launch': 2 * fetch * 1024 + write * 1024,
      'algorithmic_bytes_per_launch': algo, 'note': note}
doc = dict(out['lagcov'])
doc['source'] = ('profiles/%s_hotkernels_pmc1.txt (FETCH_SIZE) and profiles/%s_hotkernels_pmc2.txt '
                 '(WRITE_SIZE): rocprofv3 --pmc, separate passes, tools/prof_kernels.py (whole chip)'
                 % (tag, tag))
doc['correction'] = ('gfx950: FETCH_SIZE counts 128-B read requests at 64 B -> x2 '
                     '(MI355X_MICROARCH.md, HBM); WRITE_SIZE exact for 16-B-per-lane streaming stores')
doc['lagcov_targets_mfma_kernel'] = out['targets']
doc['gram_bf16x3_kernel'] = out['gram']
doc['cca_project_stream_kernel'] = out['project']
doc['fir_stream_kernel'] = out['fir']
json.dump(doc, open(os.path.join(root, tag + '_lagcov_pmc.json'), 'w'), indent=1)
print(json.dumps(doc, indent=1))
