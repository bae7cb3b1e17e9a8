"""profiles/<tag>_parity.json from gpurun_out/parity.jsonl (the rows tests/parity_log.py wrote during a full
`python -m pytest tests -m gpu` run): the last row of every case.   python tools/make_parity_json.py r05 [n_tests]"""
import json, os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
n_tests = sys.argv[2] if len(sys.argv) > 2 else '?'
rows = {}
with open(os.path.join(root, 'gpurun_out', 'parity.jsonl')) as f:
  for line in f:
    line = line.strip()
    if line:
      r = json.loads(line)
      rows[r['case']] = r
doc = {'source': 'tests/parity_log.py rows of the full `python -m pytest tests -m gpu` run on one MI355X (%s GPU tests, '
                 'round %s): |gpu - ref64| etc. per case, the last run of each case' % (n_tests, tag.lstrip('r0')),
       'rows': list(rows.values())}
json.dump(doc, open(os.path.join(root, 'profiles', tag + '_parity.json'), 'w'), indent=1)
print('%d cases -> profiles/%s_parity.json' % (len(rows), tag))
