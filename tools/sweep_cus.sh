#!/bin/bash
# CU partition sweep of the pipelined fit:  tools/sweep_cus.sh 32 64 96
for cus in "$@"; do
  python bench.py --steps 100 --no-cpu --no-extra --no-decode --solve-cus $cus --solve-streams ${STREAMS:-2} 2>/dev/null | tail -1 > /tmp/sweep_$cus.json
  python - $cus <<'PY'
import json, sys
d = json.loads(open('/tmp/sweep_%s.json' % sys.argv[1]).read())
print('solve-cus', sys.argv[1], {k: round(d.get(k), 3) for k in ('ms_per_step', 'serial_ms_per_step', 'accumulate_only_ms_per_step')},
      'kernel ms', round(d['roofline']['avg_launch_ms'], 3))
PY
done
