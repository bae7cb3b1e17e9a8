"""Start-to-start spacing of the lag kernel launches of a rocprofv3 kernel trace of the pipelined
bench (the first launches: warm-up, then the timed region), and where the solve chains end:
    python tools/pipeline_steps.py <t_kernel_trace.csv>"""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
lag = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in rows if 'lagcov_split_kernel' in r['Kernel_Name'])
emit = sorted(int(r['End_Timestamp']) for r in rows if 'ridge_emit_kernel' in r['Kernel_Name'])
t0 = lag[0][0]
prev = None
for i, (s, e) in enumerate(lag):
  print('lag launch %2d: start %9.1f us, dur %7.1f, since previous start %7.1f' % (i, (s - t0) / 1e3, (e - s) / 1e3, (s - prev) / 1e3 if prev else 0))
  prev = s
print('solves end at', ' '.join('%.0f' % ((t - t0) / 1e3) for t in emit))
