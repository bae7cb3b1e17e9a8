"""Prints the kernel timeline (start offset, duration, gap to the previous kernel) of the last
repetitions in a rocprofv3 kernel-trace CSV:  python tools/timeline.py <dir> [n_last]"""
import csv, glob, os, sys
d = sys.argv[1]
n_last = int(sys.argv[2]) if len(sys.argv) > 2 else 12
rows = []
for f in glob.glob(os.path.join(d, '**', '*kernel_trace.csv'), recursive=True):
  for r in csv.DictReader(open(f)):
    rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']))
rows.sort()
rows = rows[-n_last:]
t0 = rows[0][0]
prev_end = None
for s, e, name in rows:
  name = name.replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0][:48]
  gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
  print('%10.2f us  dur %8.2f  gap %7.2f  %s' % ((s - t0) / 1e3, (e - s) / 1e3, gap, name))
  prev_end = e
