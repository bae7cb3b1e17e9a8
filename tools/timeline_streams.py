"""Per-queue kernel timeline of the last part of a rocprofv3 kernel trace (which stream waits
for which):  python tools/timeline_streams.py <dir> [t_from_end_us]"""
import csv, glob, os, sys
d = sys.argv[1]
span = float(sys.argv[2]) if len(sys.argv) > 2 else 6000.0
rows = []
for f in glob.glob(os.path.join(d, '**', '*kernel_trace.csv'), recursive=True):
  for r in csv.DictReader(open(f)):
    rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'],
                 r.get('Queue_Id', '?')))
rows.sort()
t_end = rows[-1][1]
rows = [r for r in rows if r[0] >= t_end - span * 1e3]
t0 = rows[0][0]
last = {}
for s, e, name, q in rows:
  name = name.replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0][:40]
  gap = (s - last[q]) / 1e3 if q in last else 0.0
  print('q%-3s %9.1f  dur %8.1f  gap %7.1f  %s' % (q, (s - t0) / 1e3, (e - s) / 1e3, gap, name))
  last[q] = e
