"""Summarises rocprofv3 CSV output (kernel trace or PMC) per kernel name."""
import csv, glob, os, sys, collections

d = sys.argv[1]
pmc = '--pmc' in sys.argv

def short(n):
  n = n.replace('(anonymous namespace)::', '').replace('void ', '')
  return n.split('(')[0][:60]

if not pmc:
  files = glob.glob(os.path.join(d, '**', '*kernel_trace.csv'), recursive=True)
  agg = collections.defaultdict(list)
  for f in files:
    for r in csv.DictReader(open(f)):
      agg[short(r['Kernel_Name'])].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
  tot = sum(sum(v) for v in agg.values()) or 1
  print('%-60s %6s %10s %10s %10s %10s %6s' % ('kernel', 'calls', 'total_us', 'avg_us', 'min_us', 'max_us', '%'))
  for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    print('%-60s %6d %10.1f %10.2f %10.2f %10.2f %6.2f' % (k, len(v), sum(v), sum(v) / len(v), min(v), max(v), 100 * sum(v) / tot))
else:
  files = glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True)
  agg = collections.defaultdict(lambda: collections.defaultdict(list))
  for f in files:
    for r in csv.DictReader(open(f)):
      agg[short(r['Kernel_Name'])][r['Counter_Name']].append(float(r['Counter_Value']))
  for k, cs in agg.items():
    print(k)
    for c, v in cs.items():
      print('    %-36s n=%-5d avg=%-16.6g sum=%.6g' % (c, len(v), sum(v) / len(v), sum(v)))
