"""Accumulate-queue kernel timeline and solve-queue chain lengths of a rocprofv3 kernel trace of
the pipelined bench:  python tools/pipeline_timeline.py <t_kernel_trace.csv>"""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
def nm(r):
  return r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0][:36]
byq = collections.defaultdict(list)
for r in rows:
  byq[r['Queue_Id']].append(r)
accq = max(byq, key=lambda q: sum(1 for r in byq[q] if 'lagcov_mfma' in r['Kernel_Name']))
solq = max(byq, key=lambda q: sum(1 for r in byq[q] if 'chol_update' in r['Kernel_Name']))
q3 = sorted(byq[accq], key=lambda r: int(r['Start_Timestamp']))
q4 = sorted(byq[solq], key=lambda r: int(r['Start_Timestamp']))
t0 = int(q3[0]['Start_Timestamp'])
prev = None
print('--- accumulate queue', accq)
for r in q3[-16:]:
  s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
  print('%9.1f dur %8.1f gap %7.1f %s' % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev) / 1e3 if prev else 0, nm(r)))
  prev = e
print('--- solve queue', solq, '(chains from the first kernel after a >100 us idle)')
chain = None
prev_e = None
for r in q4:
  s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
  if prev_e is None or s - prev_e > 100000:
    if chain:
      print('chain %9.1f .. %9.1f  len %7.1f busy %7.1f  idle before %6.1f' % (
          (chain[0] - t0) / 1e3, (chain[1] - t0) / 1e3, (chain[1] - chain[0]) / 1e3, chain[2] / 1e3, chain[3] / 1e3))
    chain = [s, e, 0, (s - prev_e) if prev_e else 0]
  chain[1] = e
  chain[2] += e - s
  prev_e = e
