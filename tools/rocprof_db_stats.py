"""Turns a rocprofv3 results .db (kernel trace) into the per-kernel stats CSV kept under
profiles/ (name, calls, total/avg/min/max duration in us, share of GPU time)."""
import csv
import sqlite3
import sys


def main(db_path, out_path):
  db = sqlite3.connect(db_path)
  cur = db.cursor()
  rows = cur.execute(
      'select name, count(*), sum(end - start), avg(end - start), min(end - start), '
      'max(end - start) from kernels group by name order by 3 desc').fetchall()
  total = sum(r[2] for r in rows) or 1
  with open(out_path, 'w', newline='') as f:
    w = csv.writer(f)
    w.writerow(['Name', 'Calls', 'TotalDurationUs', 'AverageUs', 'MinUs', 'MaxUs', 'Percentage'])
    for name, calls, tot, avg, mn, mx in rows:
      w.writerow([name, calls, '%.3f' % (tot / 1e3), '%.3f' % (avg / 1e3), '%.3f' % (mn / 1e3),
                  '%.3f' % (mx / 1e3), '%.2f' % (100.0 * tot / total)])


if __name__ == '__main__':
  main(sys.argv[1], sys.argv[2])
