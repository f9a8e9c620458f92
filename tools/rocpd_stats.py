"""Per-kernel summary (calls, total, average, min, max) of a rocprofv3 `*_results.db` (rocpd SQLite output of
`rocprofv3 --kernel-trace --stats`), written as the CSV layout of rocprofv3's kernel_stats: profiles/*.csv are made
with this from the database the GPU box returns."""
import csv
import sqlite3
import sys


def main(db_path, out_csv=None):
    db = sqlite3.connect(db_path)
    rows = db.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                      "from kernels group by name order by sum(end-start) desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    out = open(out_csv, "w", newline="") if out_csv else sys.stdout
    w = csv.writer(out, quoting=csv.QUOTE_NONNUMERIC)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
    for name, n, tot, avg, mn, mx in rows:
        w.writerow([name, n, tot, round(avg, 3), round(100.0 * tot / total, 3), mn, mx])
    if out_csv:
        out.close()
    return rows


if __name__ == "__main__":
    main(*sys.argv[1:3])
