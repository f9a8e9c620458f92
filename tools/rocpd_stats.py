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
    # optional third argument --by-grid: see by_grid() below (defined later in the file, so dispatched at the end)


def by_grid(db_path, out_csv):
    """The same summary per (kernel, grid size): the fused AMG kernels serve several levels, which only the launch geometry tells
    apart -- input of tools/scaling_model.py."""
    db = sqlite3.connect(db_path)
    cols = [r[1] for r in db.execute("pragma table_info(kernels)").fetchall()]
    gx = next((c for c in ("grid_x", "grid_size_x", "grid_size") if c in cols), None)
    wx = next((c for c in ("workgroup_x", "workgroup_size_x", "workgroup_size") if c in cols), None)
    if gx is None:
        raise SystemExit("kernels view has no grid column: %s" % cols)
    q = "select name, %s, %s, count(*), sum(end-start), avg(end-start), min(end-start) from kernels group by name, %s order by sum(end-start) desc" % (
        gx, wx or "0", gx)
    with open(out_csv, "w", newline="") as f:
        w = csv.writer(f, quoting=csv.QUOTE_NONNUMERIC)
        w.writerow(["Name", "Grid", "Workgroup", "Calls", "TotalDurationNs", "AverageNs", "MinNs"])
        for r in db.execute(q).fetchall():
            w.writerow([r[0], r[1], r[2], r[3], r[4], round(r[5], 3), r[6]])


if __name__ == "__main__" and len(sys.argv) > 3 and sys.argv[3] == "--by-grid":
    by_grid(sys.argv[1], sys.argv[2].replace(".csv", "_by_grid.csv"))
