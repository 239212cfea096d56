"""Per (kernel, grid) duration statistics from a rocprofv3 rocpd database (finer than top_kernels: separates the
per-octave launches of one kernel).   python tools/rocpd_bygrid.py results.db out.csv"""
import collections
import csv
import sqlite3
import sys


def main(db_path, out_path):
    db = sqlite3.connect(db_path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    if not cols:
        print("no `kernels` view; objects:", [r[0] for r in cur.execute("select name from sqlite_master")])
        return
    pick = lambda *names: next((c for c in names if c in cols), None)
    name, gx, gy, gz = pick("name", "kernel_name"), pick("grid_x", "grid_size_x"), pick("grid_y", "grid_size_y"), pick("grid_z", "grid_size_z")
    st, en, du = pick("start"), pick("end"), pick("duration")
    if not (name and gx and (du or (st and en))):
        print("unexpected columns:", cols)
        return
    expr = du if du else "(%s - %s)" % (en, st)
    agg = collections.defaultdict(lambda: [0, 0.0])
    for n, x, y, z, d in cur.execute("select %s, %s, %s, %s, %s from kernels" % (name, gx, gy or 1, gz or 1, expr)):
        a = agg[(n.split("(")[0], x, y, z)]
        a[0] += 1; a[1] += d
    with open(out_path, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "GridX", "GridY", "GridZ", "Calls", "TotalUs", "AverageUs"])
        for (n, x, y, z), (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            w.writerow([n, x, y, z, c, "%.1f" % (t / 1e3), "%.2f" % (t / 1e3 / c)])
    print("wrote", out_path)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
