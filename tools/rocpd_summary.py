"""Dump the per-kernel statistics of a rocprofv3 (rocpd SQLite) result into a CSV for profiles/.

    python tools/rocpd_summary.py gpurun_out/prof_x/x_results.db profiles/r01_x_kernel_stats.csv
"""
import csv
import sqlite3
import sys


def main(db_path, out_path):
    db = sqlite3.connect(db_path)
    cur = db.cursor()
    rows = list(cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    with open(out_path, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationUs", "AverageUs", "Percentage"])
        for name, calls, tot, avg, pct in rows:
            w.writerow([name, calls, "%.3f" % (tot / 1.0), "%.3f" % avg, "%.4f" % pct])
    print("wrote", out_path, len(rows), "kernels")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
