"""Per-kernel statistics (calls, total / average / min / max ns, share) from a rocprofv3 rocpd database (`*_results.db`), in the
column layout of rocprofv3's own `--stats` CSV.  Usage: python tools/rocpd_stats.py RESULTS.db [OUT.csv]"""
import csv
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    rows = db.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels group by name "
                      "order by sum(duration) desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    out = csv.writer(open(sys.argv[2], "w", newline="") if len(sys.argv) > 2 else sys.stdout)
    out.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
    for name, calls, tot, avg, mn, mx in rows:
        out.writerow([name, calls, tot, f"{avg:.1f}", f"{100.0 * tot / total:.2f}", mn, mx])


if __name__ == "__main__":
    main()
