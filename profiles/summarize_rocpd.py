#!/usr/bin/env python3
"""Turn a rocprofv3 rocpd database (…_results.db) into the per-kernel text summary kept under profiles/."""
import sqlite3
import sys


def main(path, out=sys.stdout):
    cur = sqlite3.connect(path).cursor()
    print(f"# rocprofv3 --kernel-trace --stats summary of {path.split('/')[-1]} (durations in microseconds)", file=out)
    print(f"{'kernel':70s} {'calls':>6s} {'total_us':>12s} {'avg_us':>10s} {'pct':>6s} {'vgpr':>5s} {'lds':>7s} {'grid':>10s}", file=out)
    rows = list(cur.execute("select name,total_calls,total_duration,average,percentage from top_kernels"))
    for name, calls, tot, avg, pct in rows:
        extra = cur.execute("select vgpr_count, lds_size, grid_x from kernels where name=? limit 1", (name,)).fetchone() or ("", "", "")
        print(f"{name[:70]:70s} {calls:6d} {tot:12.1f} {avg:10.2f} {pct:6.2f} {str(extra[0]):>5s} {str(extra[1]):>7s} {str(extra[2]):>10s}", file=out)


if __name__ == "__main__":
    main(sys.argv[1])
