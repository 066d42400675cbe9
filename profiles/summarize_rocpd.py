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
    # A kernel launched with several grid sizes is several different amounts of work under one name (k_chanfir: a whole block, or - the
    # first block of a timed region, copied and channelised in pieces - a quarter of one): its average over all launches is the
    # average of neither.  Listed per grid size, so that the whole-block launches can be compared with bench.py's avg_launch_ms.
    multi = [r[0] for r in cur.execute("select name from kernels group by name having count(distinct grid_x) > 1 and name like '%vdl2::%'")]
    for name in multi:
        print(f"# {name[:70]} by grid size:", file=out)
        for gx, n, tot, avg in cur.execute("select grid_x, count(*), sum(duration), avg(duration) from kernels where name=? group by grid_x order by sum(duration) desc limit 6", (name,)):
            print(f"#   grid {gx:>10d}: {n:5d} launches, total {tot / 1e3:12.1f} us, avg {avg / 1e3:10.2f} us", file=out)


if __name__ == "__main__":
    main(sys.argv[1])
