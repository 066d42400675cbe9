"""Kernel timeline out of a rocprofv3 rocpd database (--kernel-trace): which kernel ran when, on which queue - to see what a feed's way
through the device really looks like (what waits for what).  usage: python dev/gpu_timeline.py <results.db> [t0_ms] [t1_ms]
Prints the schema of the `kernels` view once, then one line per dispatch in [t0, t1] of the LAST 60 % of the trace (steady state)."""
import sqlite3
import sys


def main(path, w0=None, w1=None):
    cur = sqlite3.connect(path).cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    print("# kernels columns:", cols)
    want = [c for c in ("name", "start", "end", "duration", "queue_id", "stream_id", "grid_x", "dispatch_id", "tid") if c in cols]
    rows = list(cur.execute(f"select {','.join(want)} from kernels order by start"))
    if not rows:
        print("no kernels"); return
    ix = {c: i for i, c in enumerate(want)}
    t_first, t_last = rows[0][ix["start"]], rows[-1][ix["end"]]
    print(f"# {len(rows)} dispatches over {(t_last - t_first) / 1e6:.2f} ms")
    base = t_first
    lo = (t_last - t_first) * 0.55 if w0 is None else w0 * 1e6
    hi = lo + 12e6 if w1 is None else w1 * 1e6
    for r in rows:
        s, e = r[ix["start"]] - base, r[ix["end"]] - base
        if e < lo or s > hi:
            continue
        nm = r[ix["name"]].replace("vdl2::", "").split("(")[0][:34]
        q = r[ix["queue_id"]] if "queue_id" in ix else -1
        st = r[ix["stream_id"]] if "stream_id" in ix else -1
        print(f"{s / 1e6:9.3f} ms  +{(e - s) / 1e3:8.1f} us  q{q} s{st}  grid {r[ix['grid_x']] if 'grid_x' in ix else '':>8}  {nm}")


if __name__ == "__main__":
    main(sys.argv[1], *(float(a) for a in sys.argv[2:4]))
