#!/bin/bash
# Development aid: timeline of one steady-state step of the bench from a rocprofv3 kernel trace (start/end of every kernel)
R="$(cd "$(dirname "$0")/.." && pwd)"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/gap
rocprofv3 --kernel-trace -d /tmp/gap -o g -- python $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-verify > /tmp/gap.log 2>&1
python - <<'PY'
import sqlite3, glob
db = glob.glob("/tmp/gap/**/*.db", recursive=True)[0]
cur = sqlite3.connect(db).cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
v = "kernels" if "kernels" in tabs else [t for t in tabs if "kernel" in t.lower()][0]
cols = [r[1] for r in cur.execute(f"pragma table_info({v})")]
rows = list(cur.execute(f"select name, start, end, stream_id from {v} order by start")) if "stream_id" in cols else list(cur.execute(f"select name, start, end, 0 from {v} order by start"))
ks = [(n.split('(')[0].replace('void ', '').replace('vdl2::', ''), s, e, st) for n, s, e, st in rows if 'vdl2' in n]
# last full step: find the 3rd-from-last k_chanfir
idx = [i for i, k in enumerate(ks) if 'chanfir' in k[0]]
i0, i1 = idx[-6], idx[-4]
t0 = ks[i0][1]
print(f"two steady-state steps ({(ks[i1][1]-t0)/1e3:.1f} us): kernel start/end relative to the first k_chanfir, us")
for n, s, e, st in ks[i0:i1]:
    print(f"  {n[:24]:24s} stream {st}  {(s-t0)/1e3:8.1f} -> {(e-t0)/1e3:8.1f}  ({(e-s)/1e3:6.1f})")
PY
