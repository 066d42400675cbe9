#!/bin/bash
# Development aid: SQ/LDS counters of one kernel (default the channeliser K1; KFILTER=sync_screen for K3a ...) on C channels of
# noise (argument 1, default 8).  One rocprofv3 pass per counter group.
R="$(cd "$(dirname "$0")/.." && pwd)"
cd /tmp && export TMPDIR=/tmp
i=0
for G in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES" "SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_LDS" "SQ_WAIT_INST_ANY SQ_WAVES GRBM_GUI_ACTIVE"; do
	i=$((i+1))
	rocprofv3 --kernel-trace --pmc $G -d /tmp/pmc$i -o p -- python $R/dev/gpu_k1_bench.py ${1:-8} 16 2 > /tmp/pmc$i.log 2>&1
	python - "$i" "${KFILTER:-chanfir}" <<'PY'
import sqlite3, sys, glob
i = sys.argv[1]; kf = sys.argv[2]
dbs = glob.glob(f"/tmp/pmc{i}/**/*.db", recursive=True)
if not dbs:
    print("no db for group", i); print(open(f"/tmp/pmc{i}.log").read()[-1500:]); sys.exit(0)
cur = sqlite3.connect(dbs[0]).cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
v = [t for t in tabs if "counters_collection" in t or "pmc" in t.lower()]
cols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
kcol = "kernel_name" if "kernel_name" in cols else [c for c in cols if "kernel" in c.lower() or c == "name"][0]
ccol = "counter_name" if "counter_name" in cols else [c for c in cols if "counter" in c.lower() and "name" in c.lower()][0]
vcol = "value" if "value" in cols else [c for c in cols if "value" in c.lower()][0]
rows = list(cur.execute(f"select {kcol}, {ccol}, sum({vcol}), count(*) from counters_collection where {kcol} like '%{kf}%' group by {kcol}, {ccol}"))
if not rows: print("columns:", cols)
for r in rows: print(f"{r[0][:40]:40s} {r[1]:26s} sum={r[2]:.5g} n={r[3]} avg={r[2]/r[3]:.5g}")
PY
done
