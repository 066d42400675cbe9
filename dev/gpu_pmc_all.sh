#!/bin/bash
# SQ counters of EVERY kernel of the pipeline (one rocprofv3 pass per counter group; counter collection serialises the dispatches,
# so these are per-kernel totals, not a picture of the overlap): dev/gpu_pmc_all.sh <workload> [parts] -> per kernel and counter: sum / dispatches
R="$(cd "$(dirname "$0")/.." && pwd)"
W=${1:-config4}; PARTS=${2:-all}
python $R/dev/gpu_variants.py --synth-only --workloads $W > /dev/null 2>&1
cd /tmp && export TMPDIR=/tmp
i=0
for G in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_WAIT_INST_ANY"; do
	i=$((i+1)); rm -rf /tmp/pmca$i
	timeout 300 rocprofv3 --kernel-trace --pmc $G -d /tmp/pmca$i -o p -- python $R/dev/gpu_variants.py --child --workload $W --parts $PARTS --steps 2 --repeats 1 > /tmp/pmca$i.log 2>&1
	python - "$i" <<'PY'
import sqlite3, sys, glob
i = sys.argv[1]
dbs = glob.glob(f"/tmp/pmca{i}/**/*.db", recursive=True)
if not dbs:
    print("no db for group", i); print(open(f"/tmp/pmca{i}.log").read()[-1500:]); sys.exit(0)
cur = sqlite3.connect(dbs[0]).cursor()
cols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
kcol = "kernel_name" if "kernel_name" in cols else [c for c in cols if "kernel" in c.lower() or c == "name"][0]
ccol = "counter_name" if "counter_name" in cols else [c for c in cols if "counter" in c.lower() and "name" in c.lower()][0]
vcol = "value" if "value" in cols else [c for c in cols if "value" in c.lower()][0]
dcol = "dispatch_id" if "dispatch_id" in cols else None
q = f"select {kcol}, {ccol}, sum({vcol}), count(distinct {dcol}) from counters_collection group by {kcol}, {ccol} order by 1, 2"
for name, cn, tot, n in cur.execute(q):
    if "vdl2" not in name: continue
    print(f"{name[:44]:44s} {cn:22s} n={n:3d} per_dispatch={tot / n:14.5g}")
PY
done
