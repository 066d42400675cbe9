#!/bin/bash
# arbitrary SQ counter groups for every kernel of the pipeline: dev/gpu_pmc_groups.sh <workload> "<group 1>" "<group 2>" ...
R="$(cd "$(dirname "$0")/.." && pwd)"
W=$1; shift
python $R/dev/gpu_variants.py --synth-only --workloads $W > /dev/null 2>&1
cd /tmp && export TMPDIR=/tmp
i=0
for G in "$@"; do
	i=$((i+1)); rm -rf /tmp/pmcg$i
	timeout 300 rocprofv3 --kernel-trace --pmc $G -d /tmp/pmcg$i -o p -- python $R/dev/gpu_variants.py --child --workload $W --parts all --steps 2 --repeats 1 > /tmp/pmcg$i.log 2>&1
	python - "$i" <<'PY'
import sqlite3, sys, glob
i = sys.argv[1]
dbs = glob.glob(f"/tmp/pmcg{i}/**/*.db", recursive=True)
if not dbs:
    print("no db for group", i); print(open(f"/tmp/pmcg{i}.log").read()[-800:]); sys.exit(0)
cur = sqlite3.connect(dbs[0]).cursor()
try:
    q = "select kernel_name, counter_name, sum(value), count(distinct dispatch_id) from counters_collection group by kernel_name, counter_name order by 1, 2"
    for name, cn, tot, n in cur.execute(q):
        if "vdl2" not in name: continue
        print(f"{name[:44]:44s} {cn:22s} n={n:3d} per_dispatch={tot / n:14.5g}")
except Exception as e:
    print("group", i, "failed:", e); print(open(f"/tmp/pmcg{i}.log").read()[-800:])
PY
done
