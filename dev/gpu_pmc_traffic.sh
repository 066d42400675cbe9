#!/bin/bash
# HBM traffic per kernel of the default bench (config4, 256 channels): rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes
# (never combined with other trace domains), summarised per kernel.  Output: the text kept under profiles/ as rNN_pmc_hbm_traffic_*.txt
R="$(cd "$(dirname "$0")/.." && pwd)"
W=${1:-config4}
cd /tmp && export TMPDIR=/tmp
python $R/dev/gpu_variants.py --synth-only --workloads $W > /dev/null 2>&1
echo "# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on \`dev/gpu_variants.py --child --workload $W --parts all --steps 3 --repeats 1\`: the 16 s block (33.6 M samples) resident in HBM, every channeliser launch a whole block (no cold-start pieces)"
echo "# units: KB per dispatch, averaged over dispatches.  FETCH_SIZE under-reports wide coalesced reads by 2x on gfx950 (MI355X_MICROARCH.md, HBM section)"
for CN in FETCH_SIZE WRITE_SIZE; do
	rm -rf /tmp/pmc_$CN
	rocprofv3 --kernel-trace --pmc $CN -d /tmp/pmc_$CN -o p -- python $R/dev/gpu_variants.py --child --workload $W --parts all --steps 3 --repeats 1 > /tmp/pmc_$CN.log 2>&1
	python - "$CN" <<'PY'
import sqlite3, sys, glob
cn = sys.argv[1]
dbs = glob.glob(f"/tmp/pmc_{cn}/**/*.db", recursive=True)
cur = sqlite3.connect(dbs[0]).cursor()
cols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
kcol = "kernel_name" if "kernel_name" in cols else [c for c in cols if "kernel" in c.lower() or c == "name"][0]
ccol = "counter_name" if "counter_name" in cols else [c for c in cols if "counter" in c.lower() and "name" in c.lower()][0]
vcol = "value" if "value" in cols else [c for c in cols if "value" in c.lower()][0]
dcol = "dispatch_id" if "dispatch_id" in cols else None
q = f"select {kcol}, sum({vcol}), count(distinct {dcol}) from counters_collection where {ccol}='{cn}' group by {kcol} order by 2 desc" if dcol else \
    f"select {kcol}, sum({vcol}), count(*) from counters_collection where {ccol}='{cn}' group by {kcol} order by 2 desc"
for name, tot, n in cur.execute(q):
    if "vdl2" not in name: continue
    print(f"{cn:11s} {name[:70]:70s} n={n:3d} avg_KB={tot / n:14.1f}  avg_MB={tot / n / 1024:10.1f}")
PY
done
