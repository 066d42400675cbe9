#!/bin/bash
# round 2, GPU call Z: k_chanseq (LDS input staging, 32-byte output groups) - K1 alone both ways, counters, suite, bench
R="$(cd "$(dirname "$0")/.." && pwd)"
cd "$R"; mkdir -p gpurun_out
O=gpurun_out/r02z
for k in seq tile; do for C in 64 256; do VDL2HIP_K1=$k timeout 300 python tests/gpu_k1_bench.py $C 16 3 2>&1 | grep -v amdgpu.ids | sed "s/^default/$k/" | cut -c1-200; done; done | tee $O.k1.txt
timeout 900 python -m pytest tests -x -q -m gpu > $O.pytest.txt 2>&1; tail -4 $O.pytest.txt
timeout 600 python bench.py --no-secondary --no-cpu-baseline > $O.bench1.json 2> $O.bench1.err; python - $O.bench1.json <<'P'
import json,sys
j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
st=j['config'].get('stage_ms_per_step',{})
print(f"host {j['value']:.1f} MS/s {j['ms_per_step']:.4f} ms | hbm {j.get('ms_per_step_hbm_resident')} ms | K1 {j['roofline'].get('avg_launch_ms'):.4f} " + " ".join(f"{k[:-3]} {v}" for k,v in st.items() if k!='chanfir_ms'), j['config'].get('verified'))
P
VDL2HIP_K1=seq KFILTER=chanseq timeout 300 bash tests/gpu_k1_pmc.sh 256 > $O.sq_seq.txt 2>&1; cat $O.sq_seq.txt
cd /tmp && export TMPDIR=/tmp
for CN in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmcz_$CN; VDL2HIP_K1=seq timeout 200 rocprofv3 --kernel-trace --pmc $CN -d /tmp/pmcz_$CN -o p -- python $R/tests/gpu_k1_bench.py 256 16 2 > /tmp/pmcz_$CN.log 2>&1
  python - $CN <<'P'
import sqlite3, sys, glob
cn=sys.argv[1]
dbs=glob.glob(f"/tmp/pmcz_{cn}/**/*.db", recursive=True)
cur=sqlite3.connect(dbs[0]).cursor()
for name, tot, n in cur.execute(f"select kernel_name, sum(value), count(distinct dispatch_id) from counters_collection where counter_name='{cn}' group by kernel_name order by 2 desc"):
    if 'chan' in name: print(f"{cn} {name[:50]} n={n} avg_MB={tot/n/1024:.1f}")
P
done
