#!/bin/bash
# round 2, GPU calls T, U: back-end kernels regrouped into multi-wave workgroups (noise-floor replay, walkers, burst decoder): suite + bench + stage times alone
R="$(cd "$(dirname "$0")/.." && pwd)"
cd "$R"; mkdir -p gpurun_out
O=gpurun_out/r02t
timeout 900 python -m pytest tests -x -q -m gpu > $O.pytest.txt 2>&1; tail -3 $O.pytest.txt
for i in 1 2; do
timeout 600 python bench.py --no-secondary --no-cpu-baseline --no-verify > $O.bench$i.json 2> $O.bench$i.err; python - $O.bench$i.json <<'P'
import json,sys
j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
st=j['config'].get('stage_ms_per_step',{})
print(f"host {j['value']:.1f} MS/s {j['ms_per_step']:.4f} ms | hbm {j.get('ms_per_step_hbm_resident')} ms | K1 {j['roofline'].get('avg_launch_ms'):.4f} " + " ".join(f"{k[:-3]} {v}" for k,v in st.items() if k!='chanfir_ms'))
P
done
for C in 8 64 256; do timeout 300 python tests/gpu_k1_bench.py $C 16 3 | cut -c1-230; done
