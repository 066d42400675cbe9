#!/bin/bash
# round 2, GPU call Z (last form): suite incl. the forced channel-per-lane channeliser, then the bench with the default kernels
R="$(cd "$(dirname "$0")/.." && pwd)"
cd "$R"; mkdir -p gpurun_out
O=gpurun_out/r02z
timeout 900 python -m pytest tests -x -q -m gpu > $O.pytest.txt 2>&1; tail -4 $O.pytest.txt
timeout 600 python bench.py --no-secondary --no-cpu-baseline > $O.bench1.json 2> $O.bench1.err; python - $O.bench1.json <<'P'
import json,sys
j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
st=j['config'].get('stage_ms_per_step',{})
print(f"host {j['value']:.1f} MS/s {j['ms_per_step']:.4f} ms | hbm {j.get('ms_per_step_hbm_resident')} ms | K1 {j['roofline'].get('avg_launch_ms'):.4f} " + " ".join(f"{k[:-3]} {v}" for k,v in st.items() if k!='chanfir_ms'), j['config'].get('verified'))
P
