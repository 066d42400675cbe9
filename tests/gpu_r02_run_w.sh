#!/bin/bash
# round 2, GPU call W: exact tier of the sync metric on the walk stream (beside the next feed's channeliser) vs on the front stream
R="$(cd "$(dirname "$0")/.." && pwd)"
cd "$R"; mkdir -p gpurun_out
O=gpurun_out/r02w
: > $O.txt
for rep in 1 2; do for xs in front walk; do for wl in config4 config3 config2; do
  VDL2HIP_SYNC_ON=$xs timeout 600 python bench.py --no-secondary --no-cpu-baseline --no-verify --workload $wl > $O.tmp.json 2> $O.err
  python - "$xs" "$wl" >> $O.txt <<'P'
import json,sys
j=json.loads(open('gpurun_out/r02w.tmp.json').read().strip().splitlines()[-1])
st=j['config'].get('stage_ms_per_step',{})
print(f"exact_on={sys.argv[1]:5s} {sys.argv[2]} host {j['value']:.1f} MS/s {j['ms_per_step']:.4f} ms | hbm {j.get('value_hbm_resident')} {j.get('ms_per_step_hbm_resident')} ms | K1 {j['roofline'].get('avg_launch_ms'):.4f} K3 {st.get('sync_ms')}")
P
done; done; done
cat $O.txt
VDL2HIP_SYNC_ON=walk timeout 900 python -m pytest tests -x -q -m gpu > $O.pytest_walk.txt 2>&1; tail -3 $O.pytest_walk.txt
