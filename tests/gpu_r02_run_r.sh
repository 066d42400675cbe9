#!/bin/bash
# round 2, GPU call R: K3 on a stream of its own (beside the next feed's channeliser) - off / front priority / walk priority
R="$(cd "$(dirname "$0")/.." && pwd)"
cd "$R"; mkdir -p gpurun_out
O=gpurun_out/r02r
: > $O.txt
for ss in front own own-high front own own-high; do
  for wl in config4 config3; do
    VDL2HIP_SYNC_ON=$ss timeout 600 python bench.py --no-secondary --no-cpu-baseline --no-verify --workload $wl > $O.tmp.json 2> $O.err
    python - "$ss" "$wl" >> $O.txt <<'P'
import json,sys
j=json.loads(open('gpurun_out/r02r.tmp.json').read().strip().splitlines()[-1])
st=j['config'].get('stage_ms_per_step',{})
print(f"sync_stream={sys.argv[1]} {sys.argv[2]} host {j['value']:.1f} MS/s {j['ms_per_step']:.4f} ms | hbm {j.get('value_hbm_resident')} {j.get('ms_per_step_hbm_resident')} ms | K1 {j['roofline'].get('avg_launch_ms'):.4f} K3 {st.get('sync_ms')}")
P
  done
done
cat $O.txt
VDL2HIP_SYNC_ON=own timeout 900 python -m pytest tests -x -q -m gpu > $O.pytest_low.txt 2>&1; tail -3 $O.pytest_low.txt
