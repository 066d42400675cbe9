#!/bin/bash
# round 2: do more hardware queues (GPU_MAX_HW_QUEUES, read by the HIP runtime at start-up) let the streams overlap better, and
# does the sync tier then hide beside the next feed's channeliser?
R="$(cd "$(dirname "$0")/.." && pwd)"
cd "$R"; mkdir -p gpurun_out
O=gpurun_out/r02hwq
: > $O.txt
for q in default 8 16; do for so in front own own-high; do
  if [ $q = default ]; then unset GPU_MAX_HW_QUEUES; else export GPU_MAX_HW_QUEUES=$q; fi
  VDL2HIP_SYNC_ON=$so timeout 300 python bench.py --no-secondary --no-cpu-baseline --no-verify > $O.tmp.json 2> $O.err
  python - "$q" "$so" >> $O.txt <<'P'
import json,sys
try:
    j=json.loads(open('gpurun_out/r02hwq.tmp.json').read().strip().splitlines()[-1])
    st=j['config'].get('stage_ms_per_step',{})
    print(f"hw_queues={sys.argv[1]:7s} sync_on={sys.argv[2]:8s} host {j['value']:.1f} MS/s {j['ms_per_step']:.4f} ms | hbm {j.get('ms_per_step_hbm_resident')} ms | K1 {j['roofline'].get('avg_launch_ms'):.4f} K3 {st.get('sync_ms')} walk {st.get('walk_ms')} nf {st.get('nf_ms')} burst {st.get('burst_ms')}")
except Exception as e: print(sys.argv[1], sys.argv[2], 'failed', e)
P
done; done
cat $O.txt
