#!/bin/bash
# round 2, GPU call K: which back-end streams should outrank the channeliser?
R="$(cd "$(dirname "$0")/.." && pwd)"
cd "$R"; mkdir -p gpurun_out
O=gpurun_out/r02k
for P in "" "nf" "nf,burst" "nf,burst,walk"; do
  for W in config4 config2; do
    VDL2HIP_LOW_PRIO="$P" timeout 300 python bench.py --workload $W --steps 20 --warmup 3 --no-secondary --no-cpu-baseline --no-verify 2>/dev/null | python -c "
import json,sys
b=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('low=[$P] $W host %.1f MS/s %.4f ms | hbm %.1f MS/s %.4f ms | K1 %.4f ms' % (b['value'], b['ms_per_step'], b['value_hbm_resident'], b['ms_per_step_hbm_resident'], b['roofline']['avg_launch_ms']))" >> $O.prio.txt
  done
done
cat $O.prio.txt
