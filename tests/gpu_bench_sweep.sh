#!/bin/bash
# Development aid: bench.py over pipeline sub-block sizes (VDL2HIP_SUB_SAMPLES).
cd "$(dirname "$0")/.."
for S in 100000000 16800000 8400000 4200000 2100000; do
  VDL2HIP_SUB_SAMPLES=$S python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-verify 2>/dev/null > /tmp/b.json
  python - "$S" <<'PY'
import json, sys
d = json.load(open("/tmp/b.json"))
print("sub", sys.argv[1], d["value"], d["ms_per_step"], d["config"]["stage_ms_per_step"], d["roofline"]["frac"])
PY
done
