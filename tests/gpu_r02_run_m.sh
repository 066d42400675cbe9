#!/bin/bash
# round 2, GPU call M: the channeliser's wave scan on DPP (no ds_bpermute)
R="$(cd "$(dirname "$0")/.." && pwd)"
cd "$R"; mkdir -p gpurun_out
O=gpurun_out/r02m
timeout 900 python -m pytest tests -m gpu -x -q > $O.pytest.txt 2>&1; echo "pytest rc=$?" >> $O.pytest.txt
tail -5 $O.pytest.txt
for C in 8 64 256; do timeout 200 python tests/gpu_k1_bench.py $C 16 3 2>&1 | grep -v amdgpu.ids | cut -c1-230 >> $O.isolated.txt; done; cat $O.isolated.txt
timeout 900 python bench.py --no-cpu-baseline > $O.bench.json 2> $O.bench.err; echo "bench rc=$?"; tail -c 300 $O.bench.err
python - <<'PY'
import json
b=json.loads(open('gpurun_out/r02m.bench.json').read().strip().splitlines()[-1])
print(b['value'], b['ms_per_step'], b['value_hbm_resident'], b['ms_per_step_hbm_resident'], b['roofline']['avg_launch_ms'], b['config']['verified'])
for s in b['config']['secondary']: print(s['workload'], s['value_hbm_resident'], s['ms_per_step_hbm_resident'], s['k_chanfir_ms'])
PY
