#!/bin/bash
# round 2, GPU call Q: screening tier on shared tap differences (turns), with and without the SLP vectoriser; suite; bench
R="$(cd "$(dirname "$0")/.." && pwd)"
cd "$R"; mkdir -p gpurun_out
O=gpurun_out/r02q
CHANS="8 64 256" timeout 900 bash tests/gpu_k1_variants.sh "base:" "noslp:-fno-slp-vectorize" 2>&1 | grep -v amdgpu.ids > $O.k1var.txt; cut -c1-200 $O.k1var.txt
timeout 900 python -m pytest tests -x -q -m gpu > $O.pytest.txt 2>&1; tail -3 $O.pytest.txt
timeout 600 python bench.py --no-secondary > $O.bench.json 2> $O.bench.err; python - <<'P'
import json
j=json.loads(open('gpurun_out/r02q.bench.json').read().strip().splitlines()[-1])
print(j['value'], j['ms_per_step'], j.get('value_hbm_resident'), j.get('ms_per_step_hbm_resident'), j['roofline'].get('avg_launch_ms'))
P
VDL2HIP_LIB=/tmp/k1_noslp.so timeout 600 python bench.py --no-secondary > $O.bench_noslp.json 2> $O.bench_noslp.err; tail -c 600 $O.bench_noslp.json
