#!/bin/bash
# round 2, last GPU call: smoke(), the N = 2 rehearsal of the driver's multi-GPU command (both ranks on GPU 0 over gloo), the driver's bench line
R="$(cd "$(dirname "$0")/.." && pwd)"
cd "$R"; mkdir -p gpurun_out
O=gpurun_out/r02final
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
VDL2_BENCH_REHEARSAL=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 2 --steps 5 --warmup 2 --duration 4 > $O.rehearsal2.json 2> $O.rehearsal2.err; echo "rehearsal rc=$?"; tail -c 400 $O.rehearsal2.json
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O.bench_default.json 2> $O.bench_default.err; echo "bench rc=$?"
python - <<'PY'
import json
b=json.loads(open('gpurun_out/r02final.bench_default.json').read().strip().splitlines()[-1])
print(b['value'], b['ms_per_step'], b['value_hbm_resident'], b['ms_per_step_hbm_resident'], b['roofline']['avg_launch_ms'], b['roofline']['frac'], b['config'].get('verified'))
for s in b['config'].get('secondary', []): print('   ', s['workload'], s.get('value'), s.get('ms_per_step'), s['value_hbm_resident'], s['ms_per_step_hbm_resident'], s['k_chanfir_ms'])
PY
