#!/bin/bash
# round 2, GPU call X: numbers and profiles of the tree as committed (suite, driver's bench line, config5, kernel traces, HBM traffic, SQ counters of K3a)
R="$(cd "$(dirname "$0")/.." && pwd)"
cd "$R"; mkdir -p gpurun_out
O=gpurun_out/r02x
timeout 900 python -m pytest tests -m gpu -x -q > $O.pytest.txt 2>&1; echo "pytest rc=$?" >> $O.pytest.txt
tail -3 $O.pytest.txt
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O.bench_default.json 2> $O.bench_default.err; echo "bench rc=$?"; tail -c 300 $O.bench_default.err
timeout 600 python bench.py --workload config5 --no-secondary --no-cpu-baseline > $O.bench_config5.json 2> $O.bench_config5.err; echo "config5 rc=$?"
python - <<'PY'
import json
for f in ('gpurun_out/r02x.bench_default.json', 'gpurun_out/r02x.bench_config5.json'):
    b=json.loads(open(f).read().strip().splitlines()[-1])
    print(f, b['value'], b['ms_per_step'], b['value_hbm_resident'], b['ms_per_step_hbm_resident'], b['roofline']['avg_launch_ms'], b['roofline']['frac'], b['config'].get('verified'))
    for s in b['config'].get('secondary', []): print('   ', s['workload'], s.get('value'), s.get('ms_per_step'), s['value_hbm_resident'], s['ms_per_step_hbm_resident'], s['k_chanfir_ms'])
    if 'cpu_baseline' in b: print('   cpu', b['cpu_baseline']['value'], b['cpu_baseline'].get('fast_math'))
PY
cd /tmp && export TMPDIR=/tmp
for W in config4 config2; do
  rm -rf /tmp/prof_$W
  timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_$W -o r -- python $R/bench.py --workload $W --steps 10 --warmup 2 --no-cpu-baseline --no-verify --no-secondary > /dev/null 2>&1
  DB=$(find /tmp/prof_$W -name "*.db" | head -1); [ -n "$DB" ] && python $R/profiles/summarize_rocpd.py $DB > $R/$O.kernel_trace_bench_$W.txt
done
cd $R
for C in 8 64 256; do timeout 200 python tests/gpu_k1_bench.py $C 16 3 2>&1 | grep -v amdgpu.ids | cut -c1-230 >> $O.isolated.txt; done; cat $O.isolated.txt
cd /tmp
rm -rf /tmp/prof_iso; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_iso -o r -- python $R/tests/gpu_stage_times.py config4 16 3 > /dev/null 2>&1
DB=$(find /tmp/prof_iso -name "*.db" | head -1); [ -n "$DB" ] && python $R/profiles/summarize_rocpd.py $DB > $R/$O.kernel_trace_isolated_config4.txt
cd $R
timeout 400 bash tests/gpu_pmc_traffic.sh config4 > $O.pmc_hbm_traffic_config4.txt 2>&1
KFILTER=sync_screen timeout 400 bash tests/gpu_k1_pmc.sh 256 > $O.sq_k3a_256ch.txt 2>&1
head -20 $O.kernel_trace_bench_config4.txt | cut -c1-150
