#!/bin/bash
# round 2, GPU call L: final numbers with the walk-only-high stream priorities; SQ counters of the screening kernel
R="$(cd "$(dirname "$0")/.." && pwd)"
cd "$R"; mkdir -p gpurun_out
O=gpurun_out/r02o
timeout 900 python -m pytest tests -m gpu -x -q > $O.pytest.txt 2>&1; echo "pytest rc=$?" >> $O.pytest.txt
tail -3 $O.pytest.txt
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O.bench_default.json 2> $O.bench_default.err; echo "bench rc=$?"; tail -c 300 $O.bench_default.err
timeout 600 python bench.py --workload config5 --no-secondary --no-cpu-baseline > $O.bench_config5.json 2> $O.bench_config5.err; echo "config5 rc=$?"
cd /tmp && export TMPDIR=/tmp
for W in config4 config2; do
  rm -rf /tmp/prof_$W
  timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_$W -o r -- python $R/bench.py --workload $W --steps 10 --warmup 2 --no-cpu-baseline --no-verify --no-secondary > /dev/null 2>&1
  DB=$(find /tmp/prof_$W -name "*.db" | head -1); [ -n "$DB" ] && python $R/profiles/summarize_rocpd.py $DB > $R/$O.kernel_trace_bench_$W.txt
done
cd $R
for C in 8 64 256; do timeout 200 python tests/gpu_k1_bench.py $C 16 3 2>&1 | grep -v amdgpu.ids | cut -c1-230 >> $O.isolated.txt; done
cd /tmp
rm -rf /tmp/prof_iso; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_iso -o r -- python $R/tests/gpu_stage_times.py config4 16 3 > /dev/null 2>&1
DB=$(find /tmp/prof_iso -name "*.db" | head -1); [ -n "$DB" ] && python $R/profiles/summarize_rocpd.py $DB > $R/$O.kernel_trace_isolated_config4.txt
cd $R
timeout 400 bash tests/gpu_k1_pmc.sh 256 > $O.sq_k1_256ch.txt 2>&1
