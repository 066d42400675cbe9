#!/bin/bash
# round 2, GPU call J: wave-private compaction in the screening kernel (A/B), noise-floor replay from an LDS slice
R="$(cd "$(dirname "$0")/.." && pwd)"
cd "$R"; mkdir -p gpurun_out
O=gpurun_out/r02j
CHANS="8 64 256" timeout 900 bash tests/gpu_k1_variants.sh "wc:" "old:-DVDL2_K3_WAVE_COMPACT=0" 2>&1 | grep -v amdgpu.ids > $O.k3var.txt; cat $O.k3var.txt | cut -c1-200
timeout 900 python -m pytest tests -m gpu -x -q > $O.pytest.txt 2>&1; echo "pytest rc=$?" >> $O.pytest.txt
tail -3 $O.pytest.txt
timeout 300 python tests/gpu_stage_times.py config4 16 3 2>&1 | grep -v amdgpu.ids > $O.stage.txt; cat $O.stage.txt
timeout 900 python bench.py --no-secondary > $O.bench.json 2> $O.bench.err; echo "bench rc=$?"; tail -c 300 $O.bench.err
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_c4; timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_c4 -o r -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-verify --no-secondary > /dev/null 2>&1
DB=$(find /tmp/prof_c4 -name "*.db" | head -1); [ -n "$DB" ] && python $R/profiles/summarize_rocpd.py $DB > $R/$O.kernel_trace_bench_config4.txt
