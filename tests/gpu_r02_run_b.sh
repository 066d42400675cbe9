#!/bin/bash
# round 2, GPU call B: after the phase refactor (no stored phase stream, K3 split in screening + exact kernels)
R="$(cd "$(dirname "$0")/.." && pwd)"
cd "$R"; mkdir -p gpurun_out
O=gpurun_out/r02b
timeout 900 python -m pytest tests -m gpu -x -q > $O.pytest.txt 2>&1; echo "pytest rc=$?" >> $O.pytest.txt
tail -5 $O.pytest.txt
timeout 900 python bench.py > $O.bench.json 2> $O.bench.err; echo "bench rc=$?"
tail -c 1500 $O.bench.err
timeout 900 bash tests/gpu_k1_variants.sh > $O.k1var.txt 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof256 -o r -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-verify --no-secondary > $R/$O.prof256.log 2>&1
DB=$(find /tmp/prof256 -name "*.db" | head -1); [ -n "$DB" ] && python $R/profiles/summarize_rocpd.py $DB > $R/$O.kernel_trace_bench_256ch.txt
