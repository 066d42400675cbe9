#!/bin/bash
# round 2, GPU call A: parity suite, the new default bench line, issue-rate micro-benchmarks, SQ counters of K1 (before)
R="$(cd "$(dirname "$0")/.." && pwd)"
cd "$R"; mkdir -p gpurun_out
O=gpurun_out/r02a
{ nproc; grep -m1 "model name" /proc/cpuinfo; free -g | head -2; gcc --version | head -1; } > $O.host.txt 2>&1
timeout 900 python -m pytest tests -m gpu -x -q > $O.pytest.txt 2>&1; echo "pytest rc=$?" >> $O.pytest.txt
tail -5 $O.pytest.txt
timeout 600 python bench.py > $O.bench.json 2> $O.bench.err; echo "bench rc=$?"
tail -c 600 $O.bench.err
hipcc --offload-arch=gfx950 -O3 -o /tmp/ub tests/gpu_ubench_valu.hip 2>/dev/null && timeout 120 /tmp/ub > $O.ubench.txt 2>&1
for C in 256 8; do timeout 400 bash tests/gpu_k1_pmc.sh $C > $O.sq_k1_${C}ch.txt 2>&1; done
