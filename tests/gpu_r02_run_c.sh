#!/bin/bash
# round 2, GPU call C: K1 variants (software-pipelined gathers, swizzled LUT), each checked for parity, then the bench
R="$(cd "$(dirname "$0")/.." && pwd)"
cd "$R"; mkdir -p gpurun_out
O=gpurun_out/r02c
CHANS="8 64 256" timeout 1500 bash tests/gpu_k1_variants.sh "base:-DVDL2_K1_PREFETCH=0" "pf2:-DVDL2_K1_PREFETCH=2" "pf2swz:-DVDL2_K1_PREFETCH=2 -DVDL2_K1_SWZ=1" \
   "swz:-DVDL2_K1_PREFETCH=0 -DVDL2_K1_SWZ=1" "pf2c3:-DVDL2_K1_PREFETCH=2 -DVDL2_K1_MIN_BLOCKS_CR4=3" "pf2swzc3:-DVDL2_K1_PREFETCH=2 -DVDL2_K1_SWZ=1 -DVDL2_K1_MIN_BLOCKS_CR4=3" \
   "pf2swzb5:-DVDL2_K1_PREFETCH=2 -DVDL2_K1_SWZ=1 -DVDL2_K1_MIN_BLOCKS=5" > $O.k1var.txt 2>&1
for v in pf2 pf2swz pf2swzc3 pf2swzb5; do
  for C in 64 256; do VDL2HIP_CR=2 VDL2HIP_LIB=/tmp/k1_$v.so timeout 300 python tests/gpu_k1_bench.py $C 16 3 | sed "s|^/tmp/k1_||" | cut -c1-230 >> $O.k1var.txt; done
done
for v in pf2 pf2swz; do
  echo "== parity $v" >> $O.k1var.txt
  VDL2HIP_LIB=/tmp/k1_$v.so timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "golden_cases_single_feed or decimated_stream or chunking or other_oversampling or uint8" 2>&1 | tail -3 >> $O.k1var.txt
done
timeout 900 python -m pytest tests -m gpu -x -q > $O.pytest.txt 2>&1; echo "pytest rc=$?" >> $O.pytest.txt
tail -3 $O.pytest.txt
timeout 900 python bench.py --no-secondary > $O.bench.json 2> $O.bench.err; echo "bench rc=$?"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof256 -o r -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-verify --no-secondary > $R/$O.prof256.log 2>&1
DB=$(find /tmp/prof256 -name "*.db" | head -1); [ -n "$DB" ] && python $R/profiles/summarize_rocpd.py $DB > $R/$O.kernel_trace_bench_256ch.txt
