#!/bin/bash
# Development aid: build K1 variants on the GPU box and time them (python dev/gpu_k1_bench.py).
# usage: dev/gpu_k1_variants.sh "<name>:<flags>" ...   (default set below); channel counts from $CHANS (default "8 64 256")
cd "$(dirname "$0")/.."
F="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -fPIC -shared"
if [ $# -eq 0 ]; then set -- "base:" "mb5:-DVDL2_K1_MIN_BLOCKS=5" "cr4mb5:-DVDL2_K1_MIN_BLOCKS_CR4=5" "cr4mb3:-DVDL2_K1_MIN_BLOCKS_CR4=3" "unr4:-DVDL2_K1_UNROLL=4" "unr10:-DVDL2_K1_UNROLL=10"; fi
for v in "$@"; do ( hipcc $F ${v#*:} -o /tmp/k1_${v%%:*}.so dumpvdl2_amd/csrc/vdl2hip.hip 2>/dev/null || echo "build of $v failed" ) & done
wait
for v in "$@"; do
  for C in ${CHANS:-8 64 256}; do VDL2HIP_LIB=/tmp/k1_${v%%:*}.so timeout 300 python dev/gpu_k1_bench.py $C 16 3 | sed "s|^/tmp/k1_||" | cut -c1-230; done
done
