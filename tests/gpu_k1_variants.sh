#!/bin/bash
# Development aid: build K1 variants on the GPU box and time them (python tests/gpu_k1_bench.py).
set -e
cd "$(dirname "$0")/.."
F="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared"
build() { hipcc $F $2 -o /tmp/$1.so dumpvdl2_amd/csrc/vdl2hip.hip; }
build base "" &
build r1 "-DVDL2_K1_RUN=1" &
build r1u10 "-DVDL2_K1_RUN=1 -DVDL2_K1_UNROLL=10" &
build u4 "-DVDL2_K1_UNROLL=4" &
wait
for v in base r1 r1u10 u4; do
  for C in 8 64 256; do VDL2HIP_LIB=/tmp/$v.so python tests/gpu_k1_bench.py $C 16 3 | cut -c1-130; done
done
