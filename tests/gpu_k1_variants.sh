#!/bin/bash
# Development aid: build K1 variants on the GPU box and time them (python tests/gpu_k1_bench.py).
set -e
cd "$(dirname "$0")/.."
F="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared"
build() { hipcc $F $2 -o /tmp/$1.so dumpvdl2_amd/csrc/vdl2hip.hip; }
build base "" &
build u2 "-DVDL2_K1_UNROLL=2" &
build u5 "-DVDL2_K1_UNROLL=5" &
build w4 "-DVDL2_K1_WAVES_PER_EU=4" &
build r2 "-DVDL2_K1_RUN=2" &
build r2w5 "-DVDL2_K1_RUN=2 -DVDL2_K1_WAVES_PER_EU=5" &
wait
for v in base u2 u5 w4 r2 r2w5; do
  for C in 8 64; do VDL2HIP_LIB=/tmp/$v.so python tests/gpu_k1_bench.py $C 16 4 | cut -c1-160; done
done
VDL2HIP_LIB=/tmp/base.so VDL2HIP_CR=1 python tests/gpu_k1_bench.py 8 16 4 | cut -c1-160
VDL2HIP_LIB=/tmp/base.so VDL2HIP_CR=4 python tests/gpu_k1_bench.py 8 16 4 | cut -c1-160
VDL2HIP_LIB=/tmp/base.so python tests/gpu_k1_bench.py 256 16 4 | cut -c1-160
