#!/bin/bash
# Development aid: build K1 variants on the GPU box and time them (python tests/gpu_k1_bench.py).
set -e
cd "$(dirname "$0")/.."
F="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared"
build() { hipcc $F $2 -o /tmp/$1.so dumpvdl2_amd/csrc/vdl2hip.hip; }
build base "" &
build inl5 "-DVDL2_K1_INLINE_PHASE" &
build inl4 "-DVDL2_K1_INLINE_PHASE -DVDL2_K1_MIN_BLOCKS=4" &
wait
for v in base inl5 inl4; do
  for C in 8; do VDL2HIP_LIB=/tmp/$v.so python tests/gpu_k1_bench.py $C 16 3 | cut -c1-110; done
done
