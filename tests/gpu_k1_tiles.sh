#!/bin/bash
cd "$(dirname "$0")/.."
for T in 1 2 4 8 16; do for C in 8 256; do VDL2HIP_K1_TILES=$T python tests/gpu_k1_bench.py $C 16 3 | cut -c1-150 | sed "s/^/tiles=$T /"; done; done
