#!/bin/bash
# round 2, GPU call N: K1 epilogue variants
R="$(cd "$(dirname "$0")/.." && pwd)"
cd "$R"; mkdir -p gpurun_out
O=gpurun_out/r02n
CHANS="8 64 256" timeout 1200 bash tests/gpu_k1_variants.sh "base:" "si:-DVDL2_K1_STAGE_INC=1" "ub:-DVDL2_K1_UNROLL_BLOCKS=1" "ubsi:-DVDL2_K1_UNROLL_BLOCKS=1 -DVDL2_K1_STAGE_INC=1" 2>&1 | grep -v amdgpu.ids > $O.k1var.txt; cut -c1-150 $O.k1var.txt
