#!/bin/bash
# dev/gpu_k1_nolut.sh - build the table-free channeliser variant next to the in-tree library (both travel to the GPU box with the
# snapshot) and, on a GPU box, run the A/B of dev/gpu_k1_nolut.py:   bash dev/gpu_k1_nolut.sh build   (here)
#                                                                     gpurun -- 'bash dev/gpu_k1_nolut.sh run'
R="$(cd "$(dirname "$0")/.." && pwd)"; cd "$R"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -fPIC -shared"
case "$1" in
build) python -c "from dumpvdl2_amd import build; build.build()" && hipcc $FLAGS -DVDL2_K1_NOLUT -o dumpvdl2_amd/libvdl2hip_nolut.so dumpvdl2_amd/csrc/vdl2hip.hip \
	&& hipcc $FLAGS -DVDL2_K1_NOLUT -DVDL2_K1_UNROLL=4 -o dumpvdl2_amd/libvdl2hip_nolut_u4.so dumpvdl2_amd/csrc/vdl2hip.hip \
	&& hipcc $FLAGS -DVDL2_K1_NOLUT -DVDL2_K1_UNROLL=10 -o dumpvdl2_amd/libvdl2hip_nolut_u10.so dumpvdl2_amd/csrc/vdl2hip.hip && ls -la dumpvdl2_amd/*.so ;;
run) mkdir -p gpurun_out; timeout ${2:-280} python dev/gpu_k1_nolut.py dumpvdl2_amd/libvdl2hip.so dumpvdl2_amd/libvdl2hip_nolut.so dumpvdl2_amd/libvdl2hip_nolut_u4.so dumpvdl2_amd/libvdl2hip_nolut_u10.so 2>&1 | grep -v amdgpu.ids | tee gpurun_out/k1_nolut.txt ;;
*) echo "usage: $0 build | run [timeout_s]" ;;
esac
