#!/bin/bash
# dev/build_ref.sh [<commit>] - build the library of an earlier commit (default: the last round's head, b8f5d57 = round 3) into
# dev/_ref/libvdl2hip_<tag>.so, so that dev/gpu_variants.py can run it beside the current tree ON THE SAME GPU BOX
# (`--variant r03:@dev/_ref/libvdl2hip_r03.so`): boxes differ by +-3 %, a comparison across two gpurun calls is not worth much.
# dev/_ref/ is git-ignored; the .so travels with the gpurun snapshot.
R="$(cd "$(dirname "$0")/.." && pwd)"
C=${1:-b8f5d57}; TAG=${2:-r03}
T=$(mktemp -d); mkdir -p "$R/dev/_ref"
( cd "$R" && git archive "$C" dumpvdl2_amd/csrc include | tar -x -C "$T" ) || exit 1
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -fPIC -shared -o "$R/dev/_ref/libvdl2hip_$TAG.so" "$T/dumpvdl2_amd/csrc/vdl2hip.hip" && echo "built dev/_ref/libvdl2hip_$TAG.so from $C"
rm -rf "$T"
