#!/bin/bash
# round 2, GPU call H: parity suite on the final kernels; rehearsal of the driver's N=8 command on one GPU (gloo transport)
R="$(cd "$(dirname "$0")/.." && pwd)"
cd "$R"; mkdir -p gpurun_out
O=gpurun_out/r02h
timeout 900 python -m pytest tests -m gpu -x -q > $O.pytest.txt 2>&1; echo "pytest rc=$?" >> $O.pytest.txt
tail -3 $O.pytest.txt
timeout 300 python tests/gpu_stage_times.py config4 16 3 2>&1 | grep -v amdgpu.ids > $O.stage.txt; cat $O.stage.txt
VDL2_BENCH_REHEARSAL=1 timeout 1200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29513 bench.py --gpus 8 --steps 3 --warmup 2 > $O.rehearsal8.json 2> $O.rehearsal8.err; echo "rehearsal8 rc=$?"; grep -v "amdgpu.ids\|socket.cpp\|OMP_NUM\|\*\*\*\*" $O.rehearsal8.err | tail -20
