#!/bin/bash
# round 2, GPU call E: where does the walk chain spend its time on config4; rehearsal of the N=2 control flow; config5 bench
R="$(cd "$(dirname "$0")/.." && pwd)"
cd "$R"; mkdir -p gpurun_out
O=gpurun_out/r02e
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -DVDL2_K5_PROF -o /tmp/prof.so dumpvdl2_amd/csrc/vdl2hip.hip 2>/dev/null &
timeout 300 python tests/gpu_stage_times.py config4 16 3 > $O.stage.txt 2>&1
VDL2HIP_SEG_MAX=1 timeout 300 python tests/gpu_stage_times.py config4 16 2 >> $O.stage.txt 2>&1
timeout 300 python tests/gpu_stage_times.py config2 16 3 >> $O.stage.txt 2>&1
wait
VDL2HIP_LIB=/tmp/prof.so timeout 300 python tests/gpu_stage_times.py config4 16 2 >> $O.stage.txt 2>&1
cat $O.stage.txt | grep -v amdgpu.ids
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_iso; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_iso -o r -- python $R/tests/gpu_stage_times.py config4 16 3 > /dev/null 2>&1
DB=$(find /tmp/prof_iso -name "*.db" | head -1); [ -n "$DB" ] && python $R/profiles/summarize_rocpd.py $DB > $R/$O.kernel_trace_isolated_config4.txt
cd $R
VDL2_BENCH_REHEARSAL=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 2 --steps 5 --warmup 2 --duration 4 > $O.rehearsal2.json 2> $O.rehearsal2.err; echo "rehearsal rc=$?"; tail -c 1500 $O.rehearsal2.err
timeout 600 python bench.py --workload config5 --no-secondary > $O.bench_config5.json 2> $O.bench_config5.err; echo "config5 rc=$?"; tail -c 600 $O.bench_config5.err
