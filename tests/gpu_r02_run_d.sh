#!/bin/bash
# round 2, GPU call D: table-free atan2, group API; full bench with secondaries; profiles for profiles/
R="$(cd "$(dirname "$0")/.." && pwd)"
cd "$R"; mkdir -p gpurun_out
O=gpurun_out/r02d
timeout 900 python -m pytest tests -m gpu -x -q > $O.pytest.txt 2>&1; echo "pytest rc=$?" >> $O.pytest.txt
tail -3 $O.pytest.txt
timeout 900 python bench.py > $O.bench.json 2> $O.bench.err; echo "bench rc=$?"; tail -c 800 $O.bench.err
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 10 --warmup 2 --no-secondary --no-cpu-baseline > $O.bench_torchrun.json 2> $O.bench_torchrun.err; echo "torchrun bench rc=$?"
timeout 600 python bench.py --workload config5 --no-secondary > $O.bench_config5.json 2> $O.bench_config5.err; echo "config5 rc=$?"
cd /tmp && export TMPDIR=/tmp
for W in config4 config2; do
  rm -rf /tmp/prof_$W
  timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_$W -o r -- python $R/bench.py --workload $W --steps 10 --warmup 2 --no-cpu-baseline --no-verify --no-secondary > $R/$O.prof_$W.log 2>&1
  DB=$(find /tmp/prof_$W -name "*.db" | head -1); [ -n "$DB" ] && python $R/profiles/summarize_rocpd.py $DB > $R/$O.kernel_trace_bench_$W.txt
done
cd $R
timeout 600 bash tests/gpu_pmc_traffic.sh config4 > $O.pmc_hbm_traffic_config4.txt 2>&1
for C in 256 64 8; do timeout 400 bash tests/gpu_k1_pmc.sh $C > $O.sq_k1_${C}ch.txt 2>&1; done
for C in 8 64 256; do timeout 200 python tests/gpu_k1_bench.py $C 16 3 | cut -c1-230 >> $O.isolated.txt; VDL2HIP_NO_FUSE=1 timeout 200 python tests/gpu_k1_bench.py $C 16 3 | cut -c1-230 >> $O.isolated.txt; done
