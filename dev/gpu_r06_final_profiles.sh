#!/bin/bash
# round 6: rocprofv3 kernel trace of the bench command (without its secondary workloads, parity gate and CPU leg: the same K1 launches),
# then the PMC passes for K1's HBM traffic
R="$(cd "$(dirname "$0")/.." && pwd)"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt
timeout 110 rocprofv3 --kernel-trace --stats -d /tmp/kt -o kt -- python $R/bench.py --steps 20 --warmup 3 --repeats 1 --no-secondary --no-verify --no-cpu-baseline > /tmp/kt.log 2>&1
tail -1 /tmp/kt.log | cut -c1-300
DB=$(find /tmp/kt -name "*.db" | head -1)
if [ -n "$DB" ]; then { echo "# rocprofv3 --kernel-trace --stats -- python bench.py --steps 20 --warmup 3 --repeats 1 --no-secondary --no-verify --no-cpu-baseline (round 6, referee on)"; grep -o '"ms_per_step_hbm_resident": [0-9.]*\|"avg_launch_ms": [0-9.]*' /tmp/kt.log | head -3 | sed 's/^/# bench line: /'; python $R/profiles/summarize_rocpd.py $DB; } > $R/gpurun_out/r06_kernel_trace_bench_config4.txt; else echo "no db"; ls -R /tmp/kt | head; fi
head -16 $R/gpurun_out/r06_kernel_trace_bench_config4.txt | cut -c1-170
if [ "$1" != "trace" ]; then
timeout 130 bash $R/dev/gpu_pmc_traffic.sh config4 > $R/gpurun_out/r06_pmc_hbm_traffic_config4.txt 2>&1
grep chanfir $R/gpurun_out/r06_pmc_hbm_traffic_config4.txt | head -4
fi
