#!/bin/bash
# The one GPU-box runner: dev/gpu_run.sh <tag> <job> [<job> ...]   (outputs under gpurun_out/<tag>.*; copy what is to be judged to profiles/)
#   pytest        python -m pytest tests -m gpu
#   bench         the driver's line: python bench.py --gpus 1 --steps 20 --warmup 5
#   bench5        config5 (injected errors), no secondary / cpu baseline
#   benchw:<w>    bench.py --workload <w> without gate / baseline / secondary: value, ms/step, stage times
#   shard32       dev/gpu_shard32.py: the rank-sized workload (32 of 256 channels), ranks 0,3,7
#   trace32       rocprofv3 --kernel-trace --stats of the rank-sized workload (rank 3)
#   trace256      rocprofv3 --kernel-trace --stats of bench.py config4
#   tracedropin   the same of dev/gpu_dropin_rate.py config4 (320 000-byte blocks)
#   pmc256        HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes) of bench.py config4
#   sq_k1:<C>     SQ counters of the channeliser on C channels of noise;  sq_k3a:<C> the same for k_sync_screen
#   iso           per-stage kernel times of config4, nothing overlapped
#   k1:<C>        dev/gpu_k1_bench.py on C channels
#   ubench        dev/gpu_ubench_valu.hip: issue rates, LDS gathers, and K1's inner loop with its tap sums on the VALU / on the matrix pipe
#   k1phases:<C>  -DVDL2_K1_PROF build: shader clocks per phase of a channeliser tile (dev/gpu_k1_phases.py)
#   ubclock       dev/gpu_ubench_clock.hip: issue costs in shader clocks (s_memtime), chip at its working clock
#   clocks        dev/gpu_clocks.sh: rocm-smi clock / power samples while the micro-benchmark and bench.py run
#   env:<A=B>     export A=B for the jobs that follow;  unenv:<A>  unset it
#   exp           build the library with -DVDL2_EXPERIMENTS (the VDL2HIP_CR / K1_TILES / K3B_WPL / SYNC_ON / LOW_PRIO / ABLATE / GAPS
#                 switches exist only there) into /tmp/vdl2hip_exp.so and point VDL2HIP_LIB at it for the jobs that follow
#   exp:<flags>   the same with extra compiler flags (-D...)
R="$(cd "$(dirname "$0")/.." && pwd)"
cd "$R"; mkdir -p gpurun_out
TAG=$1; shift
O=gpurun_out/$TAG
sfx() { env | grep '^VDL2HIP_' | sort | tr '\n' ' ' ; }
trace() {  # name, command...
	local name=$1; shift
	( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_$name && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_$name -o r -- "$@" > /tmp/prof_$name.log 2>&1 )
	local DB=$(find /tmp/prof_$name -name "*.db" | head -1)
	if [ -n "$DB" ]; then { echo "# rocprofv3 --kernel-trace --stats -- $*   [$(sfx)]"; python $R/profiles/summarize_rocpd.py $DB; } > $O.kernel_trace_$name.txt; else echo "no trace db for $name"; tail -5 /tmp/prof_$name.log; fi
}
for job in "$@"; do
	echo "=== $job [$(sfx)]"
	case $job in
	exp|exp:*) X="${job#exp}"; X="${X#:}"; hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -fPIC -shared -DVDL2_EXPERIMENTS $X -o /tmp/vdl2hip_exp.so dumpvdl2_amd/csrc/vdl2hip.hip 2>/dev/null && export VDL2HIP_LIB=/tmp/vdl2hip_exp.so || echo "experiment build failed" ;;
	env:*) export "${job#env:}" ;;
	unenv:*) unset "${job#unenv:}" ;;
	pytest) timeout 1200 python -m pytest tests -m gpu -x -q > $O.pytest.txt 2>&1; echo "pytest rc=$?" >> $O.pytest.txt; tail -3 $O.pytest.txt ;;
	bench) timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O.bench_default.json 2> $O.bench_default.err; echo "bench rc=$?"; tail -c 400 $O.bench_default.err; cut -c1-600 $O.bench_default.json ;;
	bench5) timeout 600 python bench.py --workload config5 --no-secondary --no-cpu-baseline > $O.bench_config5.json 2> $O.bench_config5.err; echo "config5 rc=$?"; cut -c1-400 $O.bench_config5.json ;;
	benchq) timeout 600 python bench.py --no-secondary --no-cpu-baseline --no-verify >> $O.benchq.jsonl 2>> $O.benchq.err; echo "rc=$?"; tail -1 $O.benchq.jsonl | python -c "import json,sys; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'], j['ms_per_step_hbm_resident'], j['config']['stage_ms_per_step'], j['config'].get('stage_ms_per_step_hbm_resident'), j['roofline']['avg_launch_ms'], j['roofline']['avg_launch_ms_host_fed'])" ;;
	benchw:*) timeout 600 python bench.py --workload ${job#benchw:} --no-secondary --no-cpu-baseline --no-verify --repeats 2 2>> $O.benchw.err | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('$job [$(sfx)]', j['value'], j['ms_per_step'], j['ms_per_step_hbm_resident'], j['config']['stage_ms_per_step'])" | tee -a $O.benchw.txt ;;
	shard32) timeout 900 python dev/gpu_shard32.py --json $O.shard32.json 2> $O.shard32.err | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
for r in j['ranks']: print('rank', r['rank'], r['ms_per_step'], 'K1', r['k_chanfir_ms'], 'pipe', r['stage_ms_in_pipeline'], 'alone', r['stage_ms_alone'], 'lat', r['ms_per_step_one_block_in_flight'])
"; tail -c 300 $O.shard32.err ;;
	shard32q) timeout 600 python dev/gpu_shard32.py --ranks 3 --repeats 2 2>> $O.shard32q.err | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
for r in j['ranks']: print(j['env'], 'rank', r['rank'], r['ms_per_step'], 'K1', r['k_chanfir_ms'], 'pipe', r['stage_ms_in_pipeline'])
" | tee -a $O.shard32q.txt ;;
	sweep32:*) timeout 900 python dev/gpu_shard32.py --ranks 3 --repeats 2 --sweep "${job#sweep32:}" --json $O.sweep32_$(date +%s).json 2>&1 >/dev/null | grep '^{' | cut -c1-330 | tee -a $O.sweep32.txt ;;
	trace32) trace shard32 python $R/dev/gpu_shard32.py --ranks 3 --repeats 1 --no-check ;;
	tracedropin) trace dropin python $R/dev/gpu_dropin_rate.py config4 2 ;;
	trace256) trace bench_config4 python $R/bench.py --workload config4 --steps 10 --warmup 2 --no-cpu-baseline --no-verify --no-secondary ;;
	pmc256) timeout 500 bash dev/gpu_pmc_traffic.sh config4 > $O.pmc_hbm_traffic_config4.txt 2>&1; cat $O.pmc_hbm_traffic_config4.txt | cut -c1-160 ;;
	sq_k1:*) timeout 500 bash dev/gpu_k1_pmc.sh ${job#sq_k1:} > $O.sq_k1_${job#sq_k1:}ch.txt 2>&1; cut -c1-140 $O.sq_k1_${job#sq_k1:}ch.txt ;;
	sq_k3a:*) KFILTER=sync_screen timeout 500 bash dev/gpu_k1_pmc.sh ${job#sq_k3a:} > $O.sq_k3a_${job#sq_k3a:}ch.txt 2>&1; cut -c1-140 $O.sq_k3a_${job#sq_k3a:}ch.txt ;;
	k5prof:*) hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -fPIC -shared -DVDL2_K5_PROF -o /tmp/vdl2hip_prof.so dumpvdl2_amd/csrc/vdl2hip.hip 2>/dev/null && VDL2HIP_LIB=/tmp/vdl2hip_prof.so timeout 300 python dev/gpu_stage_times.py ${job#k5prof:} 16 2 2>&1 | grep -v amdgpu.ids | tee $O.k5prof_${job#k5prof:}.txt | cut -c1-200 ;;
	iso) timeout 300 python dev/gpu_stage_times.py config4 16 3 2>&1 | grep -v amdgpu.ids | tee $O.stage_times_alone.txt | cut -c1-400 ;;
	k1:*) timeout 300 python dev/gpu_k1_bench.py ${job#k1:} 16 3 2>&1 | grep -v amdgpu.ids | cut -c1-260 | tee -a $O.k1.txt ;;
	ubench) hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -o /tmp/ub dev/gpu_ubench_valu.hip 2>/dev/null && timeout 300 /tmp/ub 2>&1 | tee $O.ubench.txt | head -12 ;;
	k1phases:*) hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -fPIC -shared -DVDL2_K1_PROF -o /tmp/vdl2hip_k1prof.so dumpvdl2_amd/csrc/vdl2hip.hip 2>/dev/null && VDL2HIP_LIB=/tmp/vdl2hip_k1prof.so timeout 300 python dev/gpu_k1_phases.py ${job#k1phases:} 2>&1 | grep -v amdgpu.ids | tee -a $O.k1_phases.txt ;;
	ubclock) hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -Wno-unused-result -o /tmp/ubc dev/gpu_ubench_clock.hip 2>/dev/null && timeout 300 /tmp/ubc 2>&1 | tee $O.ubench_clock.txt ;;
	clocks) timeout 400 bash dev/gpu_clocks.sh > $O.clocks.txt 2>&1; tail -5 $O.clocks.txt ;;
	*) echo "unknown job $job" ;;
	esac
done
