#!/bin/bash
# dev/gpu_batch.sh <tag> - this round's variant batches (one gpurun call each); outputs under gpurun_out/<tag>.*
R="$(cd "$(dirname "$0")/.." && pwd)"
cd "$R"; mkdir -p gpurun_out
TAG=$1; shift
O=gpurun_out/$TAG
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -fPIC -shared"
build() { hipcc $FLAGS $2 -o /tmp/vdl2hip_$1.so dumpvdl2_amd/csrc/vdl2hip.hip 2>/dev/null || echo "build $1 failed"; }
for job in "$@"; do
	echo "=== $job"
	case $job in
	pytest) timeout 900 python -m pytest tests -m gpu -x -q > $O.pytest.txt 2>&1; echo "pytest rc=$?" >> $O.pytest.txt; tail -4 $O.pytest.txt ;;
	b1)
		build exp "-DVDL2_EXPERIMENTS"
		timeout 1500 python dev/gpu_variants.py --out $O.variants.jsonl --steps 16 --repeats 3 --variant base --variant r03:@dev/_ref/libvdl2hip_r03.so --variant syncwalk:@/tmp/vdl2hip_exp.so:VDL2HIP_SYNC_ON=walk --variant base2 2>&1 | tee $O.variants.txt ;;
	k5prof:*) build prof "-DVDL2_K5_PROF"; VDL2HIP_LIB=/tmp/vdl2hip_prof.so timeout 300 python dev/gpu_stage_times.py ${job#k5prof:} 16 2 2>&1 | grep -v amdgpu.ids | tee $O.k5prof_${job#k5prof:}.txt | cut -c1-220 ;;
	bench) timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 > $O.bench_default.json 2> $O.bench_default.err; echo "bench rc=$?"; tail -c 600 $O.bench_default.err; cut -c1-700 $O.bench_default.json ;;
	dropin) for w in config4 config2; do timeout 300 python dev/gpu_dropin_rate.py $w 4 2>&1 | grep -v amdgpu.ids | tee -a $O.dropin.txt; done ;;
	b2) timeout 900 python dev/gpu_variants.py --out $O.variants2.jsonl --steps 16 --repeats 3 --workloads config2,config3 --parts all --variant base --variant r03:@dev/_ref/libvdl2hip_r03.so 2>&1 | tee $O.variants2.txt ;;
	segs) timeout 1500 python dev/gpu_variants.py --out $O.segs.jsonl --steps 16 --repeats 3 --variant base --variant seg1:VDL2HIP_SEG_MAX=1 --variant seg2:VDL2HIP_SEG_MAX=2 --variant seg4:VDL2HIP_SEG_MAX=4 --variant seg6:VDL2HIP_SEG_MAX=6 --variant base2 2>&1 | tee $O.segs.txt ;;
	ablate)
		build exp "-DVDL2_EXPERIMENTS"
		timeout 1500 python dev/gpu_variants.py --out $O.ablate.jsonl --steps 16 --repeats 3 --workloads config4,config4_bursty --variant full:@/tmp/vdl2hip_exp.so --variant nowalk:@/tmp/vdl2hip_exp.so:VDL2HIP_ABLATE=walk --variant nonf:@/tmp/vdl2hip_exp.so:VDL2HIP_ABLATE=nf --variant noburst:@/tmp/vdl2hip_exp.so:VDL2HIP_ABLATE=burst --variant noback:@/tmp/vdl2hip_exp.so:VDL2HIP_ABLATE=walk,nf,burst 2>&1 | tee $O.ablate.txt ;;
	pmc:*) W=${job#pmc:}; timeout 900 bash dev/gpu_pmc_all.sh $W all > $O.pmc_all_$W.txt 2>&1; cut -c1-120 $O.pmc_all_$W.txt ;;
	fuzz:*) timeout $(( ${job#fuzz:} + 120 )) python tests/fuzz_gpu.py ${job#fuzz:} ${FUZZ_SEED0:-1} all > $O.fuzz_gpu.txt 2>&1; echo "fuzz rc=$?"; grep -c ': ok' $O.fuzz_gpu.txt; grep -v ': ok' $O.fuzz_gpu.txt | cut -c1-600 | tail -12 ;;
	*) echo "unknown job $job" ;;
	esac
done
