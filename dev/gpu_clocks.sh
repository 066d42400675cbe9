#!/bin/bash
# Development probe: shader clock and power while (1) the FMA micro-benchmark and (2) bench.py's timed region run.
# usage on the GPU box: bash dev/gpu_clocks.sh > gpurun_out/<tag>.clocks.txt
sample() {  # label, pid
	while kill -0 $2 2>/dev/null; do
		echo "$1 $(date +%s.%N | cut -c1-14) $(rocm-smi --showclocks --showpower 2>/dev/null | grep -E 'sclk|fclk|mclk|Power' | sed 's/.*GPU\[0\]//' | tr -s ' \t' ' ' | tr '\n' '|')"
		sleep 0.25
	done
}
hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -o /tmp/ub dev/gpu_ubench_valu.hip 2>/dev/null
echo "idle $(rocm-smi --showclocks --showpower 2>/dev/null | grep -E 'sclk|fclk|mclk|Power' | sed 's/.*GPU\[0\]//' | tr -s ' \t' ' ' | tr '\n' '|')"
( for i in 1 2 3 4 5 6; do /tmp/ub > /tmp/ub.out; done ) & P=$!
sample ubench $P | tail -12
head -4 /tmp/ub.out
python bench.py --no-secondary --no-cpu-baseline --no-verify --steps 200 --repeats 2 > /tmp/b.json 2>/dev/null & P=$!
sample bench $P | tail -24
python -c "import json; j=json.load(open('/tmp/b.json')); print(j['value'], j['ms_per_step'], j['ms_per_step_hbm_resident'])"
