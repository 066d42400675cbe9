#!/bin/bash
# round 2, GPU call Y: the channel-per-lane channeliser k_chanseq - suite (default dispatch and forced), K1 alone both ways, bench
R="$(cd "$(dirname "$0")/.." && pwd)"
cd "$R"; mkdir -p gpurun_out
O=gpurun_out/r02y
timeout 900 python -m pytest tests -x -q -m gpu > $O.pytest.txt 2>&1; tail -4 $O.pytest.txt
VDL2HIP_K1=seq timeout 900 python -m pytest tests -q -m gpu > $O.pytest_seq.txt 2>&1; tail -8 $O.pytest_seq.txt
for k in tile seq; do for C in 64 256; do VDL2HIP_K1=$k timeout 300 python tests/gpu_k1_bench.py $C 16 3 2>&1 | grep -v amdgpu.ids | sed "s/^default/$k/" | cut -c1-200; done; done | tee $O.k1.txt
for i in 1 2; do
timeout 600 python bench.py --no-secondary --no-cpu-baseline > $O.bench$i.json 2> $O.bench$i.err; python - $O.bench$i.json <<'P'
import json,sys
j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
st=j['config'].get('stage_ms_per_step',{})
print(f"host {j['value']:.1f} MS/s {j['ms_per_step']:.4f} ms | hbm {j.get('ms_per_step_hbm_resident')} ms | K1 {j['roofline'].get('avg_launch_ms'):.4f} " + " ".join(f"{k[:-3]} {v}" for k,v in st.items() if k!='chanfir_ms'), j['config'].get('verified'))
P
done
