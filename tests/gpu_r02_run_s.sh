#!/bin/bash
# round 2, GPU call S: what each back-end stage costs the front by running beside it (ablation build, results not valid),
# and the walker's segment count
R="$(cd "$(dirname "$0")/.." && pwd)"
cd "$R"; mkdir -p gpurun_out
O=gpurun_out/r02s
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -fPIC -shared -DVDL2_ABLATE -o /tmp/ablate.so dumpvdl2_amd/csrc/vdl2hip.hip 2>/dev/null || echo build failed
: > $O.txt
one() {  # label, env...
  local label="$1"; shift
  env "$@" timeout 600 python bench.py --no-secondary --no-cpu-baseline --no-verify --workload config4 > $O.tmp.json 2> $O.err
  python - "$label" >> $O.txt <<'P'
import json,sys
try:
    j=json.loads(open('gpurun_out/r02s.tmp.json').read().strip().splitlines()[-1])
    st=j['config'].get('stage_ms_per_step',{})
    print(f"{sys.argv[1]:28s} host {j['value']:.1f} MS/s {j['ms_per_step']:.4f} ms | hbm {j.get('ms_per_step_hbm_resident')} ms | K1 {j['roofline'].get('avg_launch_ms'):.4f} " + " ".join(f"{k[:-3]} {v}" for k,v in st.items() if k!='chanfir_ms'))
except Exception as e:
    print(sys.argv[1], 'failed', e)
P
}
one base VDL2HIP_LIB=/tmp/ablate.so
one no-nf VDL2HIP_LIB=/tmp/ablate.so VDL2HIP_ABLATE=nf
one no-burst VDL2HIP_LIB=/tmp/ablate.so VDL2HIP_ABLATE=burst
one no-walk-nf-burst VDL2HIP_LIB=/tmp/ablate.so VDL2HIP_ABLATE=walk,nf,burst
one base VDL2HIP_LIB=/tmp/ablate.so
for sm in 1 2 4 6 10; do one seg_max=$sm VDL2HIP_SEG_MAX=$sm; done
cat $O.txt
